# Profile set of the SHIPPED library, one box, one run directory: bench lines (default + sustained), kernel traces and PMC passes
# of the BASELINE shapes and of the slab-tiled kernel, the chain-major (3-launch) variant of the default workload, the split
# A/B, the in-kernel stage profiles and one bench line per frame-count family.  Every file it emits starts with (JSON: holds) a
# MANIFEST -- git HEAD (passed in: the GPU box has no .git), sha256 of mocodad_amd/libmocodad_hip.so, box id, run directory --
# and bench.py refuses PMC numbers whose manifest hash is not that of the library it loaded.
#   usage: gpurun -- bash tools/profile_set.sh <tag> "<configs or empty>" <git HEAD>
set -x
TAG=${1:-r06zz}
CFGS=${2:-"avenue stc ubnormal_concat seq24"}
PCFGS=${2:-"avenue avenue_chainmajor stc ubnormal_concat seq24 concat24 concat32"}
HEAD=${3:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SO=$(sha256sum mocodad_amd/libmocodad_hip.so | cut -c1-64)
BOX=$( (cat /etc/machine-id 2>/dev/null || hostname) | cut -c1-12)
GPU=$(python -c "import torch; print(torch.cuda.get_device_name(0).replace(' ', '_'))" 2>/dev/null)
MAN="# manifest: head=$HEAD so_sha256=$SO box=$BOX gpu=$GPU run_dir=$TAG date=$(date -u +%Y-%m-%dT%H:%MZ)"
echo "$MAN" > $O/MANIFEST.txt
stamp() { { echo "$MAN"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
stamp_json() { python - "$1" "$HEAD" "$SO" "$BOX" "$TAG" <<'PY'
import json, sys
p, head, so, box, tag = sys.argv[1:6]
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    d["manifest"] = {"head": head, "so_sha256": so, "box": box, "run_dir": tag}
    open(p, "w").write(json.dumps(d) + "\n")
except Exception as e:
    print("stamp_json:", p, e)
PY
}
for c in $CFGS; do
  timeout 600 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  stamp_json $O/bench_$c.json
done
if [ -z "$2" ]; then
# sustained: a TIMED region of >= 10 s for the four BASELINE shapes (value = the sustained rate)
for c in $CFGS; do
  timeout 600 python bench.py --config $c --min-seconds 10 --no-cpu-baseline --no-auc --sustained-seconds 0 --e2e-windows 0 > $O/sustained_$c.json 2> $O/sustained_$c.err
  stamp_json $O/sustained_$c.json
done
# every other frame-count family: one line each (no CPU leg)
{ for c in seg4 seg10 seg14 seg16 seg18 seg20 seg22 concat12 seg32 concat24 concat32 seg10_eunet seg32_eunet; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s %10.1f clips/s  frac %.4f  kernel ms/step %8.4f  %s' % ('$c', d['value'], r['frac'], r['kernel_ms_per_step'], r.get('kernel', '')))"
  done; } > $O/shapes.txt 2>&1
stamp $O/shapes.txt
# chain-major (3 launches) vs window-major (one launch, the default) on this box, interleaved
{ for rep in 1 2 3; do for sp in 5 1; do echo -n "avenue B=1024 split=$sp: "; python bench.py --split $sp --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s  frac', d['roofline']['frac'], ' kernel ms/step', d['roofline']['kernel_ms_per_step'])"; done; done
  for sp in 5 1; do echo -n "avenue B=4096 split=$sp: "; python bench.py --batch 4096 --split $sp --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s  frac', d['roofline']['frac'])"; done; } > $O/split_ab.txt 2>&1
stamp $O/split_ab.txt
# the shipped sample count of the reference's test configs (n_generated_samples: 50)
timeout 600 python bench.py --samples 50 --steps 20 --warmup 2 --no-cpu-baseline --no-auc --sustained-seconds 0 --e2e-windows 0 > $O/bench_avenue_S50.json 2> $O/bench_avenue_S50.err
stamp_json $O/bench_avenue_S50.json
# in-kernel stage profiles (a -DMCD_PROFILE build of the same sources)
for c in avenue ubnormal_concat seq24; do
  timeout 900 python tools/stage_profile.py 0 $c > $O/${c}_stage_profile.txt 2> $O/${c}_stage_profile.err
  stamp $O/${c}_stage_profile.txt
done
for c in concat24 concat32; do      # (the slab-tiled kernel: stage cycles of waves 0 / last + the per-wave event trace of one pass)
  timeout 600 python tools/tiled_stage_profile.py $c 2>&1 | grep -v amdgpu.ids > $O/${c}_stage_profile.txt
  stamp $O/${c}_stage_profile.txt
done
# shard-sized end-to-end runs (eval_MoCoDAD.py: test_step loop + gather + frame scores + AUC)
bash tools/e2e_shards.sh $TAG $HEAD > /dev/null 2>&1
fi
cd /tmp && export TMPDIR=/tmp
for c in $PCFGS; do
  st=""; ex=""; cfg=$c                      # kernel trace: bench.py's own default steps / warm-up (the command of the bench line)
  [ $c = seq24 ] && st="--steps 3 --warmup 2" && ex="--batch 1024"
  [ $c = avenue_chainmajor ] && cfg=avenue && ex="--split 5"
  [ $c = concat32 ] && st="--steps 3 --warmup 2"
  [ $c = concat24 ] && st="--steps 3 --warmup 2"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -- python $R/bench.py --config $cfg $st --no-cpu-baseline --no-extras $ex > $O/prof_$c.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/prof_$c -name "*_results.db" | head -1) > $O/${c}_kernel_stats.txt
  stamp $O/${c}_kernel_stats.txt
  rm -rf $O/prof_$c
  [ $c = avenue_chainmajor ] && continue
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_${c}_$i -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-extras $ex > $O/pmc_${c}_$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $(find $O/pmc_${c}_* -name "*_results.db") > $O/${c}_pmc.txt
  stamp $O/${c}_pmc.txt
  rm -rf $O/pmc_${c}_*
done
rm -f $O/*.log
ls -la $O; [ -f $O/split_ab.txt ] && cat $O/split_ab.txt

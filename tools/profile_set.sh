# Profile set of the shipped kernels (rounds 2+): bench lines, kernel traces and PMC passes of the three trajectory-kernel shapes
# and of the slab-tiled kernel, the chain-major (3-launch) variant of the default workload, the split A/B, and one bench line per
# frame-count family.   usage: gpurun -- bash tools/profile_set.sh <tag> ["<configs>"]
set -x
TAG=${1:-r03m}
CFGS=${2:-"avenue stc ubnormal_concat seq24"}
PCFGS=${2:-"avenue avenue_chainmajor ubnormal_concat seq24 concat32"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for c in $CFGS; do
  timeout 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
done
if [ -z "$2" ]; then
# every other frame-count family: one line each (no CPU leg)
{ for c in seg4 seg10 seg14 seg16 seg18 seg20 seg22 concat12 seg32 concat24 concat32 seg10_eunet seg32_eunet; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s %10.1f clips/s  frac %.4f  kernel ms/step %8.4f  %s' % ('$c', d['value'], r['frac'], r['kernel_ms_per_step'], r.get('kernel', '')))"
  done; } > $O/shapes.txt 2>&1
# chain-major (3 launches) vs window-major (one launch, the default) on this box, interleaved
{ for rep in 1 2 3; do for sp in 5 1; do echo -n "avenue B=1024 split=$sp: "; python bench.py --split $sp --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s  frac', d['roofline']['frac'], ' kernel ms/step', d['roofline']['kernel_ms_per_step'])"; done; done
  for sp in 5 1; do echo -n "avenue B=4096 split=$sp: "; python bench.py --batch 4096 --split $sp --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'clips/s  frac', d['roofline']['frac'])"; done; } > $O/split_ab.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
for c in $PCFGS; do
  st=""; ex=""; cfg=$c                      # kernel trace: bench.py's own default steps / warm-up (the command of the bench line)
  [ $c = seq24 ] && st="--steps 3 --warmup 2" && ex="--batch 1024"
  [ $c = avenue_chainmajor ] && cfg=avenue && ex="--split 5"
  [ $c = concat32 ] && st="--steps 3 --warmup 2"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -- python $R/bench.py --config $cfg $st --no-cpu-baseline --no-extras $ex > $O/prof_$c.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/prof_$c -name "*_results.db" | head -1) > $O/${c}_kernel_stats.txt
  rm -rf $O/prof_$c
  [ $c = avenue_chainmajor ] && continue
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_${c}_$i -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-extras $ex > $O/pmc_${c}_$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $(find $O/pmc_${c}_* -name "*_results.db") > $O/${c}_pmc.txt
  rm -rf $O/pmc_${c}_*
done
ls -la $O; [ -f $O/split_ab.txt ] && cat $O/split_ab.txt

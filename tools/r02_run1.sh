# round 2, GPU call 1: full GPU test suite, bench lines of the four configurations, kernel traces + SQ counters of the three
# trajectory-kernel shapes.  usage: gpurun -- bash tools/r02_run1.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for c in avenue stc ubnormal_concat seq24; do
  st=20; [ $c = seq24 ] && st=5
  timeout 400 python bench.py --config $c --steps $st > $O/bench_$c.json 2> $O/bench_$c.err
  tail -c 400 $O/bench_$c.json
done
cd /tmp && export TMPDIR=/tmp
for c in avenue ubnormal_concat seq24; do
  st=20; ex=""; [ $c = seq24 ] && st=3 && ex="--batch 1024"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -- python $R/bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-extras $ex > $O/prof_$c.log 2>&1
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_${c}_$i -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extras $ex > $O/pmc_${c}_$i.log 2>&1
  done
  db=$(find $O/prof_$c -name "*_results.db" | head -1)
  python $R/tools/rocpd_summary.py $db > $O/${c}_kernel_stats.txt
  python $R/tools/pmc_summary.py $(find $O/pmc_${c}_* -name "*_results.db") > $O/${c}_pmc.txt
  # keep the merged-back output small: the databases stay on the box
  rm -rf $O/prof_$c $O/pmc_${c}_*
done
ls -la $O

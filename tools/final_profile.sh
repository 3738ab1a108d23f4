set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 300 python $R/bench.py > $O/bench_r01t.json 2> $O/bench_r01t.err
tail -c 600 $O/bench_r01t.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_r1t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_r1t.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_t$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_t$i.log 2>&1
  tail -1 $O/pmc_t$i.log | head -c 200; echo
done
find $O/prof_r1t $O/pmc_t* -name "*_results.db" | head
# the opt-in split-bf16 GEMM path (not the headline): bench line + kernel trace
timeout 300 python $R/bench.py --bf16x3 > $O/bench_r01t_bf16x3.json 2> $O/bench_r01t_bf16x3.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_r1t_bf16x3 -- python $R/bench.py --bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_r1t_bf16x3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/pmc_t_bf16x3 -- python $R/bench.py --bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_t_bf16x3.log 2>&1

mkdir -p gpurun_out/r05y
{ bash tools/ab_bench.sh "--config concat24" mocodad_amd/lib_d0.so mocodad_amd/lib_d1.so
bash tools/ab_bench.sh "--config concat32" mocodad_amd/lib_e0.so mocodad_amd/lib_e1.so
} > gpurun_out/r05y/dense_ab.txt 2>&1
cat gpurun_out/r05y/dense_ab.txt

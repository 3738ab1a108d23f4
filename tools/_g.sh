mkdir -p gpurun_out/r05zi
{ bash tools/ab_bench.sh "--config concat32" mocodad_amd/lib_e0.so mocodad_amd/lib_e1.so; bash tools/ab_bench.sh "--config concat32" mocodad_amd/lib_e0.so mocodad_amd/lib_e1.so; } > gpurun_out/r05zi/tiled32_l5w2_ab2.txt 2>&1
cat gpurun_out/r05zi/tiled32_l5w2_ab2.txt

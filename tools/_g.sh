mkdir -p gpurun_out/r05ze
bash tools/ab_bench.sh "--config concat32" mocodad_amd/lib_e0.so mocodad_amd/lib_e1.so > gpurun_out/r05ze/tiled32_l10mfma_ab.txt 2>&1
cat gpurun_out/r05ze/tiled32_l10mfma_ab.txt

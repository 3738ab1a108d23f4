mkdir -p gpurun_out/r05t
{ bash tools/ab_bench.sh "--config seq24 --batch 1024 --steps 3 --warmup 1" mocodad_amd/lib_t12_minreg.so mocodad_amd/lib_t12_defsched.so mocodad_amd/lib_t12_minreg_pd2.so mocodad_amd/lib_t12_defsched_pd2.so
bash tools/ab_bench.sh "--config ubnormal_concat" mocodad_amd/lib_t6_stash1.so mocodad_amd/lib_t6_stash0.so mocodad_amd/lib_t6_stash1_lo4.so mocodad_amd/lib_t6_stash0_lo4.so
bash tools/ab_bench.sh "--config seg18" mocodad_amd/lib_t9_minreg.so mocodad_amd/lib_t9_defsched.so
} > gpurun_out/r05t/slack_ab.txt 2>&1
cat gpurun_out/r05t/slack_ab.txt

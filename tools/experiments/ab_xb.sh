cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "" mocodad_amd/libab_full_xb0.so mocodad_amd/libmocodad_hip.so
bash tools/ab_bench.sh "--config seg4" mocodad_amd/libab_full_xb0.so mocodad_amd/libmocodad_hip.so
bash tools/ab_bench.sh "--config seg10" mocodad_amd/libab_full_xb0.so mocodad_amd/libmocodad_hip.so
bash tools/ab_bench.sh "--config seg10 --batch 2048" mocodad_amd/libab_full_xb0.so mocodad_amd/libmocodad_hip.so

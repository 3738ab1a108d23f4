cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_layers_gpu.py tests/test_samples50_gpu.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh "" mocodad_amd/libab_full_base.so mocodad_amd/libmocodad_hip.so
bash tools/ab_bench.sh "--config ubnormal_concat" mocodad_amd/libab_full_base.so mocodad_amd/libmocodad_hip.so
bash tools/ab_bench.sh "--config seq24 --batch 1024 --steps 3 --warmup 1" mocodad_amd/libab_full_base.so mocodad_amd/libmocodad_hip.so
for c in seg4 seg10 seg14 seg20; do bash tools/ab_bench.sh "--config $c" mocodad_amd/libab_full_base.so mocodad_amd/libmocodad_hip.so | head -4; done

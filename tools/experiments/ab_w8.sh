cd $GRAFT_REPO_ROOT
echo "== parity with the 3-frame W-first build"
MCD_LIB=$PWD/mocodad_amd/libab_w8_t3.so python -m pytest tests/test_hip_parity.py -q -x -k "inject and not concat and not T12" 2>&1 | tail -5
echo "== A/B"
bash tools/ab_bench.sh "" mocodad_amd/libab_old_t3.so mocodad_amd/libab_w8_t3.so
bash tools/ab_bench.sh "--config ubnormal_concat" mocodad_amd/libab_old_t6.so mocodad_amd/libab_w8_t6.so
bash tools/ab_bench.sh "--config seq24 --batch 1024 --steps 3 --warmup 1" mocodad_amd/libab_old_t12.so mocodad_amd/libab_w8_t12.so

for rep in 1 2 3; do
for ph in -1 0 -32 -33 -34 -35 -36; do
  echo -n "phase $ph: "
  MCD_LIB=$PWD/mocodad_amd/libtl16_prio.so timeout 300 python bench.py --config seg32 --steps 20 --warmup 3 --phase $ph --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done; done

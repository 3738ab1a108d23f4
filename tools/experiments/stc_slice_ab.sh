# stc (2048 windows: two rounds of workgroups): the self-calibrated priority slice (default) against the host's estimate (--phase -2)
# and fixed slices (--phase -(16 + shift))
for rep in 1 2 3; do
for ph in 0 -2 -31 -32 -33 -34; do
  echo -n "stc phase $ph: "
  timeout 300 python bench.py --config stc --steps 60 --warmup 10 --phase $ph --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done; done

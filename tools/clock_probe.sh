# samples sclk / power while a long bench.py run is on the GPU (is the FP32 peak's 2.4 GHz the clock the kernel runs at?)
(python bench.py --steps 8000 --no-cpu-baseline --no-extras > gpurun_out/clk_bench.txt 2>&1 &)
for i in $(seq 1 26); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*: //' | tr '\n' ' ')"; sleep 1; done
wait
sleep 2
tail -1 gpurun_out/clk_bench.txt | cut -c1-200

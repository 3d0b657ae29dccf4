#!/bin/bash
# Samples the GPU's shader clock and power (rocm-smi, read-only) while bench.py's timed loop runs: what clock the
# fp32 MFMA kernels actually sustain, against the 2.4 GHz the 157.3 TFLOP/s peak is quoted at.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clock; mkdir -p $O
rocm-smi --showclocks --showpower > $O/idle.txt 2>&1
timeout -k 10 300 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12                                      # library load + the first (warm-up) step
for i in $(seq 1 16); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';' >> $O/samples.txt
  echo >> $O/samples.txt
  sleep 0.5
done
wait $BP
echo "bench rc $?" >> $O/samples.txt

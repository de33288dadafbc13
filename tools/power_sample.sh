#!/bin/bash
# Samples rocm-smi power / sclk while the 512^2x128 f16s8 train step loops (evidence for DESIGN 3.4: power-limited chain kernel).
# usage (on the GPU box): bash tools/power_sample.sh > gpurun_out/power_sample.log
cd "$GRAFT_REPO_ROOT"
python3 bench.py --no-cpu --steps 150 --warmup 2 > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/.*sclk clock level: /sclk /; s/.*Power (W): /W /' | tr '\n' ' '
  echo
  sleep 0.3
done
wait $BP
tail -c 300 gpurun_out/power_bench.json

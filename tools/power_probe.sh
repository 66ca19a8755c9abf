#!/bin/bash
# Sample package power and shader clock (rocm-smi) while the bench loop runs: tools/power_probe.sh [steps]
python bench.py --no-cpu-baseline --no-rollouts --no-profile --steps ${1:-100} --warmup 2 > gpurun_out/pp_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/.*sclk clock level: [^(]*(\([0-9]*Mhz\)).*/sclk \1/' -e 's/.*Power (W): \(.*\)/power \1 W/' | tr '\n' ' '; echo
  sleep 0.5
done
tail -1 gpurun_out/pp_bench.json | cut -c1-200

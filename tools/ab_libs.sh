#!/bin/bash
# Same-box A/B of library builds (boxes differ by up to 9 % in frames/s, so only alternating runs on ONE box resolve
# sub-percent changes).  On the authoring machine:
#     mkdir -p tmp_ab; python -m tspo_amd.build; cp tspo_amd/libtspo_hip.so tmp_ab/lib_prev.so
#     <edit a kernel>;  python -m tspo_amd.build; cp tspo_amd/libtspo_hip.so tmp_ab/lib_new.so
#     gpurun -- 'ROUNDS=3 bash tools/ab_libs.sh prev new'        (tmp_ab/ travels with the snapshot; delete it afterwards)
# Prints, per build and round: frames/s, GEMM TFLOP/s, the per-class kernel times, and the clock / power sampled during the
# timed steps (the encode is power-limited: a change that moves bytes shows up in sclk before it shows up in a micro-benchmark).
cp tspo_amd/libtspo_hip.so tmp_ab/.lib_shipped.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    cp tmp_ab/lib_$v.so tspo_amd/libtspo_hip.so
    timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-rollouts --no-pruned --no-720p --no-comm-probe | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; b=r['breakdown_ms']
print('%-8s round $round: %8.1f frames/s  step %7.3f ms (min %7.3f)  GEMM %6.1f TFLOP/s %7.2f ms  attn %6.3f ms  other %5.2f ms  sclk %6.1f MHz  %6.1f W' % ('$v', j['value'], j['ms_per_step'], j['ms_per_step_min'], r['achieved'], b['gemm_ms'], b['attn_ms'], b['total_ms']-b['gemm_ms']-b['attn_ms'], r.get('sclk_mhz') or 0, r.get('power_w') or 0))"
  done
done
cp tmp_ab/.lib_shipped.so tspo_amd/libtspo_hip.so

#!/bin/bash
# Same-box A/B of library builds (boxes differ by up to 9 % in frames/s, so only alternating runs on ONE box resolve
# sub-percent changes).  On the authoring machine:
#     mkdir -p tmp_ab; python -m tspo_amd.build; cp tspo_amd/libtspo_hip.so tmp_ab/lib_prev.so
#     <edit a kernel>;  python -m tspo_amd.build; cp tspo_amd/libtspo_hip.so tmp_ab/lib_new.so
#     gpurun -- 'bash tools/ab_libs.sh prev new'        (tmp_ab/ travels with the snapshot; delete it afterwards)
# Prints frames/s, GEMM TFLOP/s and the per-class kernel times for each build, two alternating rounds.
for round in 1 2; do
  for v in "$@"; do
    cp tmp_ab/lib_$v.so tspo_amd/libtspo_hip.so
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollouts --no-pruned --no-720p --no-comm-probe | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['roofline']['achieved'], j['roofline']['breakdown_ms'])"
  done
done

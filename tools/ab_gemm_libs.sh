#!/bin/bash
# Same-box A/B of library builds on the four encoder GEMM shapes (tools/bench_gemm.py, production variant), alternating processes:
#     gpurun -- 'ROUNDS=3 bash tools/ab_gemm_libs.sh base new'      (libraries tmp_ab/lib_<name>.so, see tools/ab_libs.sh)
cp tspo_amd/libtspo_hip.so tmp_ab/.lib_shipped.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    cp tmp_ab/lib_$v.so tspo_amd/libtspo_hip.so
    timeout 600 python tools/bench_gemm.py ${T:-1024} ${VARIANTS:-77} ${REPS:-4} 2>&1 | grep -E "^(qkv|out|fc1|fc2)" | sed "s/^/$v round $round: /"
  done
done
cp tmp_ab/.lib_shipped.so tspo_amd/libtspo_hip.so

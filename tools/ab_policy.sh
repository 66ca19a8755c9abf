#!/bin/bash
# same-box A/B of library builds on the policy step:  ROUNDS=3 MODE=dpc bash tools/ab_policy.sh base new   (MODE: "" = B=4 fused step, dp, dpc)
cp tspo_amd/libtspo_hip.so tmp_ab/.lib_shipped.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    cp tmp_ab/lib_$v.so tspo_amd/libtspo_hip.so
    echo -n "$v round $round: "; timeout 300 python tools/bench_policy.py 400 fp32 $MODE 2>/dev/null | grep "rollouts/s"
  done
done
cp tmp_ab/.lib_shipped.so tspo_amd/libtspo_hip.so

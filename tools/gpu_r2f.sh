#!/bin/bash
# same-box whole-encoder A/B: 8-wave LDS-DMA kernel (6) vs 4-wave AGPR register-staged kernel (82), alternating
OUT=gpurun_out/r2f; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 6 82; do
    TSPO_GEMM_VARIANT=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollouts --no-pruned > $OUT/bench_v${v}_$rep.json 2> $OUT/bench_v${v}_$rep.err
    python - <<PY
import json
j=json.loads(open("$OUT/bench_v${v}_$rep.json").read().strip().splitlines()[-1])
print("variant $v rep $rep:", j["value"], "frames/s", j["roofline"]["achieved"], "TF gemm", j["roofline"]["breakdown_ms"])
PY
  done
done

#!/bin/bash
OUT=gpurun_out/r2h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_bf16" 2>&1 | tail -5 | tee $OUT/pytest_gemm.log
timeout 400 python tools/bench_gemm_fit.py 82 2>&1 | grep variant | tee $OUT/fit.txt
timeout 600 python tools/bench_gemm.py 1024 6,82,-1 6 2>&1 | grep -v amdgpu | tee $OUT/bench_gemm.log
TSPO_GEMM_VARIANT=82 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "clip_vit_forward_70" -s 2>&1 | tail -8 | tee $OUT/pytest_clip.log
for rep in 1 2; do
  for v in 6 82; do
    TSPO_GEMM_VARIANT=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollouts --no-pruned > $OUT/bench_v${v}_$rep.json 2> $OUT/bench_v${v}_$rep.err
    python - <<PY
import json
j=json.loads(open("$OUT/bench_v${v}_$rep.json").read().strip().splitlines()[-1])
print("variant $v rep $rep:", j["value"], "frames/s", j["roofline"]["achieved"], "TF gemm", j["roofline"]["breakdown_ms"])
PY
  done
done

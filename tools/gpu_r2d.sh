#!/bin/bash
set -x
OUT=gpurun_out/r2d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_bf16" 2>&1 | tail -15 > $OUT/pytest_gemm.log
cat $OUT/pytest_gemm.log
timeout 600 python tools/bench_gemm.py 1024 6,82,-1 6 > $OUT/bench_gemm.log 2>&1
cat $OUT/bench_gemm.log
TSPO_GEMM_VARIANT=82 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "clip_vit_forward_70" -s 2>&1 | tail -15 > $OUT/pytest_clip.log
cat $OUT/pytest_clip.log
TSPO_GEMM_VARIANT=82 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollouts --no-pruned > $OUT/bench_v82.json 2> $OUT/bench_v82.err
cat $OUT/bench_v82.json
tail -3 $OUT/bench_v82.err

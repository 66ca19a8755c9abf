#!/bin/bash
# round 5, first GPU pass: remainder-phase correctness + A/B, e2e diagnostics on three heavy-tailed videos
OUT=gpurun_out/r5a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | tail -15 | tee $OUT/pytest_gemm.log
timeout 600 python tools/bench_gemm_tail.py 7 2>&1 | tee $OUT/gemm_tail_ab.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -s 2>&1 | tail -80 | tee $OUT/pytest_e2e.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-1200 $OUT/bench.json

#!/bin/bash
set -x
OUT=gpurun_out/r2c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/bench_gemm.py 1024 6,80,82,-1 5 > $OUT/bench_gemm.log 2>&1
cat $OUT/bench_gemm.log

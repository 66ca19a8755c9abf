#!/bin/bash
# full -m gpu suite + smoke + default bench (what the driver runs at round end)
OUT=gpurun_out/${1:-full}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json | cut -c1-1500

#!/bin/bash
# Round profile set (run on the GPU box from the repo root):  tools/prof_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the bench command      -> gpurun_out/<tag>/kernel_trace.txt (+ the bench line)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)  -> gpurun_out/<tag>/gemm_hbm_traffic.json
#   3. rocprofv3 --pmc SQ counters (own pass)                     -> gpurun_out/<tag>/{gemm,attn}_sq_counters.json
# Counters are collected in their own runs with --kernel-trace only (no sys/hip tracing), as the guide prescribes.
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
export PYTHONPATH=$ROOT TMPDIR=/tmp
rm -rf $OUT && mkdir -p $OUT
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pruned"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err)
python tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_trace.txt
head -12 $OUT/kernel_trace.txt
PMCBENCH="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pruned --no-rollouts --no-profile --no-720p --no-comm-probe"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o p -- $PMCBENCH > /dev/null 2> $OUT/pmc_f.err)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o p -- $PMCBENCH > /dev/null 2> $OUT/pmc_w.err)
python tools/pmc_traffic.py $(find $OUT/pmc_f -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) gemm_bf16_a9 $OUT/gemm_hbm_traffic.json | head -8
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- $PMCBENCH > /dev/null 2> $OUT/pmc_sq.err)
SQ=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
python tools/pmc_sq.py $SQ gemm_bf16_a9 $OUT/gemm_sq_counters.json | grep -E "mfma_util|_frac|launches" | head -40
python tools/pmc_sq.py $SQ clip_attn $OUT/attn_sq_counters.json | grep -E "mfma_util|_frac|launches" | head -12
rm -rf $OUT/kt $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq      # raw traces are large; the summaries above are what gets committed
ls -la $OUT

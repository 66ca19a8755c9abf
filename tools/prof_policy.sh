#!/bin/bash
# rocprofv3 kernel-trace of the policy step (run on the GPU box from the repo root): tools/prof_policy.sh [steps] [fp32|bf16x3] [dp]
set -e
ROOT=$(pwd)
export PYTHONPATH=$ROOT TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pol && mkdir -p $ROOT/gpurun_out/pol
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/pol -- python $ROOT/tools/bench_policy.py ${1:-50} ${2:-fp32} $3 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/pol -name "*.db" | head -1)

#!/bin/bash
# rocprofv3 kernel-trace summary of any command (run on the GPU box from the repo root): tools/prof_cmd.sh <tag> <cmd...>
TAG=$1; shift
ROOT=$(pwd)
export PYTHONPATH=$ROOT TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/$TAG && mkdir -p $ROOT/gpurun_out/$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG/kt -- "$@" > $ROOT/gpurun_out/$TAG/stdout.txt 2> $ROOT/gpurun_out/$TAG/stderr.txt)
python tools/rocpd_summary.py $(find gpurun_out/$TAG/kt -name "*.db" | head -1) > gpurun_out/$TAG/kernel_trace.txt
rm -rf gpurun_out/$TAG/kt
head -30 gpurun_out/$TAG/kernel_trace.txt | cut -c1-200

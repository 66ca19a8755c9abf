#!/bin/bash
# end-of-round evidence set (run on the GPU box from the repo root): full -m gpu suite, smoke, default bench line, the configs[4]
# per-GPU share, kernel traces of the policy step (B = 4 fused; the reference's B = 1 x 2 window sequential and coalesced),
# then tools/prof_round.sh (kernel trace + FETCH/WRITE + SQ passes of the bench command).   tools/gpu_final.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r6z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 python bench.py --frames 4096 --rollout-cfg 1,4096,16,16 --no-cpu-baseline --no-pruned --no-720p > $OUT/bench_T4096.json 2> $OUT/bench_T4096.err
bash tools/prof_policy.sh 50 fp32 > $OUT/policy_trace.txt 2>&1
bash tools/prof_policy.sh 50 fp32 dp > $OUT/policy_trace_dp.txt 2>&1
bash tools/prof_policy.sh 50 fp32 dpc > $OUT/policy_trace_dpc.txt 2>&1
cat $OUT/pytest_gpu.log $OUT/smoke.log; tail -3 $OUT/bench.err
bash tools/prof_round.sh ${TAG}_prof > $OUT/prof_round.log 2>&1; tail -30 $OUT/prof_round.log

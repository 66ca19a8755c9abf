#!/bin/bash
# SQ counters of the policy step's kernels (run on the GPU box from the repo root): tools/pmc_policy.sh [dp]
#   -> gpurun_out/polpmc/policy_sq_counters.json (MFMA busy, issue-wait and wait fractions, LDS bank conflicts per kernel)
ROOT=$(pwd)
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/polpmc; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- python $ROOT/tools/bench_policy.py 20 fp32 $1 > /dev/null 2> $OUT/sq.err)
python tools/pmc_sq.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) "anonymous" $OUT/policy_sq_counters.json > /dev/null
python - <<PY
import json
j=json.load(open("$OUT/policy_sq_counters.json"))
for k,r in sorted(j["kernels"].items(), key=lambda kv: -kv[1].get("wall_cycles",0)*kv[1]["launches"]):
    print("%-62s n=%4d wall %8.0f cyc  mfma %.3f  issue-wait %.2f  wait %.2f  active %.2f  ldsconf %.3f" % (k, r["launches"], r.get("wall_cycles",0), r.get("mfma_util",0), r.get("SQ_WAIT_INST_ANY_frac_of_wave_cycles",0), r.get("SQ_WAIT_ANY_frac_of_wave_cycles",0), r.get("SQ_ACTIVE_INST_ANY_frac_of_wave_cycles",0), r.get("lds_bank_conflict_frac",0)))
PY
rm -rf $OUT/sq

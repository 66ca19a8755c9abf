#!/bin/bash
# FETCH_SIZE per GEMM form for several library builds (tmp_ab/lib_<name>.so), same box:  bash tools/ab_fetch.sh base ntA ...
ROOT=$(pwd)
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/abf; mkdir -p $OUT
PMCBENCH="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pruned --no-rollouts --no-profile --no-720p --no-comm-probe"
for v in "$@"; do
  cp tmp_ab/lib_$v.so tspo_amd/libtspo_hip.so
  rm -rf $OUT/f
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- $PMCBENCH > /dev/null 2> $OUT/f_$v.err)
  F=$(find $OUT/f -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $F gemm_bf16_a9 $OUT/fetch_$v.json > /dev/null
  python - <<PY
import json; j=json.load(open("$OUT/fetch_$v.json"))
print("$v", "avg fetch x2 MB/launch %.0f" % (j["fetch_bytes_per_launch_avg_x2"]/1e6), " ".join("%s:%.0f" % (r["kernel"].split("kernel<")[1][:1], r["fetch_MB_per_launch_x2"]) for r in j["rows"]))
PY
done
rm -rf $OUT/f

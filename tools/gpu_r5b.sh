#!/bin/bash
# round 5, second GPU pass: coalesced micro-steps (tests + the dp-path bench figures)
OUT=gpurun_out/r5b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_api.py tests/test_gpu_flow.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_dp.log
timeout 900 python bench.py --no-cpu-baseline --no-pruned --no-720p > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
l=[json.loads(x) for x in open('gpurun_out/r5b/bench.json') if x.startswith('{')][0]
print(json.dumps(l["rollouts_dp_path"], indent=1))
print("value", l["value"], "rollouts", l["rollouts_per_s"], l["rollouts_roofline"]["launches_per_step"], l["roofline"]["frac"], l["breakdown_ms"] if "breakdown_ms" in l else "")
PY

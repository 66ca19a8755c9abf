#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on launches of known byte count (run on the GPU box from the repo root):
#   bash tools/fetch_calibrate.sh   -> gpurun_out/cal/fetch_calibration.json
ROOT=$(pwd)
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/cal; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- python $ROOT/tools/fetch_calibrate.py > $OUT/f.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- python $ROOT/tools/fetch_calibrate.py > $OUT/w.log 2>&1)
python tools/fetch_calibrate.py --summarise $(find $OUT/f -name "*counter_collection.csv" | head -1) $(find $OUT/w -name "*counter_collection.csv" | head -1) $OUT/fetch_calibration.json
tail -3 $OUT/f.log
rm -rf $OUT/f $OUT/w

#!/bin/bash
# tools/reproduce_published_idx.sh <dataset> <TSPO-0.4B dir> <video dir> <reference checkout> [extra args]
# Local weights + videos -> FrameIdGenerator -> frame-index JSON -> tools/compare_frame_idx.py against the reference's published
# evaluation/jsons_idx/TSPO_<dataset>_frameIdx.json.  See tools/reproduce_published_idx.py (needs a GPU, decord, the checkpoint).
set -e
cd "$(dirname "$0")/.."
exec python tools/reproduce_published_idx.py --dataset "$1" --weights "$2" --videos "$3" --reference "$4" "${@:5}"

#!/usr/bin/env python
"""Determinism screen of the whole encode at configs[1] size: the same 1024-frame video encoded repeatedly must give
bitwise identical features (the GEMM / attention kernels have no atomics and a fixed reduction order).
    python tools/encode_repeat.py [frames] [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
from tspo_amd.pipeline import FrameScorer
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DEV = torch.device("cuda", 0)
c = bench.CLIP_L14
clipw = ops.ClipVitWeights(bench.random_clip_state(c, DEV), c, DEV)
flat = bench.flat_from_state(bench.random_selector_state(768, DEV), 768, DEV)
scorer = FrameScorer(clipw, flat)
gen = torch.Generator(device=DEV).manual_seed(1234)
px = torch.randint(0, 256, (1, T, 3, 224, 224), generator=gen, device=DEV, dtype=torch.uint8)
first = None
for i in range(reps):
    f = scorer.encode(px)
    torch.cuda.synchronize()
    print(f"run {i}: finite {bool(torch.isfinite(f).all())} absmax {f.float().abs().max().item():.4f}", end="")
    if first is None:
        first = f.clone(); print()
    else:
        d = (f.float() - first.float()).abs()
        bad = (d.amax(-1) > 0).nonzero()
        print(f"  frames differing from run 0: {bad.shape[0]}  max diff {d.max().item():.4g}  first few {bad[:8, -1].tolist()}")

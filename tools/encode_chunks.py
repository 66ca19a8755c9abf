#!/usr/bin/env python
"""Does the encoder gain from a working set that fits the 256 MB memory-side cache?  One 1024-frame video encoded as ONE call
(activations 0.5-2.2 GB per tensor: every GEMM / attention operand streams from HBM) against the same frames in chunks of
512 / 256 / 128 / 64 frames (at 64-128 frames the q|k|v and MLP-hidden tensors are 100-270 MB).  Same kernels, same FLOPs.
    python tools/encode_chunks.py [frames] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
DEV = torch.device("cuda", 0)
c = bench.CLIP_L14
clipw = ops.ClipVitWeights(bench.random_clip_state(c, DEV), c, DEV)
gen = torch.Generator(device=DEV).manual_seed(1234)
px = torch.randint(0, 256, (T, 3, 224, 224), generator=gen, device=DEV, dtype=torch.uint8)
ref = ops.clip_vit_forward(clipw, px)
for chunk in (T, 512, 256, 255, 128, 127, 64):
    if chunk > T:
        continue
    def run():
        return torch.cat([ops.clip_vit_forward(clipw, px[i:i + chunk]) for i in range(0, T, chunk)])
    out = run(); out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    same = bool(torch.equal(out, ref))
    print(f"chunk {chunk:5d} frames: {dt * 1e3:8.2f} ms per {T} frames = {T / dt:8.1f} frames/s   bitwise == one call: {same}", flush=True)

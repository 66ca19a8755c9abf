#!/usr/bin/env python
"""The frames/s a caller sees when it hands over HOST frames (the boundary takes device pointers; DESIGN 5 'PCIe note'): one
1024-frame video of uint8 [T,3,224,224] pixels in pinned host memory -> H2D copy -> the same scoring step, (a) copy then step on one
stream, (b) the copy of video i+1 on a second stream under the step of video i.   python tools/pcie_inclusive.py [frames] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
from tspo_amd.pipeline import FrameScorer
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
c = bench.CLIP_L14
scorer = FrameScorer(ops.ClipVitWeights(bench.random_clip_state(c, dev), c, dev), bench.flat_from_state(bench.random_selector_state(768, dev), 768, dev))
host = [torch.randint(0, 256, (1, T, 3, 224, 224), dtype=torch.uint8).pin_memory() for _ in range(2)]
txt = torch.randn(1, 1, 768, device=dev)
dbuf = [torch.empty((1, T, 3, 224, 224), dtype=torch.uint8, device=dev) for _ in range(2)]


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_copy = timed(lambda: dbuf[0].copy_(host[0], non_blocking=True), reps)
t_step = timed(lambda: scorer(dbuf[0], txt, 32), reps)
t_serial = timed(lambda: (dbuf[0].copy_(host[0], non_blocking=True), scorer(dbuf[0], txt, 32)), reps)
side = torch.cuda.Stream()
state = {"i": 0}
ev = [torch.cuda.Event(), torch.cuda.Event()]


def overlapped():
    i = state["i"] & 1
    with torch.cuda.stream(side):                       # next video's pixels on the copy stream ...
        side.wait_event(ev[i ^ 1]) if state["i"] else None
        dbuf[i ^ 1].copy_(host[i ^ 1], non_blocking=True)
        cp = torch.cuda.Event(); cp.record(side)
    scorer(dbuf[i], txt, 32)                            # ... under this video's step
    ev[i].record()
    torch.cuda.current_stream().wait_event(cp)
    state["i"] += 1


t_ov = timed(overlapped, reps)
mb = host[0].numel() / 1e6
print(f"H2D copy of {mb:.0f} MB pinned: {t_copy * 1e3:.2f} ms = {mb / 1e3 / t_copy:.1f} GB/s")
print(f"step, pixels resident (bench.py's value): {t_step * 1e3:.2f} ms = {T / t_step:.1f} frames/s")
print(f"copy then step, one stream:               {t_serial * 1e3:.2f} ms = {T / t_serial:.1f} frames/s (PCIe-inclusive, serial)")
print(f"copy of video i+1 under the step of i:    {t_ov * 1e3:.2f} ms = {T / t_ov:.1f} frames/s (PCIe-inclusive, overlapped on a second stream)")

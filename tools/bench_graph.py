"""The scoring step (configs[1]: 1024 frames -> indices) eager vs replayed as ONE captured graph (torch.cuda.CUDAGraph = hipGraph), same
box, alternating: the whole path is capturable (every launch on the current stream, no host synchronisation, no allocation by the
library) - and replay changes nothing (+-0.1 %, round 5): the step is GPU-bound, its ~220 launches leave no gaps to close.
    python tools/bench_graph.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
from tspo_amd.pipeline import FrameScorer
dev = "cuda"
c = bench.CLIP_L14
T, k = 1024, 32
clipw = ops.ClipVitWeights(bench.random_clip_state(c, dev), c, dev)
flat = bench.flat_from_state(bench.random_selector_state(768, dev), 768, dev)
scorer = FrameScorer(clipw, flat)
g = torch.Generator(device=dev).manual_seed(1234)
pixels = torch.randint(0, 256, (1, T, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
txt = torch.randn(1, 1, 768, generator=torch.Generator(device=dev).manual_seed(4321), device=dev)
out = {}
def step():
    out["idx"], out["scores"], _ = scorer(pixels, txt, k)
for _ in range(2): step()
torch.cuda.synchronize()
ref_idx = out["idx"].clone()
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(gr):
    step()
torch.cuda.synchronize()
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for r in range(3):
    e = timeit(step); gms = timeit(gr.replay)
    print(f"round {r}: eager {e:.3f} ms  graph {gms:.3f} ms  ({(e / gms - 1) * 100:+.2f} %)")
print("same indices:", torch.equal(out["idx"], ref_idx))

#!/usr/bin/env python
"""Policy-step micro-benchmark (configs[2]: B=4, T=512, G=8, k=16) for profiling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tspo_amd import ops
from tspo_amd.pipeline import PolicyTrainer
dev = "cuda"
B, T, G, k, tau = 4, 512, 8, 16, 0.025
gen = torch.Generator(device=dev).manual_seed(99)
feats = torch.randn(B, T, 768, generator=gen, device=dev)
txt = torch.randn(B, 1, 768, generator=gen, device=dev)
clip = ops.clip_scores(txt, feats)
rew = (torch.rand(B, G, generator=gen, device=dev) > 0.5).float() + torch.rand(B, G, generator=gen, device=dev)
flat = bench.flat_from_state(bench.random_selector_state(768, dev), 768, dev)
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
tr = PolicyTrainer(flat, gemm_precision=prec)
for _ in range(5):
    tr.step(feats, txt, clip, lambda i: rew, G, k, tau)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t0 = time.perf_counter()
for _ in range(n):
    tr.step(feats, txt, clip, lambda i: rew, G, k, tau)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"policy step ({prec}): {dt * 1e3:.3f} ms -> {B * G / dt:.0f} rollouts/s")

#!/usr/bin/env python
"""Race screen for a GEMM variant: the same launch repeated back to back must give bitwise identical outputs.
    python tools/gemm_repeat.py variants [repeats]      e.g.  77,73,82 12"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tspo_amd import ops
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [77]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
M = 257 * 1024
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, act, resid in [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False),
                               ("fc2", 1024, 4096, 0, True), ("k640", 1024, 640, 0, False), ("k320r", 1024, 320, 0, True), ("k256r", 512, 256, 0, True)]:
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda") * 0.1
    R = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16) if resid else None
    rows = torch.randint(0, M, (256,), device="cuda")
    rr = A[rows].float() @ W.float().t() + bias
    if act == 1:
        rr = rr * torch.sigmoid(1.702 * rr)
    if resid:
        rr = rr + R[rows].float()
    for v in variants:
        first, bad, worst = None, 0, 0.0
        for i in range(reps):
            out = ops.gemm_bf16(A, W, bias=bias, residual=R, act=act | (v << 8))
            torch.cuda.synchronize()
            err = (out[rows].float() - rr).abs().max().item() / rr.abs().max().item()
            worst = max(worst, err)
            if first is None:
                first = out.clone()
            elif not torch.equal(out, first):
                bad += 1
                d = (out.float() - first.float()).abs()
                nz = d.nonzero()
                print(f"   run {i}: {nz.shape[0]} elements differ, max {d.max().item():.3g}, rows {nz[:, 0].min().item()}..{nz[:, 0].max().item()} cols {nz[:, 1].min().item()}..{nz[:, 1].max().item()}")
        print(f"{name:6s} N={N} K={K} variant {v}: {bad}/{reps - 1} repeats differ; worst relerr vs fp32 rows {worst:.2e}", flush=True)

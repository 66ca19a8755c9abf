#!/usr/bin/env python
"""Micro-benchmark + check of the bf16 MFMA GEMM on the CLIP-L shapes (M = 257*T rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tspo_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
M = 257 * T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True)]
for name, N, K, act, resid in shapes:
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if resid else None
    ref = None
    for v in variants:
        f = lambda: ops.gemm_bf16(A, W, bias=bias, residual=R, act=act | (v << 8))
        out = f()
        torch.cuda.synchronize()
        if ref is None:
            rows = torch.randint(0, M, (512,), device=dev)
            rr = A[rows].float() @ W.float().t() + bias
            if act == 1:
                rr = rr * torch.sigmoid(1.702 * rr)
            if resid:
                rr = rr + R[rows].float()
            ref = (rows, rr)
        err = (out[ref[0]].float() - ref[1]).abs().max().item() / ref[1].abs().max().item()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            f()
        st.record()
        n = 10
        for _ in range(n):
            f()
        en.record()
        torch.cuda.synchronize()
        ms = st.elapsed_time(en) / n
        print(f"{name:4s} N={N:5d} K={K:5d} variant {v}: {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s  relerr {err:.2e}", flush=True)

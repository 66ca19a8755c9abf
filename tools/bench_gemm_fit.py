#!/usr/bin/env python
"""Per-tile overhead vs per-K-step time of the persistent GEMM: time C = A W^T for several K at fixed (M, N) and fit
t_tile = a + b * (K / 64).   python tools/bench_gemm_fit.py [variant]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tspo_amd import ops
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = 257 * 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for N in (1024, 3072):
    for epi in ("bias", "gelu", "resid"):
        pts = []
        for K in (512, 1024, 2048, 4096):
            A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
            bias = torch.randn(N, generator=g, device=dev) * 0.1
            R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if epi == "resid" else None
            f = lambda: ops.gemm_bf16(A, W, bias=bias, residual=R, act=(1 if epi == "gelu" else 0) | (v << 8))
            for _ in range(3):
                f()
            ts = []
            for _ in range(5):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(4):
                    f()
                en.record()
                torch.cuda.synchronize()
                ts.append(st.elapsed_time(en) / 4)
            ms = statistics.median(ts)
            rounds = (M // 256) * (N // 256) / 256.0
            pts.append((K // 64, ms * 1e3 / rounds))
            del A, W, R
        n = len(pts)
        sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        print(f"variant {v} N={N} {epi:5s}: per tile-round us {[round(p[1], 1) for p in pts]} at nk {[p[0] for p in pts]} -> "
              f"overhead a = {a:.1f} us/tile, b = {b:.3f} us/K-step ({2 * 256 * 256 * 64 * 256 / b / 1e6:.0f} TFLOP/s asymptotic)", flush=True)

#!/usr/bin/env python
"""Per-K-step wait + barrier cycles of the 4-wave AGPR GEMM (--dev build, variant 78): every wave accumulates the s_memtime
cycles it spends between 'my LDS traffic is done' and 'everybody passed the barrier', and the length of its K-steps.
    python tools/probe_gemm_barrier.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from tspo_amd import _lib, ops
dev = "cuda"; M = 257 * 1024
l = _lib.lib()
l.tspo_dev_set_debug.argtypes = [C.c_void_p]
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, act, resid in [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True)]:
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if resid else None
    dbg = torch.zeros(256 * 4 * 3, dtype=torch.int64, device=dev)
    l.tspo_dev_set_debug(C.c_void_p(dbg.data_ptr()))
    for _ in range(3):
        ops.gemm_bf16(A, W, bias=bias, residual=R, act=act | (78 << 8))
    torch.cuda.synchronize()
    l.tspo_dev_set_debug(None)
    d = dbg.view(256, 4, 3).cpu().double()
    bar, ks, n = d[..., 0], d[..., 1], d[..., 2]
    print(f"{name}: K-steps per wave {n.mean():.0f}; cycles per K-step (incl. the tile epilogues) {(ks / (n - 1)).mean():.0f}; "
          f"wait+barrier per K-step: mean {(bar / n).mean():.0f}  by wave {[int(v) for v in (bar / n).mean(0)]}  "
          f"min/max over workgroups {(bar / n).mean(1).min():.0f}/{(bar / n).mean(1).max():.0f}", flush=True)

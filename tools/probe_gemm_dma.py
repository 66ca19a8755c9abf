#!/usr/bin/env python
"""Where a K-step of the LDS-DMA GEMM (a9) spends its cycles (--dev build; variant 74 = the production schedule):
s_memtime sums per wave: whole K-step, inside barrier 1 / 2, the vmcnt wait for the next stage, inside barrier 3.
    python tools/probe_gemm_dma.py """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from tspo_amd import _lib, ops
v = int(sys.argv[1]) if len(sys.argv) > 1 else 74
dev = "cuda"; M = 257 * 1024
l = _lib.lib()
l.tspo_dma_set_debug.argtypes = [C.c_void_p]
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, act, resid in [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True)]:
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if resid else None
    dbg = torch.zeros(256 * 4 * 12, dtype=torch.int64, device=dev)
    l.tspo_dma_set_debug(C.c_void_p(dbg.data_ptr()))
    for _ in range(3):
        ops.gemm_bf16(A, W, bias=bias, residual=R, act=act | (v << 8))
    torch.cuda.synchronize()
    l.tspo_dma_set_debug(None)
    d = dbg.view(256, 4, 12).cpu().double()
    n = d[..., 1]
    f = lambda i: (d[..., i] / n).mean().item()
    print(f"{name} variant {v}: K-steps/wave {n.mean():.0f}; cycles per K-step {f(0):.0f} (probes included); barrier1 {f(2):.0f}  barrier2 {f(3):.0f}  "
          f"vmcnt wait {f(4):.0f}  barrier3 {f(5):.0f};  vmcnt wait in a tile's first K-step {(d[..., 6] / d[..., 7]).mean().item():.0f}, second {(d[..., 8] / d[..., 7]).mean().item():.0f}; per tile: {(d[..., 10] / d[..., 11]).mean().item():.0f} cycles, epilogue {(d[..., 9] / d[..., 11]).mean().item():.0f}", flush=True)

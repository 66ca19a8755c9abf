#!/usr/bin/env python
"""Tile-phase probe of the 4-wave AGPR GEMM (--dev build, variant 85): s_memtime stamps at tile start, end of the K-loop,
end of the epilogue and end of the fragment refill, for the first 24 tiles of every workgroup; s_memrealtime (100 MHz,
chip-wide) at every tile start shows how far the 256 workgroups drift from lock-step.
    python tools/probe_gemm_tiles.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from tspo_amd import _lib, ops
dev = "cuda"; M = 257 * 1024
l = _lib.lib()
l.tspo_dev_set_debug.argtypes = [C.c_void_p]
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 85
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, act, resid in [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True)]:
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if resid else None
    dbg = torch.zeros(256 * 24 * 6, dtype=torch.int64, device=dev)
    l.tspo_dev_set_debug(C.c_void_p(dbg.data_ptr()))
    for _ in range(3):
        dbg.zero_()
        ops.gemm_bf16(A, W, bias=bias, residual=R, act=act | (VAR << 8))
    torch.cuda.synchronize()
    l.tspo_dev_set_debug(None)
    d = dbg.view(256, 24, 6).cpu().double()
    ntile = int((d[:, :, 0] > 0).sum(1).min())
    d = d[:, 1:ntile]                                   # skip tile 0 (prologue, cold)
    kl, ep, rf = d[..., 1] - d[..., 0], d[..., 2] - d[..., 1], d[..., 3] - d[..., 2]
    tile = d[:, 1:, 0] - d[:, :-1, 0]
    nk = K // 64
    rt = d[..., 5]                                      # 100 MHz
    spread = (rt - rt.mean(0, keepdim=True))            # per tile index, across workgroups
    print(f"{name} N={N} K={K}: tiles probed/WG {ntile - 1}; cycles per tile {tile.mean():.0f} = K-loop {kl.mean():.0f} "
          f"({kl.mean() / nk:.0f}/K-step) + epilogue {ep.mean():.0f} (min {ep.min():.0f} p90 {ep.flatten().kthvalue(int(0.9 * ep.numel())).values:.0f} max {ep.max():.0f}) "
          f"+ refill {rf.mean():.0f}; tile-start spread across WGs: std {spread.std() * 10:.0f} ns, range {(spread.max() - spread.min()) * 10:.0f} ns; "
          f"tile period {(rt[:, 1:] - rt[:, :-1]).mean() * 10:.0f} ns", flush=True)
    # per-XCD phase: are the workgroups of one XCD in step?
    x = spread.view(32, 8, -1)
    e = ep.view(32, 8, -1).mean((1, 2))
    print(f"      epilogue cycles by position in the XCD (wl 0,4,..,28): {[int(v) for v in e[::4]]}")
    print(f"      within-XCD std {x.std(0).mean() * 10:.0f} ns; epilogue cycles by tile index: {[int(v) for v in ep.mean(0)[:10]]}", flush=True)

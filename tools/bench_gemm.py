#!/usr/bin/env python
"""Micro-benchmark + check of the bf16 MFMA GEMM on the CLIP-L shapes (M = 257*T rows).
Variants are timed interleaved (rotating order) over several rounds; the median per variant is reported
(single back-to-back runs showed an ~8 % position bias).  Variant -1 is the same-box yardstick: the vendor GEMM
(torch.nn.functional.linear -> hipBLASLt) on the same operands, WITHOUT the activation / residual epilogue the
hand-written kernels fuse (tool only; the product never calls it).

    python tools/bench_gemm.py T variants rounds        e.g.  1024 82,77,-1 6
Variants: 77 = production (LDS-DMA operands), 82 = register-staged four-wave kernel, -1 = vendor; 76 / 67 / 75 (three-barrier schedule, vendor positions, no-DMA
ablation) need a `python -m tspo_amd.build --dev` library.
Every variant's full output is also compared with the first variant's (bitwise where the K order is the same)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tspo_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [82, 77, -1]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = 257 * T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True)]
for name, N, K, act, resid in shapes:
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
    if os.environ.get("TSPO_BENCH_FILL") == "zero":      # power probe: same kernels on zero-filled operands (DVFS)
        A.zero_(); W.zero_()
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    bias16 = bias.to(torch.bfloat16)
    R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if resid else None
    rows = torch.randint(0, M, (512,), device=dev)
    rr = A[rows].float() @ W.float().t() + bias
    if act == 1:
        rr = rr * torch.sigmoid(1.702 * rr)
    if resid:
        rr = rr + R[rows].float()

    def mk(v):
        if v == -1:
            return lambda: torch.nn.functional.linear(A, W, bias16)
        return lambda: ops.gemm_bf16(A, W, bias=bias, residual=R, act=act | (v << 8))

    fns = {v: mk(v) for v in variants}
    errs, same = {}, {}
    first = None
    for v, f in fns.items():
        out = f()
        torch.cuda.synchronize()
        if v != -1:
            errs[v] = (out[rows].float() - rr).abs().max().item() / rr.abs().max().item()
            if first is None:
                first = out
            same[v] = bool(torch.equal(out, first))
            if not same[v]:
                same[v] = "maxdiff %.3g" % (out.float() - first.float()).abs().max().item()
        else:
            errs[v], same[v] = float("nan"), "-"
        del out
        for _ in range(2):
            f()
    times = {v: [] for v in variants}
    for r in range(rounds):
        order = variants[r % len(variants):] + variants[:r % len(variants)]
        for v in order:
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(5):
                fns[v]()
            en.record()
            torch.cuda.synchronize()
            times[v].append(st.elapsed_time(en) / 5)
    for v in variants:
        ms = statistics.median(times[v])
        tag = "hipBLASLt (no act/resid)" if v == -1 else f"variant {v:3d}"
        print(f"{name:4s} N={N:5d} K={K:5d} {tag:>24s}: median {ms:7.3f} ms (min {min(times[v]):7.3f})  "
              f"{2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s  relerr {errs[v]:.2e}  same-as-first {same[v]}", flush=True)
    del first

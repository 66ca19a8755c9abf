#!/usr/bin/env python
"""What the partial last round of whole tiles costs, and what the remainder phase (round 5) takes back.

The persistent 256x256 GEMM hands out whole tiles to 256 workgroups.  The encoder's M = 257 * 1024 = 263 168 rows are 1 028
M-tiles: out-proj / fc2 (N = 1024) have 4 112 tiles = 16.06 rounds, QKV 48.19, fc1 64.25.  M = 262 144 (1 024 M-tiles) is an exact
number of rounds for every shape.  This tool times, interleaved on one box,

    variant 83  = whole tiles only (round-4 behaviour: a 17th / 49th / 65th round at 1/16 .. 1/4 occupancy)
    variant 77  = production (whole rounds + 64x64 sub-tiles of the left-over tiles over all workgroups)

at both M, prints the time per launch in units of one full round at M = 262 144 ("tile-times"), and checks that 77 and 83 agree
bit for bit (the remainder phase keeps every element's K order).

    python tools/bench_gemm_tail.py [rounds]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tspo_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 3072, 1024, 0, False), ("out", 1024, 1024, 0, True), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True)]
MS = (262144, 263168)
for name, N, K, act, resid in shapes:
    Mmax = max(MS)
    A = torch.randn(Mmax, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    R = torch.randn(Mmax, N, generator=g, device=dev).to(torch.bfloat16) if resid else None
    cases = [(M, v) for M in MS for v in (83, 77)]

    def mk(M, v):
        a, r = A[:M], (R[:M] if resid else None)
        return lambda: ops.gemm_bf16(a, W, bias=bias, residual=r, act=act | (v << 8))

    fns = {c: mk(*c) for c in cases}
    same = {}
    for M in MS:
        o83, o77 = fns[(M, 83)](), fns[(M, 77)]()
        torch.cuda.synchronize()
        same[M] = bool(torch.equal(o83, o77))
        del o83, o77
    for f in fns.values():
        for _ in range(2):
            f()
    times = {c: [] for c in cases}
    for r in range(rounds):
        order = cases[r % len(cases):] + cases[:r % len(cases)]
        for c in order:
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(5):
                fns[c]()
            en.record()
            torch.cuda.synchronize()
            times[c].append(st.elapsed_time(en) / 5)
    tilesN = N // 256
    full_rounds = 1024 * tilesN // 256
    unit = statistics.median(times[(262144, 83)]) / full_rounds            # one round of whole tiles, this shape, this box
    for M, v in cases:
        ms = statistics.median(times[(M, v)])
        tiles = (M // 256) * tilesN
        print(f"{name:4s} N={N:5d} K={K:5d} M={M:7d} ({tiles:6d} tiles = {tiles / 256:6.2f} rounds) variant {v}: median {ms:7.4f} ms (min {min(times[(M, v)]):7.4f}) "
              f"= {ms / unit:6.2f} tile-times  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s  77==83 bitwise: {same[M]}", flush=True)

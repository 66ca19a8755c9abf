#!/usr/bin/env python
"""Short-burst timing of the production bf16 GEMM on a square problem (default 4096^3, the shape kernel guides quote):
a handful of launches after an idle period, so the chip is not yet power-throttled - to compare with the sustained
numbers of bench.py / tools/bench_gemm.py (M = 263 168 rows, back-to-back for seconds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tspo_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for fill in ("uniform[-1,1)", "zeros"):
    if fill == "zeros":
        A = torch.zeros(n, n, device=dev, dtype=torch.bfloat16); W = torch.zeros(n, n, device=dev, dtype=torch.bfloat16)
    else:
        A = (torch.rand(n, n, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(n, n, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    bias = torch.zeros(n, device=dev)
    for _ in range(2):
        ops.gemm_bf16(A, W, bias=bias)
    torch.cuda.synchronize()
    time.sleep(1.0)                       # let the clocks recover
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(reps):
        ops.gemm_bf16(A, W, bias=bias)
    en.record()
    torch.cuda.synchronize()
    ms = st.elapsed_time(en) / reps
    print(f"{n}^3 bf16, {fill}: {ms:.3f} ms/launch over {reps} launches -> {2.0 * n**3 / ms / 1e9:.0f} TFLOP/s")
    # sustained: 300 launches back to back
    st.record()
    for _ in range(300):
        ops.gemm_bf16(A, W, bias=bias)
    en.record()
    torch.cuda.synchronize()
    ms = st.elapsed_time(en) / 300
    print(f"{n}^3 bf16, {fill}: {ms:.3f} ms/launch over 300 launches -> {2.0 * n**3 / ms / 1e9:.0f} TFLOP/s (sustained)")

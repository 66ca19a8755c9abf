#!/usr/bin/env python
"""Yardstick inspection (tool only; the product never calls the vendor GEMM): run the four CLIP-L GEMM shapes through
torch.nn.functional.linear (-> hipBLASLt) so that `rocprofv3 --kernel-trace` records WHICH Tensile kernel the library picks
for each (the kernel name encodes macro tile, depthU, MFMA shape, wave layout, direct-to-LDS, prefetch depth ...).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o v -- python tools/vendor_gemm_names.py 1024
    python tools/vendor_gemm_names.py --summarise OUT/.../v_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict


def run(T: int) -> None:
    import torch
    M = 257 * T
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
        A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.03).to(torch.bfloat16)
        b = (torch.randn(N, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
        for _ in range(6):
            torch.nn.functional.linear(A, W, b)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            torch.nn.functional.linear(A, W, b)
        en.record()
        torch.cuda.synchronize()
        ms = st.elapsed_time(en) / 10
        print(f"{name} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
        del A, W, b


def summarise(path: str) -> None:
    by = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r.get("Kernel_Name") or r.get("kernel_name") or ""
            if "Cijk" in n or "gemm" in n.lower():
                by[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        print(f"{len(d):4d} launches  median {d[len(d) // 2]:9.1f} us  min {d[0]:9.1f} us\n     {n}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)

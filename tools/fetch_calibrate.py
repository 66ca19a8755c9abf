#!/usr/bin/env python
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on launches whose memory-side byte count is KNOWN, in the production
GEMM's own access pattern (the guide: "calibrate on a known byte count in your own access pattern before trusting an absolute").

Cases (each a distinct kernel name / instantiation, so the counter CSV can be keyed):
  a9<0>  M=262144 N=256  K=1024  bias      one N-tile: every workgroup reads PRIVATE A panels exactly once  -> fetch = A (+ W per XCD)
  a9<1>  M=262144 N=1024 K=1024  gelu      four N-tiles of a panel inside ONE XCD                           -> fetch = A if the L2 shares
  copy   1 GiB bf16 torch elementwise y = 2x (16 B / lane streaming read: the guide's own calibration case)              -> fetch = write = 1 GiB

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT_F -o p -- python tools/fetch_calibrate.py
    cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT_W -o p -- python tools/fetch_calibrate.py
    python tools/fetch_calibrate.py --summarise OUT_F/.../p_counter_collection.csv OUT_W/.../p_counter_collection.csv [out.json]
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

M, K = 262144, 1024
CASES = {  # kernel-name substring -> (label, known fetch bytes, known write bytes)
    "gemm_bf16_a9_kernel<0": ("a9 bias, N=256: private A panels", M * K * 2 + 8 * 256 * K * 2, M * 256 * 2),
    "gemm_bf16_a9_kernel<1": ("a9 gelu, N=1024: 4 N-tiles of a panel in one XCD", M * K * 2 + 8 * 1024 * K * 2, M * 1024 * 2),
    "copy": ("torch elementwise y = 2x over 1 GiB (16 B / lane stream)", 1 << 30, 1 << 30),
}


def run() -> None:
    import torch
    from tspo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    for N, act in ((256, 0), (1024, 1)):
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.03).to(torch.bfloat16)
        b = torch.randn(N, generator=g, device="cuda") * 0.1
        for _ in range(3):
            ops.gemm_bf16(A, W, bias=b, act=act)
        torch.cuda.synchronize()
    x = torch.empty(1 << 29, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        torch.mul(x, 2.0, out=y)   # a kernel (a same-dtype copy_ would be a DMA-engine memcpy, invisible to the counters)
    torch.cuda.synchronize()


def summarise(fetch_csv: str, write_csv: str, out: str | None) -> None:
    def load(path, counter):
        agg = defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
        return agg
    fa, wa = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    rows = []
    for sub, (label, kf, kw) in CASES.items():
        for name in fa:
            if sub not in name or (sub == "copy" and "elementwise" not in name):
                continue
            fv, wv = fa[name], wa.get(name, [])
            if sub == "copy":   # only the 1 GiB copies (the fills are other launches of the same family)
                fv = [v for v in fv if v > 0.2 * kf]
                wv = [v for v in wv if v > 0.5 * kw]
            if not fv:
                continue
            f_avg, w_avg = sum(fv) / len(fv), (sum(wv) / len(wv) if wv else float("nan"))
            rows.append({"case": label, "kernel": name[:80], "launches": len(fv), "known_fetch_MB": kf / 1e6,
                         "FETCH_SIZE_MB_raw": f_avg / 1e6, "fetch_raw_over_known": f_avg / kf,
                         "known_write_MB": kw / 1e6, "WRITE_SIZE_MB_raw": w_avg / 1e6, "write_raw_over_known": w_avg / kw})
    txt = json.dumps({"rows": rows}, indent=1)
    print(txt)
    if out:
        open(out, "w").write(txt)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        run()

#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes because the TCC
block has 4 slots).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> doubled; WRITE_SIZE is
reported raw.  Both rules were checked on launches of known byte count in the production
GEMM's own access pattern (tools/fetch_calibrate.sh, profiles/r4_g_fetch_calibration_and_cache_policy.txt: FETCH_SIZE 0.513, WRITE_SIZE 1.001).
Round 6: the summary is BOUND to the binary it was taken on (`library`: sha256 of tspo_amd/libtspo_hip.so and of its sources,
tspo_amd.build.lib_identity) - bench.py only quotes a traffic file whose identity matches the library it loaded - and every
form is put next to ITS algorithmic bytes (`forms`: operands read once, output written once, for the frame count given).
Usage: pmc_traffic.py fetch.csv write.csv kernel_substring [out.json] [n_frames]"""
import csv, json, os, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def form_bytes(n_frames, C=1024, mlp=4096, S=257, patch_k=640, layers=24):
    """Algorithmic bytes (reads, writes) of the CLIP-L/14 encoder's GEMM forms at n_frames, keyed by the kernel's epilogue template
    argument (csrc/gemm_bf16.h): 8 = GE_BIAS_LN_HM, q|k|v (LayerNorm-folded, head-major out); 6 = GE_GELU_LN, fc1; 7 = GE_RESID_ST,
    out-proj of every layer and fc2 of every layer but the last (one kernel name for both: the launch-weighted mean is what the
    counters average too); 2 = GE_RESID, fc2 of the LAST layer (no statistics behind it); 4 = GE_PATCH, the patch embedding.
    (The fp32-out projection, form 3, is a small problem and does not run on this kernel.)"""
    M = S * n_frames
    qkv = (2 * (M * C + 3 * C * C), 2 * M * 3 * C)
    fc1 = (2 * (M * C + mlp * C), 2 * M * mlp)
    out = (2 * (M * C + C * C + M * C), 2 * M * C)
    fc2 = (2 * (M * mlp + mlp * C + M * C), 2 * M * C)
    patch = (2 * ((M - n_frames) * patch_k + C * patch_k), 2 * (M - n_frames) * C)
    n7 = 2 * layers - 1
    rs = tuple((layers * a + (layers - 1) * b) / n7 for a, b in zip(out, fc2))
    return {"8": ("qkv", qkv), "6": ("fc1", fc1), "7": (f"out-proj x{layers} + fc2 x{layers - 1} (launch-weighted mean)", rs),
            "2": ("fc2 of the last layer", fc2), "4": ("patch embedding", patch)}


def forms_of(rows, n_frames):
    fb, forms = form_bytes(n_frames), []
    for r in rows:
        if r.get("form") in fb:
            nm, (ar, aw) = fb[r["form"]]
            rd, wr = r["fetch_MB_per_launch_x2"] * 1e6, r["write_MB_per_launch_raw"] * 1e6
            forms.append({"form": r["form"], "gemm": nm, "launches": r["launches"], "read_GB": round(rd / 1e9, 3), "alg_read_GB": round(ar / 1e9, 3),
                          "read_ratio": round(rd / ar, 2), "write_GB": round(wr / 1e9, 3), "alg_write_GB": round(aw / 1e9, 3),
                          "total_ratio": round((rd + wr) / (ar + aw), 2)})
    return forms


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg


def main():
    if sys.argv[1] == "--recompute-forms":      # the per-form table again from a summary's own rows (no counters are touched)
        tj = json.load(open(sys.argv[2]))
        tj["forms"] = forms_of(tj["rows"], tj.get("n_frames", 1024))
        open(sys.argv[3], "w").write(json.dumps(tj, indent=1))
        print(json.dumps(tj["forms"], indent=1))
        return
    f, w, sub = sys.argv[1], sys.argv[2], sys.argv[3]
    fa, wa = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    rows, tot_f, tot_w, n = [], 0.0, 0.0, 0
    for key in sorted(fa):
        if sub not in key[0]:
            continue
        fv, wv = fa[key], wa.get(key, [0.0])
        rows.append({"kernel": key[0][:70], "grid": key[1], "launches": len(fv), "form": key[0].split("gemm_bf16_a9_kernel<")[-1].split(",")[0].strip(),
                     "fetch_MB_per_launch_raw": sum(fv) / len(fv) * 1024 / 1e6,
                     "fetch_MB_per_launch_x2": 2 * sum(fv) / len(fv) * 1024 / 1e6,
                     "write_MB_per_launch_raw": sum(wv) / len(wv) * 1024 / 1e6})
        tot_f += 2 * sum(fv) * 1024
        tot_w += sum(wv) * 1024
        n += len(fv)
    n_frames = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
    forms = forms_of(rows, n_frames)
    try:
        from tspo_amd.build import lib_identity
        ident = lib_identity()
    except Exception as e:
        ident = {"error": f"{type(e).__name__}: {e}"}
    out = {"kernel_filter": sub, "launches": n, "library": ident, "n_frames": n_frames, "forms": forms, "hbm_bytes_per_launch_avg": (tot_f + tot_w) / max(n, 1),
           "fetch_bytes_per_launch_avg_x2": tot_f / max(n, 1), "write_bytes_per_launch_avg_raw": tot_w / max(n, 1),
           "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 wide-read undercount), WRITE_SIZE KiB x1024 raw", "rows": rows}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt)


if __name__ == "__main__":
    main()

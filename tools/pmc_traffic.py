#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes because the TCC
block has 4 slots).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> doubled; WRITE_SIZE is
reported raw.  Both rules were checked on launches of known byte count in the production
GEMM's own access pattern (tools/fetch_calibrate.sh, profiles/r4_g_fetch_calibration_and_cache_policy.txt: FETCH_SIZE 0.513, WRITE_SIZE 1.001).  Usage: pmc_traffic.py fetch.csv write.csv kernel_substring [out.json]"""
import csv, json, sys, collections


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg


def main():
    f, w, sub = sys.argv[1], sys.argv[2], sys.argv[3]
    fa, wa = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    rows, tot_f, tot_w, n = [], 0.0, 0.0, 0
    for key in sorted(fa):
        if sub not in key[0]:
            continue
        fv, wv = fa[key], wa.get(key, [0.0])
        rows.append({"kernel": key[0][:70], "grid": key[1], "launches": len(fv),
                     "fetch_MB_per_launch_raw": sum(fv) / len(fv) * 1024 / 1e6,
                     "fetch_MB_per_launch_x2": 2 * sum(fv) / len(fv) * 1024 / 1e6,
                     "write_MB_per_launch_raw": sum(wv) / len(wv) * 1024 / 1e6})
        tot_f += 2 * sum(fv) * 1024
        tot_w += sum(wv) * 1024
        n += len(fv)
    out = {"kernel_filter": sub, "launches": n, "hbm_bytes_per_launch_avg": (tot_f + tot_w) / max(n, 1),
           "fetch_bytes_per_launch_avg_x2": tot_f / max(n, 1), "write_bytes_per_launch_avg_raw": tot_w / max(n, 1),
           "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 wide-read undercount), WRITE_SIZE KiB x1024 raw", "rows": rows}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt)


if __name__ == "__main__":
    main()

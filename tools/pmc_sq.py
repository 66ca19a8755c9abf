#!/usr/bin/env python
"""Aggregate one rocprofv3 --pmc pass (counter_collection.csv) per kernel: mean of every counter over the launches whose
name contains `kernel_substring`, plus derived ratios (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs... reported raw + ratios).
Usage: pmc_sq.py counter_collection.csv kernel_substring [out.json]"""
import collections, csv, json, sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"kernel_filter": sub, "kernels": {}}
    for k, cs in agg.items():
        row = {c: sum(v) / len(v) for c, v in cs.items()}
        row["launches"] = len(next(iter(cs.values())))
        wc = row.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in row:
                    row[c + "_frac_of_wave_cycles"] = row[c] / wc
        if "SQ_LDS_BANK_CONFLICT" in row and row.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_frac"] = row["SQ_LDS_BANK_CONFLICT"] / row["SQ_LDS_IDX_ACTIVE"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("GRBM_GUI_ACTIVE"):
            # busy cycles (16 per v_mfma_f32_16x16x32_bf16) are summed over the 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE
            # is summed over the 8 XCDs -> wall cycles of the launch = GRBM_GUI_ACTIVE / 8
            row["wall_cycles"] = row["GRBM_GUI_ACTIVE"] / 8.0
            row["mfma_util"] = (row["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / row["wall_cycles"]
        out["kernels"][k] = row
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max, optionally
split by grid size.  Usage: python tools/rocpd_summary.py results.db [--by-grid] > profiles/xxx.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    key = "name, grid_x, grid_y" if by_grid else "name"
    rows = db.execute(f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      f"max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by {key} "
                      f"order by sum(duration) desc").fetchall()
    total = sum(r[-7] if not by_grid else r[4] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {sys.argv[1]}  (durations in us)")
    hdr = ("kernel", "grid", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "lds")
    print("%-72s %-14s %6s %12s %10s %10s %10s %6s %5s %5s %6s" % hdr)
    for r in rows:
        if by_grid:
            name, gx, gy, n, tot, avg, mn, mx, vg, ag, lds = r
            grid = f"{gx}x{gy}"
        else:
            name, n, tot, avg, mn, mx, vg, ag, lds = r
            grid = "-"
        short = name if len(name) <= 72 else name[:69] + "..."
        print("%-72s %-14s %6d %12.1f %10.1f %10.1f %10.1f %6.2f %5s %5s %6s" %
              (short, grid, n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, lds))


if __name__ == "__main__":
    main()

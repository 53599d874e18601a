#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel CSV kept under profiles/.

usage: tools/rocpd_summary.py gpurun_out/prof_x/x_results.db profiles/rNN_name_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name if len(name) < 160 else name[:157] + "...", calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
    print(f"{len(rows)} kernels -> {out}")
    # per (kernel, grid) averages for the rgm GEMMs: one kernel template serves several layer shapes
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
        dur = "duration" if "duration" in cols else "(end - start)"
        if gx:
            q = f"select name, {gx}, count(*), avg({dur}) from kernels where name like '%gemm%' group by name, {gx} order by 3 desc"
            for name, g, n, avg in cur.execute(q):
                print(f"  {name[:60]:60s} grid_x={g:<8} calls={n:<6} avg={avg / 1e3:9.2f} us")
        else:
            print("  (kernels view columns:", cols, ")")
    except sqlite3.Error as e:
        print("  per-grid breakdown unavailable:", e)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

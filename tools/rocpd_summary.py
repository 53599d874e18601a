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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

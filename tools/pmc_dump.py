#!/usr/bin/env python3
"""Print per-kernel averages of every counter in a rocprofv3 --pmc rocpd database: tools/pmc_dump.py x.db [name-filter]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else "rgm::"
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for n, c, v, k in rows:
    if flt in n:
        print(f"{n[:70]:70s} {c:28s} {v:16.1f}  (n={k})")

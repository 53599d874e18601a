#!/usr/bin/env python3
"""Join the rocpd databases of tools/pmc_round.sh (one --kernel-trace pass + three --pmc passes of the SAME bench workload) into one
per-kernel table: share of the GPU time, average duration, MFMA-busy fraction, HBM bytes per launch and GB/s.

usage: pmc_round_summary.py <workload> <trace dir> <FETCH_SIZE dir> <WRITE_SIZE dir> <SQ/GRBM dir>

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE tallies a 128-byte request as 64 (x 2 on gfx950);
SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE = cycles summed over the 8 XCDs."""
import glob
import os
import re
import sqlite3
import sys

N_SIMD, N_XCD = 1024, 8


def db_of(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        raise SystemExit(f"no rocpd database under {d}")
    return sqlite3.connect(dbs[0]).cursor()


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rgm::", "")
    name = re.sub(r"\(.*$", "", name)                       # drop the argument list
    return name[:70]


def counters(d):
    out = {}
    for name, cname, v, n in db_of(d).execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(short(name), {})[cname] = (v, n)
    return out


def main():
    w, trace, dfetch, dwrite, dsq = sys.argv[1:6]
    cur = db_of(trace)
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    dur = "duration" if "duration" in cols else "(end - start)"
    rows = list(cur.execute(f"select name, count(*), sum({dur}), avg({dur}) from kernels group by name"))
    # shares over the library's own kernels: the run also builds synthetic weights (torch elementwise kernels, device copies)
    rows = [r for r in rows if "rgm::" in r[0]]
    total = sum(r[2] for r in rows)
    fetch, write, sq = counters(dfetch), counters(dwrite), counters(dsq)
    print(f"# workload {w}: in-situ kernels with >= 1.5 % of the library's GPU time (rocprofv3 --kernel-trace; counters from separate --pmc passes of the same command)")
    print(f"# {'kernel':70s} {'share':>6s} {'calls':>6s} {'avg us':>8s} {'MFMA busy':>9s} {'HBM MB/launch':>13s} {'HBM GB/s':>9s} {'fetch MB':>9s} {'write MB':>9s}")
    for name, calls, tot, avg in sorted(rows, key=lambda r: -r[2]):
        if tot < 0.015 * total:
            continue
        k = short(name)
        f = fetch.get(k, {}).get("FETCH_SIZE", (None, 0))[0]
        wr = write.get(k, {}).get("WRITE_SIZE", (None, 0))[0]
        s = sq.get(k, {})
        busy = None
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s and s["GRBM_GUI_ACTIVE"][0] > 0:
            busy = s["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (s["GRBM_GUI_ACTIVE"][0] / N_XCD * N_SIMD)
        fb = 2.0 * f * 1024 if f is not None else None
        wb = wr * 1024 if wr is not None else None
        hbm = (fb or 0) + (wb or 0) if (fb is not None or wb is not None) else None
        print(f"  {k:70s} {100 * tot / total:5.1f}% {calls:6d} {avg / 1e3:8.2f} "
              f"{('%8.1f%%' % (100 * busy)) if busy is not None else '        -':>9s} "
              f"{('%13.1f' % (hbm / 1e6)) if hbm is not None else '            -'} "
              f"{('%9.0f' % (hbm / (avg * 1e-9) / 1e9)) if hbm is not None else '        -'} "
              f"{('%9.1f' % (fb / 1e6)) if fb is not None else '        -'} {('%9.1f' % (wb / 1e6)) if wb is not None else '        -'}")


if __name__ == "__main__":
    main()

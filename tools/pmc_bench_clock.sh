#!/bin/bash
# effective shader clock per kernel inside the bench: GRBM_GUI_ACTIVE (per XCD) / kernel duration
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcC
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmcC -- python $ROOT/bench.py --steps 6 --warmup 2 "$@" > /tmp/pmcC.log 2>&1
db=$(find /tmp/pmcC -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print(cols)
q = "select kernel_name, grid_size_x, count(*), avg(value), avg(end - start) from counters_collection where counter_name='GRBM_GUI_ACTIVE' and kernel_name like '%gemm2%' group by kernel_name, grid_size_x" if "grid_size_x" in cols else None
if q is None:
    gx = [c for c in cols if "grid" in c]
    print("grid cols:", gx)
    q = f"select kernel_name, {gx[0] if gx else 0}, count(*), avg(value), avg(end - start) from counters_collection where counter_name='GRBM_GUI_ACTIVE' and kernel_name like '%gemm2%' group by kernel_name, {gx[0] if gx else 0}"
for name, g, n, v, dur in cur.execute(q):
    print(f"{name[:50]:50s} grid={g} n={n} cycles/XCD={v/8:.0f} dur={dur/1e3:.1f}us clock={v/8/dur:.3f} GHz")
PY

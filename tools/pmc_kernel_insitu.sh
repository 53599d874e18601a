#!/bin/bash
# In-situ PMC of every rgm kernel of a bench workload: tools/pmc_kernel_insitu.sh "<counters>" <bench args...>  (GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C="$1"; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmc_insitu_$$
rm -rf $d
timeout 600 rocprofv3 --pmc $C --kernel-trace -d $d -- python $ROOT/bench.py --traffic-child --steps 1 --warmup 1 "$@" > /dev/null 2>&1
db=$(find $d -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, cname, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%rgm::%' group by kernel_name, counter_name order by 4 desc limit 24"):
    print(f"{name[:96]:96s} {cname:14s} {v:14.1f} (n={n})")
PY

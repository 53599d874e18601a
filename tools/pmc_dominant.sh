#!/bin/bash
# PMC passes of the dominant C2 kernel on the shape it runs most (qkv: M=4096 N=3456 K=1152, tile 143 = gemm2_kernel<128,128,2,2,0,2,0,3>):
# FETCH_SIZE / WRITE_SIZE / L2 hits / matrix-pipe busy in SEPARATE --pmc runs (kernel-trace only).  Run ON the GPU box:
#   tools/pmc_dominant.sh [tag]      -> gpurun_out/<tag>_pmc_qkv_tile143.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_c2}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc_qkv_tile143.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  d=/tmp/pmcq_${TAG}_$(echo $c | cut -c1-10 | tr ' ' '_')
  rm -rf $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -- python $ROOT/tools/gemm_one.py 4096 3456 1152 143 8 > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  python - "$db" "$c" <<'PY' >> $OUT/${TAG}_pmc_qkv_tile143.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
t = "counters_collection" if "counters_collection" in tabs else [x for x in tabs if "counter" in x.lower()][0]
for name, cname, n, v in cur.execute(f"select kernel_name, counter_name, count(*), avg(value) from {t} where kernel_name like '%gemm2_kernel%' group by kernel_name, counter_name"):
    print(f"{name[:70]:70s} {cname:28s} {v:16.1f}  (n={n})")
PY
done
cat $OUT/${TAG}_pmc_qkv_tile143.txt

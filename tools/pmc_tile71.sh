#!/bin/bash
# PMC passes of the dominant C2 kernel, gemm2_kernel<256,256,2,2,0,2,0,5> (tile 71), on its three shapes of the step -- qkv (M=4096 N=3456
# K=1152), fc1's 16 x 16 tiles (N=4096) and one K slice batch of fc2 (through the heuristic: tile 0) -- in SEPARATE --pmc runs
# (kernel-trace only).  Run ON the GPU box:   tools/pmc_tile71.sh [tag]   -> gpurun_out/<tag>_pmc_tile71.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_c2}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
F=$OUT/${TAG}_pmc_tile71.txt
: > $F
for shape in "4096 3456 1152 171" "4096 4096 1152 171" "4096 1152 4608 100"; do
  echo "== gemm_one.py $shape" >> $F
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA TA_BUSY_avr"; do
    d=/tmp/pmc71_$(echo $c | cut -c1-10 | tr ' ' '_')
    rm -rf $d
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -- python $ROOT/tools/gemm_one.py $shape 8 > /dev/null 2>&1
    db=$(find $d -name "*.db" | head -1)
    python - "$db" <<'PY' >> $F
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, cname, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%gemm2_kernel%' or kernel_name like '%splitk%' group by kernel_name, counter_name"):
    print(f"{name[:78]:78s} {cname:28s} {v:16.1f}  (n={n})")
PY
  done
done
cat $F

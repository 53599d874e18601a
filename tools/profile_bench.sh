#!/bin/bash
# rocprofv3 evidence for the bench line (run ON the GPU box through gpurun):
#   tools/profile_bench.sh <tag> [bench.py arguments...]
# 1. kernel trace + stats of `bench.py --no-extras <args>`  -> gpurun_out/<tag>_kernel_stats.csv (+ the bench JSON under rocprof)
# 2. HBM traffic of the dominant GEMM kernel on its largest shape (fc1: M=4096 N=4608 K=1152, tile 144 = gemm2_kernel<128,64,...>):
#    FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel-trace only), gfx950 half-count correction applied by the summary
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_c2}
shift || true
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $ROOT/bench.py --no-extras "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.log
find /tmp/prof_$TAG -type f | head -20 >&2; f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  d=/tmp/pmc_${TAG}_$(echo $c | cut -c1-10 | tr ' ' '_')
  rm -rf $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -- python $ROOT/tools/gemm_one.py 4096 4608 1152 144 8 > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  python - "$db" "$c" <<'PY' >> $OUT/${TAG}_pmc_fc1_tile144.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
t = "counters_collection" if "counters_collection" in tabs else [x for x in tabs if "counter" in x.lower()][0]
for name, cname, n, v in cur.execute(f"select kernel_name, counter_name, count(*), avg(value) from {t} where kernel_name like '%gemm2_kernel%' group by kernel_name, counter_name"):
    print(f"{name[:70]:70s} {cname:28s} {v:16.1f}  (n={n})")
PY
done
cat $OUT/${TAG}_pmc_fc1_tile144.txt
head -12 $OUT/${TAG}_kernel_stats.csv
tail -c 1500 $OUT/${TAG}_bench_under_rocprof.json

#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench.py run -> gpurun_out/<name>_kernel_stats.csv (+ the bench line)
#   tools/prof_bench.sh name [bench.py args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -- python $ROOT/bench.py "$@" > $ROOT/gpurun_out/${NAME}_bench_under_rocprof.log 2>&1
db=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $db $ROOT/gpurun_out/${NAME}_kernel_stats.csv
grep '^{"metric' $ROOT/gpurun_out/${NAME}_bench_under_rocprof.log | tail -1 > $ROOT/gpurun_out/${NAME}_bench_under_rocprof.json
head -12 $ROOT/gpurun_out/${NAME}_kernel_stats.csv | cut -c1-200

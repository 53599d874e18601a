#!/bin/bash
# per (kernel, grid) GEMM durations inside one bench run (kernel-trace only): tools/prof_gemm_shapes.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profS
rocprofv3 --kernel-trace --stats -d /tmp/profS -- python $ROOT/bench.py --steps 6 --warmup 2 "$@" > /tmp/profS.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/profS -name "*.db" | head -1) /tmp/profS.csv | grep gemm2

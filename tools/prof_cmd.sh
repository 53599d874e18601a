#!/bin/bash
# rocprofv3 --kernel-trace --stats over an arbitrary command -> gpurun_out/<name>_kernel_stats.csv:   tools/prof_cmd.sh name cmd...
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
( cd $ROOT && rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -- "$@" > $ROOT/gpurun_out/${NAME}_cmd.log 2>&1 )
db=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $db $ROOT/gpurun_out/${NAME}_kernel_stats.csv > /dev/null
head -14 $ROOT/gpurun_out/${NAME}_kernel_stats.csv | cut -c1-110,200-260

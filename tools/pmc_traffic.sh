#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of ONE GEMM configuration, merged into profiles/traffic.json
# (copied to gpurun_out/traffic.json so that it travels back):  tools/pmc_traffic.sh M N K tilecode
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcF /tmp/pmcW
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcF -- python $ROOT/tools/gemm_one.py "$@" 8 > /tmp/pmcF.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcW -- python $ROOT/tools/gemm_one.py "$@" 8 > /tmp/pmcW.log 2>&1
python $ROOT/tools/pmc_traffic.py $(find /tmp/pmcF -name "*.db" | head -1) $(find /tmp/pmcW -name "*.db" | head -1) $ROOT/profiles/traffic.json
cp $ROOT/profiles/traffic.json $ROOT/gpurun_out/traffic.json

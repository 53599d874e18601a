#!/bin/bash
# PMC passes (one counter set per run, kernel-trace only) over ONE GEMM configuration:
#   tools/pmc_gemm.sh M N K tilecode outname      -> gpurun_out/<outname>.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
M=$1; N=$2; K=$3; T=$4; OUT=$5
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
: > $ROOT/gpurun_out/$OUT.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -- python $ROOT/tools/gemm_one.py $M $N $K $T 6 > /tmp/pmc_$i.log 2>&1 || echo "pass $i failed: $set" >> $ROOT/gpurun_out/$OUT.txt
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $ROOT/tools/pmc_dump.py $db gemm >> $ROOT/gpurun_out/$OUT.txt
done

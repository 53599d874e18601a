#!/bin/bash
# Counter evidence of a round, IN SITU (the kernels as they run inside the sampling steps, weights cold, every shape they serve):
#   tools/pmc_round.sh r06 [workloads...]        (on the GPU box; default workloads: c2 scg c3)
# Per workload: one --kernel-trace pass (durations) + three --pmc passes, kernel-trace only, each in its own process as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE | WRITE_SIZE | "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE".
# -> gpurun_out/<tag>_pmc_<workload>.txt: per kernel with >= 1.5 % of the workload's GPU time: launches, average us, MFMA-busy fraction
#    (= SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): matrix-pipe cycles over the cycles the kernel had),
#    HBM bytes per launch (2 * FETCH_SIZE KiB + WRITE_SIZE KiB: the guide's gfx950 corrections) and GB/s.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; shift || true
WL=${@:-c2 scg c3}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  steps=2; [ $w = c2 ] && steps=3
  base=/tmp/pmcr_${TAG}_$w
  rm -rf ${base}_*
  timeout 900 rocprofv3 --kernel-trace -d ${base}_trace -- python $ROOT/bench.py --traffic-child --workload $w --steps $steps --warmup 1 > /dev/null 2>&1
  i=0
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $c --kernel-trace -d ${base}_pmc$i -- python $ROOT/bench.py --traffic-child --workload $w --steps $steps --warmup 1 > /dev/null 2>&1
  done
  python $ROOT/tools/pmc_round_summary.py $w ${base}_trace ${base}_pmc1 ${base}_pmc2 ${base}_pmc3 > $ROOT/gpurun_out/${TAG}_pmc_$w.txt 2>&1
  cat $ROOT/gpurun_out/${TAG}_pmc_$w.txt
done

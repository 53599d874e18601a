#!/bin/bash
# in-situ A/B of the 144-column tiles at B = 16 (RGM_T144 bits 2 / 4, gemm2_launch): bench line + kernel stats per setting
#   tools/t144_insitu.sh [settings...]      -> gpurun_out/t144_<v>_*
for x in ${@:-9 11 15}; do
  RGM_T144=$x bash tools/prof_bench.sh t144_$x --steps 10 --warmup 3 --no-extras --no-traffic > /dev/null 2>&1
  echo "=== RGM_T144=$x (under rocprof)"; cut -c1-160 gpurun_out/t144_${x}_kernel_stats.csv | head -9
  RGM_T144=$x python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unprofiled ms_per_step', d['ms_per_step'])"
done

#!/bin/bash
# raster A/B of the 144-column tiles in the B = 16 forward: RGM_G144_RASTER = sweep height (32 = whole columns, the old raster)
for r in 32 8 4; do
  for x in 11 15; do
    echo "=== RGM_G144_RASTER=$r RGM_T144=$x"
    RGM_G144_RASTER=$r RGM_T144=$x bash tools/prof_bench.sh t144r${r}_$x --steps 10 --warmup 3 --no-extras --no-traffic > /dev/null 2>&1
    grep "gemm144" gpurun_out/t144r${r}_${x}_kernel_stats.csv | cut -c1-30,100-160
    RGM_G144_RASTER=$r RGM_T144=$x python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unprofiled ms_per_step', d['ms_per_step'])"
  done
done
RGM_T144=9 python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T144=9 unprofiled ms_per_step', d['ms_per_step'])"
python tools/g144_stamp.py 4096 1152 1152 4096 1152 4608 1024 4608 1152 2>&1 | grep -v amdgpu

#!/bin/bash
# split-row stores of the gemm2 epilogues: RGM_SPLIT_PAIR = 0 two 8-byte stores per lane, 1 lane pairs (16 bytes per lane, DPP exchange),
# 2 eight columns per lane on the one-wave-per-SIMD tiles (two 16-byte stores, no exchange); -1 the per-family default
for v in ${@:-0 2 0 2}; do
  export RGM_SPLIT_PAIR=$v
  echo "=== RGM_SPLIT_PAIR=$v"
  python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 ms_per_step', d['ms_per_step'])"
done

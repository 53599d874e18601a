#!/bin/bash
# split-row stores of the gemm2 epilogues as lane pairs (16 bytes per lane): RGM_SPLIT_PAIR = 1 always, 0 never, -1 by kernel family (default)
for v in 1 0 -1 1 0 -1; do
  export RGM_SPLIT_PAIR=$v
  echo "=== RGM_SPLIT_PAIR=$v"
  python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 ms_per_step', d['ms_per_step'])"
  python bench.py --workload scg --steps 3 --warmup 1 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 ms_per_step', d['ms_per_step'])"
done

#!/bin/bash
# same-box A/B of the two-half-batch forward over the BASELINE workloads: RGM_DIT_HALVES=0 (never) against the default rule
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for hv in 0 -1; do
  for w in "c3 20" "scg 5" "long 5"; do set -- $w
    v=$(RGM_DIT_HALVES=$hv python bench.py --no-extras --workload $1 --steps $2 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "halves=$hv $1 $v"
  done
  v=$(RGM_DIT_HALVES=$hv python bench.py --no-extras --workload scg --simulate-ranks 8 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "halves=$hv scg_r8 $v"
  v=$(RGM_DIT_HALVES=$hv python bench.py --no-extras --workload long --simulate-ranks 8 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "halves=$hv long_r8 $v"
done; done

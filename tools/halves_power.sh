cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for hv in 0 -1; do for B in 32 64; do
RGM_DIT_HALVES=$hv python bench.py --batch $B --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('halves=$hv B=$B', d['ms_per_step'], 'ms  tflops', d['config'].get('algorithmic_tflops'), 'dominant', r['kernel'], r['avg_launch_us'], 'us frac', r['frac'], 'power', r.get('power'), 'traffic x', r.get('traffic_over_algorithmic'))
print('   by_kernel', r.get('by_kernel'))"
done; done; done

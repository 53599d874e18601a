python -m pytest tests/test_gpu_fullsize.py -x -q -k "vae or decode or conv or groupnorm or GroupNorm" 2>&1 | tail -3
for i in 1 2; do python bench.py --workload scg --steps 3 --warmup 1 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 ms_per_step', d['ms_per_step'])"; done

python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile_144 or heuristic_decompositions or split or big_tile" 2>&1 | tail -3
python -m pytest tests/test_gpu_round4.py tests/test_gpu_dit.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 unprofiled ms_per_step', d['ms_per_step'])"
python tools/batch_sweep.py 2 4 8 16 32 2>&1 | grep -v amdgpu
python bench.py --workload scg --steps 3 --warmup 1 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 ms_per_step', d['ms_per_step'])"
python bench.py --workload c3 --steps 10 --warmup 3 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 ms_per_step', d['ms_per_step'])"

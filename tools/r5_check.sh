# correctness first (the 144-tile test, the split-output GEMM tests, forward goldens), then the C2 line + kernel stats
python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile_144 or heuristic_decompositions or split" 2>&1 | tail -5
python -m pytest tests/test_gpu_round4.py tests/test_gpu_chain.py -x -q 2>&1 | tail -3
bash tools/prof_bench.sh r5a --steps 10 --warmup 3 --no-extras --no-traffic > /dev/null 2>&1
cut -c1-150 gpurun_out/r5a_kernel_stats.csv | head -10
python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 unprofiled ms_per_step', d['ms_per_step'])"
RGM_T144=9 python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 RGM_T144=9 ms_per_step', d['ms_per_step'])"
python tools/batch_sweep.py 2 4 8 16 32 2>&1 | grep -v amdgpu

python -m pytest tests/test_gpu_fullsize.py -x -q -k "fc2_reduce or cliff or half" 2>&1 | tail -3
python tools/batch_sweep.py 2 3 4 5 6 8 40 48 56 2>&1 | grep -v amdgpu

python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -2
python -m pytest tests/test_gpu_fullsize.py -x -q -k "fc2_reduce or cliff or half" 2>&1 | tail -2
python -m pytest tests/test_gpu_dit.py tests/test_gpu_round4.py tests/test_gpu_chain.py -x -q 2>&1 | tail -2

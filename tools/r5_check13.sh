python -m pytest tests/test_gpu_ops.py tests/test_gpu_dit.py -x -q 2>&1 | tail -2
for n in 1 2 4 8 16; do for q in 1 2 4 0; do RGM_ATTN_QSPLIT=$q python tools/attn_time.py $n 2>&1 | grep -v amdgpu; done; done
python tools/batch_sweep.py 2 3 4 5 6 8 2>&1 | grep -v amdgpu

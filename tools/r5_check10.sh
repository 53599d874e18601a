python -m pytest tests/test_gpu_ops.py tests/test_gpu_dit.py -x -q 2>&1 | tail -2
bash tools/ab_lib.sh rule-guided-music_amd/rgm/librgm_hip_prev.so --steps 20 --warmup 5
python tools/attn_time.py 2>&1 | grep -v amdgpu | tail -6

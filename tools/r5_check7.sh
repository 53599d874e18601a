python -m pytest tests/test_gpu_fullsize.py -x -q -k "fc2_reduce or cliff or small or tile_144" 2>&1 | tail -3
for v in 0 8 0 8; do
export RGM_P4_PF=$v
echo "=== RGM_P4_PF=$v"
python tools/batch_sweep.py 2 3 4 6 8 2>&1 | grep -v amdgpu
done

for v in 15 31 15 31; do
export RGM_T144=$v
echo "=== RGM_T144=$v"
python tools/batch_sweep.py 3 4 5 6 8 2>&1 | grep -v amdgpu
done

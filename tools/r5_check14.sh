python tools/batch_sweep.py 2 3 4 5 6 8 10 12 16 2>&1 | grep -v amdgpu
python tools/halves_exp.py 2 3 4 5 6 7 8 9 10 2>&1 | grep "B="

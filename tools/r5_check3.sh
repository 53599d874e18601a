for d in 8 16 4 8 16 4; do
export RGM_G144_PFD=$d
echo "=== RGM_G144_PFD=$d"
python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 unprofiled ms_per_step', d['ms_per_step'])"
python tools/batch_sweep.py 4 2>&1 | grep -v amdgpu
done

for v in 0 1 0 1; do
export RGM_T144_CO=$v
echo "=== RGM_T144_CO=$v"
python bench.py --workload c3 --steps 10 --warmup 3 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 ms_per_step', d['ms_per_step'])"
python tools/batch_sweep.py 20 24 32 2>&1 | grep -v amdgpu
done

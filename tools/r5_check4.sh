for v in 0 1 2 3 0 3; do
export RGM_ST_PLAIN=$v
echo "=== RGM_ST_PLAIN=$v (bit 0: fp32 rows of the 256x256 tiles plain, bit 1: their split rows plain)"
python bench.py --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 unprofiled ms_per_step', d['ms_per_step'])"
done
for v in 0 3; do
export RGM_ST_PLAIN=$v
bash tools/prof_bench.sh stp$v --steps 10 --warmup 3 --no-extras --no-traffic > /dev/null 2>&1
echo "=== RGM_ST_PLAIN=$v"; cut -c1-60,100-170 gpurun_out/stp${v}_kernel_stats.csv | head -9
done

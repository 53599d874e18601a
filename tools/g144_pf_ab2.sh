for pf in 0 1 3; do
  echo "##### RGM_G144_PF=$pf"
  export RGM_G144_PF=$pf
  for cfg in "4 9" "4 1" "16 11" "16 15"; do
    set -- $cfg
    RGM_T144=$2 bash tools/prof_bench.sh pf${pf}_b$1_t$2 --batch $1 --steps 10 --warmup 3 --no-extras --no-traffic > /dev/null 2>&1
    echo "B=$1 T144=$2: $(grep gemm144 gpurun_out/pf${pf}_b$1_t$2_kernel_stats.csv | cut -c100-170)"
    RGM_T144=$2 python bench.py --batch $1 --steps 20 --warmup 5 --no-extras --no-traffic 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   unprofiled ms_per_step', d['ms_per_step'])"
  done
done

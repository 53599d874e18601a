#!/bin/bash
# End-of-round evidence in ONE gpurun call: the default bench line, rocprofv3 kernel stats of every BASELINE workload and the small batches.
#   gpurun --timeout 3000 -- tools/end_of_round.sh r05
# (tools/isa_lint.py runs HERE, on the CPU -- it only compiles: python tools/isa_lint.py > profiles/r05_isa_lint.txt)
set -u
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_default_n1.json 2> gpurun_out/${R}_bench_default_n1.log
tools/prof_bench.sh ${R}_c2 --no-extras --steps 20 --warmup 5 > /dev/null 2>&1
tools/prof_bench.sh ${R}_c3 --no-extras --workload c3 --steps 20 --warmup 5 > /dev/null 2>&1
tools/prof_bench.sh ${R}_scg_b4_n16 --no-extras --workload scg --steps 5 --warmup 2 > /dev/null 2>&1
tools/prof_bench.sh ${R}_long_b2 --no-extras --workload long --steps 5 --warmup 2 > /dev/null 2>&1
tools/prof_bench.sh ${R}_scg_r8 --no-extras --workload scg --simulate-ranks 8 --steps 10 --warmup 3 > /dev/null 2>&1
tools/prof_bench.sh ${R}_long_b2_r8 --no-extras --workload long --simulate-ranks 8 --steps 5 --warmup 2 > /dev/null 2>&1
tools/prof_bench.sh ${R}_c2_b4 --no-extras --batch 4 --steps 20 --warmup 5 > /dev/null 2>&1
tools/prof_bench.sh ${R}_c2_b8 --no-extras --batch 8 --steps 20 --warmup 5 > /dev/null 2>&1
python tools/batch_sweep.py > gpurun_out/${R}_batch_sweep.txt 2>&1
python tools/cls_time.py 1 4 8 16 32 > gpurun_out/${R}_cls_time.txt 2>&1
python tools/hazard_soak.py 100 > gpurun_out/${R}_hazard_soak.txt 2>&1
# round 6: counter evidence per kernel, in situ (MFMA-busy %, HBM GB/s), and the round's own tests with their measurements printed
tools/pmc_round.sh ${R} c2 scg c3 > gpurun_out/${R}_pmc_all.log 2>&1
python -m pytest tests/test_gpu_round6.py tests/test_gpu_sampler.py tests/test_gpu_pins2.py -q -m gpu -s -k "chord_analyser or end_to_end or two_step or vae_decoder" 2>&1 | grep "^\[\|passed\|failed" > gpurun_out/${R}_round6_tests_printed.txt
for f in gpurun_out/${R}_*_bench_under_rocprof.json gpurun_out/${R}_bench_default_n1.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], d["config"].get("workload", "")[:60], "frac", d["roofline"]["frac"], "tflops", d["config"].get("algorithmic_tflops"))
except Exception as e:
    print("unreadable:", e)
PY
done
# un-profiled lines of the multi-stream workloads (rocprofv3's kernel trace serialises the side streams: C3 reads ~2 ms longer under it)
for w in c3 scg long; do  # (long: B = 2 per SURVEY 8d)
  python bench.py --no-extras --workload $w --steps $([ $w = c3 ] && echo 20 || echo 5) --warmup 3 > gpurun_out/${R}_${w}_bench_unprofiled.json 2>/dev/null
  python - gpurun_out/${R}_${w}_bench_unprofiled.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("unprofiled", d["config"]["workload"][:50], d["ms_per_step"], d["config"].get("repeats_ms_per_step"))
PY
done
python bench.py --no-extras --workload scg --simulate-ranks 8 --steps 10 --warmup 3 > gpurun_out/${R}_scg_r8_bench_unprofiled.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/${R}_scg_r8_bench_unprofiled.json').read().strip().splitlines()[-1]); print('unprofiled R=8', d['ms_per_step'])"

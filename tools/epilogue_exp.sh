# A/B of epilogue experiments on the SCG step (experiments build): per-kernel rows of bench.py's roofline.by_kernel
show() { grep "^{\"metric" | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"ms_per_step\"]); [print(r) for r in d[\"roofline\"][\"by_kernel\"][:3]]"; }
for e in 0 7 4; do echo "RGM_GEMM2_EXP=$e"; RGM_GEMM2_EXP=$e python bench.py --workload scg --steps 5 --no-extras --no-traffic 2>&1 | show; done

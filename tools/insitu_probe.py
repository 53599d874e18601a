#!/usr/bin/env python3
"""In-situ GEMM timing of the C2 step (DiTRotary_XL_8, B = 16, pre-split arithmetic): HIP-event duration of EVERY pre-split GEMM
launch of a few steps, grouped by its position in the block (qkv, proj, fc1, fc2) -- what tools/gemm_sweep.py measures in
isolation, measured where it runs.  usage (GPU box): python tools/insitu_probe.py [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R.set_gemm_precision("bf16x3_presplit")
work = bench.C2Workload(torch.device("cuda", 0), B)
for _ in range(5):
    work.step()
torch.cuda.synchronize()
R.check(R.lib.rgm_prof_reset())
R.check(R.lib.rgm_prof_enable(1))
STEPS = 3
for _ in range(STEPS):
    work.step()
torch.cuda.synchronize()
R.check(R.lib.rgm_prof_enable(0))
cap = 4096
ids, ms, fl = (C.c_int * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
n = R.lib.rgm_prof_dump(cap, ids, ms, fl)
rec = [(ids[i], ms[i] * 1e3, fl[i]) for i in range(n)]
per_step = n // STEPS
DEPTH = 28
assert per_step % DEPTH == 0, (n, per_step)
per_block = per_step // DEPTH                    # launches per block: 4 GEMMs, more when one is split over two kernels
print(f"{n} pre-split GEMM launches over {STEPS} steps ({per_step} per step, {per_block} per block)")
tot = 0.0
for j in range(per_block):
    sel = [r for i, r in enumerate(rec) if (i % per_step) % per_block == j]
    us = np.array([r[1] for r in sel])
    tf = sel[0][2] / (np.median(us) * 1e-6) / 1e12
    tot += us.sum() / STEPS
    print(f"launch {j} of a block: kernel id {sel[0][0]:3d}  {sel[0][2] / 1e9:7.2f} GFLOP  median {np.median(us):7.1f} us  min {us.min():7.1f}  "
          f"max {us.max():7.1f}  {tf:6.1f} TFLOP/s  first block {sel[0][1]:7.1f}  last block {sel[DEPTH - 1][1]:7.1f}")
print(f"GEMM total per step {tot / 1e3:.2f} ms (split-K reduce kernels not included)")

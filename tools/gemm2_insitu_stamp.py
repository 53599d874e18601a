#!/usr/bin/env python3
"""Stamps of the LAST gemm2 launch of tile RGM_GEMM2_DBG_TILE (default 71: the 256x256 kernel) in an XL-28 forward -- the kernel in situ.
Needs the stamped build: make BUILD=build_stamp OUT=../rgm/librgm_hip_stamp.so EXTRA=-DRGM_GEMM2_STAMPS; run with RGM_LIB_PATH=.../librgm_hip_stamp.so
usage: gemm2_insitu_stamp.py B [B ...]"""
import ctypes as C
import os
import sys

os.environ.setdefault("RGM_GEMM2_DBG_TILE", "71")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rule-guided-music_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402

arch = dict(depth=28, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=28, num_heads=16, num_classes=3, learn_sigma=False)
m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))
m = m.cuda().eval()
R.set_gemm_precision("bf16x3_presplit")
names = ["dma_wait", "barrier", "(prologue)", "read0", "mfma0", "(epilogue)", "mfma1"]
for B in [int(a) for a in sys.argv[1:]] or [16]:
    x = torch.randn(B, 4, 128, 16, device="cuda")
    t = torch.full((B,), 500, dtype=torch.int64, device="cuda")
    y = torch.ones(B, dtype=torch.int64, device="cuda")
    for _ in range(3):
        m(x, t, y)
    torch.cuda.synchronize()
    R.check(R.lib.rgm_gemm2_dbg(1, None))
    m(x, t, y)
    torch.cuda.synchronize()
    out = (C.c_longlong * 64)()
    R.check(R.lib.rgm_gemm2_dbg(2, out))
    R.check(R.lib.rgm_gemm2_dbg(0, None))
    print(f"B={B} tile {os.environ['RGM_GEMM2_DBG_TILE']} RGM_T144={os.environ.get('RGM_T144', 'default')}: per wave, cycles per K-tile")
    for w in range(8):
        kt = out[w * 8 + 7]
        if kt == 0:
            continue
        v = [out[w * 8 + i] for i in range(7)]
        loop = sum(v[i] for i in (0, 1, 3, 4, 6))
        print(f"  wave {w}: K-tiles {kt}  dma_wait {v[0] / kt:6.0f}  barrier {v[1] / kt:6.0f}  phaseA {v[4] / kt:6.0f}  phaseB {v[6] / kt:6.0f}  = {loop / kt:6.0f} / K-tile;"
              f" prologue {v[2]} K loop {loop} epilogue {v[5]}")

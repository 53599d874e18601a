#!/usr/bin/env python3
"""Stamps of the LAST 128x144 launch of an XL-28 forward (gemm144.hip, rgm_gemm144_dbg), i.e. the kernel in situ: cold weights, operands a
previous kernel just wrote, the chip at the forward's clock.  usage: [RGM_T144=mask] g144_insitu_stamp.py B [B ...]
  B = 16: RGM_T144=11 -> proj, 15 -> fc2 unsliced;  B = 4: RGM_T144=1 -> fc1, 9 -> fc2's K slices"""
import ctypes as C
import os
import sys

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
for B in [int(a) for a in sys.argv[1:]] or [16]:
    x = torch.randn(B, 4, 128, 16, device="cuda")
    t = torch.full((B,), 500, dtype=torch.int64, device="cuda")
    y = torch.ones(B, dtype=torch.int64, device="cuda")
    for _ in range(3):
        m(x, t, y)
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        m(x, t, y)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"B={B} RGM_T144={os.environ.get('RGM_T144', 'default')}: forward {np.median(ts):.3f} ms")
    R.check(R.lib.rgm_gemm144_dbg(1, None))
    m(x, t, y)
    allwg = bool(os.environ.get("ALLWG"))
    out = (C.c_longlong * (64 + 8 * 4096 if allwg else 64))()
    R.check(R.lib.rgm_gemm144_dbg(3 if allwg else 2, out))
    R.check(R.lib.rgm_gemm144_dbg(0, None))
    if allwg:            # every workgroup of the launch: wall clock (s_memrealtime, 100 MHz) and shader cycles
        w = np.array(out[64:], dtype=np.int64).reshape(4096, 8)
        w = w[w[:, 3] > 0]
        rt0 = w[:, 2].min()
        start, end = (w[:, 2] - rt0) / 100.0, (w[:, 3] - rt0) / 100.0          # us
        dur, cyc = end - start, (w[:, 1] - w[:, 0]).astype(float)
        q = lambda v: " / ".join(f"{x:.1f}" for x in np.percentile(v, [0, 50, 90, 100]))
        print(f"  {len(w)} workgroups; launch span {end.max():.1f} us; start min/med/p90/max {q(start)} us; end {q(end)} us; duration {q(dur)} us")
        print(f"  shader clock over a workgroup's life (cycles / wall): {q(cyc / dur / 1e3)} GHz; K loop cycles {q(w[:, 4])}; epilogue cycles {q(w[:, 5])}; prologue {q(w[:, 6])}")
        for x in range(8):
            sel = (w[:, 7] & 7) == x
            if sel.any():
                print(f"    XCD {x}: {sel.sum():3d} workgroups, start {np.median(start[sel]):6.1f}, end med / max {np.median(end[sel]):6.1f} / {end[sel].max():6.1f} us, K loop med {np.median(w[sel, 4]):8.0f} cycles")
    for w in range(8):
        v = [out[w * 8 + i] for i in range(8)]
        kt = max(v[7], 1)
        if w < 4:
            print(f"  consumer {w}: prologue {v[2]:6d} (set-up {v[4]}, barrier P passed at {v[5]})  K loop {v[1]:7d} = {v[1] / kt:6.0f} / K-tile x {kt} (barrier wait {v[0] / kt:5.0f})  epilogue {v[3]:6d}")
        else:
            print(f"  loader   {w}: prologue {v[2]:6d} (addresses set up at {v[3]}, first tile(s) issued and tile 0 landed at {v[6]})  K loop {v[1]:7d} = {v[1] / kt:6.0f} / K-tile (issue {v[4] / kt:5.0f}, landing {v[5] / kt:5.0f}, barrier {v[0] / kt:5.0f})")

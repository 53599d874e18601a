#!/usr/bin/env python3
"""XL-28 forward time over the batch sizes between the tuned ones (the tile ladder of gemm2_launch is fitted to B = 16 / 32 / 64 and C5's
window batches): ms per forward and per sample, and the ratio of neighbouring per-sample costs -- a value above 1.10 is a cliff
(tests/test_gpu_fullsize.py::test_forward_time_has_no_cliff_between_neighbouring_batch_sizes).  usage: batch_sweep.py [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rule-guided-music_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402


def sweep(batches, depth=28, reps=7, precision="bf16x3_presplit"):
    arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3, learn_sigma=False)
    m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))
    m = m.cuda().eval()
    R.set_gemm_precision(precision)
    out = []
    try:
        for B in batches:
            x = torch.randn(B, 4, 128, 16, device="cuda")
            t = torch.full((B,), 500, dtype=torch.int64, device="cuda")
            y = torch.ones(B, dtype=torch.int64, device="cuda")
            for _ in range(3):
                m(x, t, y)
            ts = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                m(x, t, y)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            out.append((B, float(np.median(ts))))
    finally:
        R.set_gemm_precision("fp32")
    return out


if __name__ == "__main__":
    Bs = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64]
    rows = sweep(Bs)
    prev = None
    for B, ms in rows:
        per = ms / B
        print(f"B={B:3d}  {ms:8.3f} ms  {per:7.4f} ms/sample" + (f"  x{per / prev:5.3f} of the previous" if prev else ""), flush=True)
        prev = per

#!/usr/bin/env python3
"""XL-28 forward: the blocks as two half batches on two streams (rgm_set_dit_halves) against the single-stream forward, same box, same
process: ms per forward both ways and the largest difference of the outputs.  tools/halves_exp.py [B ...]
(RGM_CO_MIN_TILES / RGM_CO_KT tune the co-scheduled tile choice of gemm2_launch; they are read when the library loads.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rule-guided-music_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402


def main(batches, depth=28, reps=9):
    arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3, learn_sigma=False)
    m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))
    m = m.cuda().eval()
    R.set_gemm_precision("bf16x3_presplit")
    try:
        for B in batches:
            x = torch.randn(B, 4, 128, 16, device="cuda")
            t = torch.full((B,), 500, dtype=torch.int64, device="cuda")
            y = torch.ones(B, dtype=torch.int64, device="cuda")
            res = {}
            for mode in (0, 2, 0, 2):
                R.check(R.lib.rgm_set_dit_halves(mode, None))
                for _ in range(3):
                    out = m(x, t, y)
                ts = []
                for _ in range(reps):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    out = m(x, t, y)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                res.setdefault(mode, []).append((float(np.median(ts)), out.clone()))
            o0, o2 = res[0][0][1], res[2][0][1]
            rel = float((o0 - o2).abs().max() / o0.abs().max())
            print(f"B={B:3d}  single {res[0][0][0]:8.3f} / {res[0][1][0]:8.3f} ms   halves {res[2][0][0]:8.3f} / {res[2][1][0]:8.3f} ms   "
                  f"max diff / max |out| {rel:.2e}  repeat-equal {bool(torch.equal(res[2][0][1], res[2][1][1]))}", flush=True)
    finally:
        R.check(R.lib.rgm_set_dit_halves(-1, None))
        R.set_gemm_precision("fp32")


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [8, 12, 16, 24, 32, 64])

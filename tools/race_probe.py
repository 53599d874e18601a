#!/usr/bin/env python3
"""Run-to-run determinism and accuracy of the XL forward at a given (N, H): a race shows as non-zero spread between identical calls."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/rule-guided-music_amd")
from rgm import synth, native as R
from guided_diffusion.dit import DiTRotary
depth = int(os.environ.get("DEPTH", "4"))
arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3, learn_sigma=False)
m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch)); m = m.cuda().eval()
for prec in os.environ.get("PRECS", "bf16x3_presplit,bf16x3").split(","):
    for N, H in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(24, 64), (28, 128)]:
        g = torch.Generator(device="cuda").manual_seed(N + H)
        x = torch.randn(N, 4, H, 16, device="cuda", generator=g); t = torch.full((N,), 500, dtype=torch.int64, device="cuda"); y = torch.ones(N, dtype=torch.int64, device="cuda")
        R.set_gemm_precision("fp32")
        exact = m(x, t, y).cpu().numpy()
        R.set_gemm_precision(prec)
        outs = [m(x, t, y).cpu().numpy() for _ in range(12)]
        spread = max(float(np.abs(o - outs[0]).max()) for o in outs)
        errs = [float(np.abs(o - exact).max() / np.abs(exact).max()) for o in outs]
        bad_rows = sorted({int(i) for o in outs for i in np.unique(np.argwhere(np.abs(o - exact) > 2e-4 * np.abs(exact).max())[:, 0])})
        print(f"{prec} depth {depth} N={N} H={H}: run-to-run spread {spread:.2e}; rel err vs fp32 min {min(errs):.2e} max {max(errs):.2e}; samples with outliers {bad_rows}", flush=True)

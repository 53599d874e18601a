#!/usr/bin/env python3
"""Soak of the image-level wait inside the fused GroupNorm launches: R decodes of N latents back to back, every roll compared with the
first one, per-decode time min / median / max (a tile whose bounded wait ran out costs 4 ms and shows as an outlier).
tools/gn_fuse_soak.py [N] [R]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from gpu_util import load_module  # noqa: E402
from taming.models.klvae_pedal import AutoencoderKL  # noqa: E402
from guided_diffusion.gaussian_diffusion import _decode  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
R.set_gemm_precision("bf16x3_presplit")
vae = load_module(AutoencoderKL(), synth.vae_state_dict(2, encoder=True))
z = torch.from_numpy(np.random.RandomState(N).randn(N, 4, 128, 16).astype(np.float32)).cuda()
first = _decode(z, vae, 1.0).clone()
ts, bad = [], 0
for _ in range(REPS):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    roll = _decode(z, vae, 1.0)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
    bad += 0 if torch.equal(roll, first) else 1
ts = sorted(ts)
print(f"N={N}: {REPS} decodes, {bad} differ from the first; ms per decode min {ts[0]:.3f} median {ts[len(ts) // 2]:.3f} p99 {ts[int(len(ts) * 0.99) - 1]:.3f} max {ts[-1]:.3f}")
R.set_gemm_precision("fp32")

#!/usr/bin/env python3
"""Timing of the VAE decode, the saving decode and the input-gradient pass.  tools/vae_grad_bench.py [N] [precision]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from taming.models.klvae_pedal import AutoencoderKL  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3_presplit"
R.set_gemm_precision(prec)
vae = AutoencoderKL()
vae.load_state_dict(synth.vae_state_dict(2, device="cuda", encoder=True))
vae = vae.to("cuda").eval()
lat = torch.randn(N, 4, 128, 16, device="cuda")
u = torch.randn(N, 3, 128, 1024, device="cuda")


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = timed(lambda: vae.decode_latent(lat, scale_factor=1.2465))
b = timed(lambda: vae.decode_latent_save(lat, scale_factor=1.2465))
c = timed(lambda: vae.decode_latent_vjp(u))
print(f"N={N} ({N * 8} squares) {prec}: decode {a:.1f} ms, decode_save {b:.1f} ms, vjp {c:.1f} ms; "
      f"grad workspace {R.lib.rgm_vae_grad_workspace_bytes(vae._handle, N * 8) / 2**30:.2f} GiB")

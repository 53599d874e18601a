#!/usr/bin/env python3
"""GroupNorm + swish inside the conv1 launches of the decoder (rgm_set_gn_fuse) against the separate pass, same process: decode time of N
latents (8 N squares), fused launches per decode, largest difference of the float rolls (mode 1 vs 0), and the fallback path (mode 2: every
tile writes raw rows and gn_fixup_kernel converts them) against mode 1.  tools/gn_fuse_ab.py [N ...]"""
import ctypes as C
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


def main(Ns):
    R.set_gemm_precision("bf16x3_presplit")
    vae = load_module(AutoencoderKL(), synth.vae_state_dict(2, encoder=True))
    try:
        for N in Ns:
            z = torch.from_numpy(np.random.RandomState(N).randn(N, 4, 128, 16).astype(np.float32)).cuda()
            res = {}
            for mode in (0, 1, 2, 0, 1):
                R.check(R.lib.rgm_set_gn_fuse(mode, None))
                n0 = R.lib.rgm_gn_fused_launches()
                roll = _decode(z, vae, 1.0)
                torch.cuda.synchronize()
                fused = R.lib.rgm_gn_fused_launches() - n0
                ts = []
                for _ in range(3 if mode != 2 else 1):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    roll = _decode(z, vae, 1.0)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                res.setdefault(mode, []).append((sorted(ts)[len(ts) // 2], roll.clone(), fused))
            r0, r1, r2 = res[0][0][1], res[1][0][1], res[2][0][1]
            rel = float((r1 - r0).abs().max() / r0.abs().max())
            print(f"N={N:3d}  separate pass {res[0][0][0]:8.3f} / {res[0][1][0]:8.3f} ms   in the conv launch {res[1][0][0]:8.3f} / {res[1][1][0]:8.3f} ms "
                  f"({res[1][0][2]} fused launches)   fallback path {res[2][0][0]:8.3f} ms   max diff / max |roll| {rel:.2e}   "
                  f"fallback == fused: {bool(torch.equal(r1, r2))}   repeat-equal {bool(torch.equal(r1, res[1][1][1]))}  finite {bool(torch.isfinite(r1).all())}", flush=True)
    finally:
        R.check(R.lib.rgm_set_gn_fuse(1, None))
        R.set_gemm_precision("fp32")


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [8, 64])

#!/usr/bin/env python3
"""GroupNorm-apply pass, eight channels per lane (gn_apply8_kernel) against four (RGM_GN_APPLY8=0), one process per setting: decode time of
N latents (8 N squares) and a digest of the float roll -- the two kernels must produce IDENTICAL rolls.  tools/gn_apply_ab.py [N]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(N):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]
    import numpy as np
    import torch
    from rgm import native as R, synth
    from gpu_util import load_module
    from taming.models.klvae_pedal import AutoencoderKL
    from guided_diffusion.gaussian_diffusion import _decode
    R.set_gemm_precision("bf16x3_presplit")
    vae = load_module(AutoencoderKL(), synth.vae_state_dict(2, encoder=True))
    z = torch.from_numpy(np.random.RandomState(0).randn(N, 4, 128, 16).astype(np.float32)).cuda()
    for _ in range(2):
        roll = _decode(z, vae, 1.0)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        roll = _decode(z, vae, 1.0)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{sorted(ts)[2]:.3f} ms  sha {hashlib.sha1(roll.cpu().numpy().tobytes()).hexdigest()[:16]}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(int(sys.argv[1]))
    else:
        N = sys.argv[1] if len(sys.argv) > 1 else "64"
        for rep in range(2):
            for v in ("0", "1"):
                out = subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, RGM_GN_APPLY8=v), capture_output=True, text=True)
                print(f"RGM_GN_APPLY8={v}  N={N}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]}", flush=True)

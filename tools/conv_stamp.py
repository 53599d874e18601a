#!/usr/bin/env python3
"""s_memtime profile of the LAST pre-split conv launch of a VAE decode (the 128 -> 128 channel conv2 of the last ResnetBlock at 128 x 128,
residual + GroupNorm sums in the epilogue; tile 72 = 512x128, channel-block-major K) -- middle workgroup, per wave:
K loop segments, prologue, epilogue.  Needs the stamped kernels:
  (cd rule-guided-music_amd/csrc && hipcc ... -DRGM_GEMM2_STAMPS -c gemm2.hip ...)  ->  RGM_LIB_PATH=.../librgm_hip_stamps.so
usage: RGM_LIB_PATH=rule-guided-music_amd/rgm/librgm_hip_stamps.so python tools/conv_stamp.py [latents=64]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from gpu_util import load_module  # noqa: E402
from taming.models.klvae_pedal import AutoencoderKL  # noqa: E402
from guided_diffusion.gaussian_diffusion import _decode  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R.set_gemm_precision("bf16x3_presplit")
vae = load_module(AutoencoderKL(), synth.vae_state_dict(2, encoder=True))
z = torch.randn(N, 4, 128, 16, device="cuda")
dbg = R.lib.rgm_gemm2_dbg
dbg.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
for _ in range(2):
    _decode(z, vae, scale_factor=1.2465)
torch.cuda.synchronize()
R.check(dbg(1, None))
_decode(z, vae, scale_factor=1.2465)
out = (C.c_longlong * 64)()
R.check(dbg(2, out))
R.check(dbg(0, None))
names = ["dma_wait", "barrier", "dma_iss", "read0", "mfma0", "read1", "mfma1"]
print("  wave " + " ".join(f"{n:>9s}" for n in names) + "   (cycles per K-tile)")
for w in range(8):
    kt = out[w * 8 + 7]
    if kt == 0:
        continue
    v = [out[w * 8 + i] / kt for i in range(7)]
    print(f"  {w:4d} " + " ".join(f"{x:9.0f}" for x in (v[0], v[1], 0, v[3], v[4], 0, v[6])) +
          f"   K-tiles {kt}  prologue {out[w * 8 + 2]}  epilogue {out[w * 8 + 5]}  K loop {sum(out[w * 8 + i] for i in (0, 1, 3, 4, 6))} cycles")
for w in range(4):
    if any(out[32 + w * 8 + i] for i in range(8)):
        print(f"  epilogue sub-stamps wave {w}: " + " ".join(f"{out[32 + w * 8 + i]:7d}" for i in range(8)))

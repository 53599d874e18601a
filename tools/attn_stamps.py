#!/usr/bin/env python3
"""Phase timing of rotary_attention_x3 (experiments build with -DRGM_ATTN_STAMPS): s_memtime at the phase boundaries of the 8 waves
of one workgroup, B = 16 x 16 heads, T = 256, head_dim 72.  usage: RGM_LIB_PATH=.../librgm_exp.so python tools/attn_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402
from rgm.synth import rotary_freqs  # noqa: E402

R.set_gemm_precision("bf16x3_presplit")
N, T, heads, hd = 16, 256, 16, 72
D = heads * hd
qkv = torch.randn(N * T, 3 * D, device="cuda")
o = torch.empty(N * T, D, device="cuda")
rot = hd // 2
ang = torch.arange(T, dtype=torch.float32)[:, None] * torch.from_numpy(rotary_freqs(rot))[None]
cs, sn = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda()
lib = C.CDLL(R.LIB_PATH)
for _ in range(5):
    R.check(R.lib.rgm_rotary_attention(R.ptr(qkv), R.ptr(o), R.ptr(cs), R.ptr(sn), N, T, heads, hd, rot // 2, R.current_stream()))
torch.cuda.synchronize()
buf = (C.c_longlong * (128 + 2048))()
assert lib.rgm_attn_stamps(buf) == 0
if os.environ.get("RGM_ATTN_BLOCKED", "1") != "0":   # key-blocked kernel: request+Q | deposit 0 | request 1 + barrier | per block: S^T | softmax + PV | deposit + request + barrier
    names = ["req0+Q", "dep0", "req1+bar"] + [f"b{b} {x}" for b in range(4) for x in ("S^T", "sm+PV", "dep+bar")]
    for w in range(8):
        t = [buf[w * 16 + i] for i in range(16)]
        print(f"wave {w}: " + " ".join(f"{names[i]} {t[i + 1] - t[i]:5d}" for i in range(14)) + f"  store {t[15] - t[14]}  total {t[15] - t[0]}")
else:
    names = ["stage K/V", "barrier", "Q frags", "S^T MFMA", "softmax", "PV MFMA", "store"]
    for w in range(8):
        t = [buf[w * 16 + i] for i in range(8)]
        print(f"wave {w}: " + "  ".join(f"{names[i]} {t[i + 1] - t[i]:6d}" for i in range(7)) + f"   total {t[7] - t[0]} shader clocks")

import numpy as np  # noqa: E402
rt = np.array(buf[128:128 + 2 * N * heads], dtype=np.int64).reshape(-1, 2)
t0 = rt[:, 0].min()
print("workgroup entry  (us after the first): min %.2f median %.2f max %.2f" % tuple(np.percentile((rt[:, 0] - t0) / 100.0, [0, 50, 100])))
print("workgroup exit   (us after the first entry): min %.2f median %.2f max %.2f" % tuple(np.percentile((rt[:, 1] - t0) / 100.0, [0, 50, 100])))
print("workgroup span us: min %.2f median %.2f max %.2f" % tuple(np.percentile((rt[:, 1] - rt[:, 0]) / 100.0, [0, 50, 100])))
last = 15 if os.environ.get("RGM_ATTN_BLOCKED", "1") != "0" else 7
print("block 7 span %.2f us -> shader clock ~ %.2f GHz" % ((rt[7, 1] - rt[7, 0]) / 100.0, (buf[last] - buf[0]) / ((rt[7, 1] - rt[7, 0]) * 10.0)))

#!/usr/bin/env python3
"""HIP-event time of rgm_rotary_attention (pre-split arithmetic) at the C2 shape: B x 16 heads, T = 256, head_dim 72.
usage: [RGM_LIB_PATH=...] python tools/attn_time.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402
from rgm.synth import rotary_freqs  # noqa: E402

R.set_gemm_precision("bf16x3_presplit")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, heads, hd = int(os.environ.get('ATTN_T', 256)), 16, 72
D = heads * hd
rot = hd // 2
ang = torch.arange(T, dtype=torch.float32)[:, None] * torch.from_numpy(rotary_freqs(rot))[None]
cs, sn = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda()
# 8 different qkv buffers in turn (8 x 57 MB at B = 16): no launch finds its input in L2
torch.manual_seed(7)
bufs = [torch.randn(N * T, 3 * D, device="cuda") for _ in range(8)]
o = torch.empty(N * T, D, device="cuda")


def run(i):
    R.check(R.lib.rgm_rotary_attention(R.ptr(bufs[i % 8]), R.ptr(o), R.ptr(cs), R.ptr(sn), N, T, heads, hd, int(os.environ.get('ROT_HALF', rot // 2)), R.current_stream()))


for i in range(10):
    run(i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200):
    run(i)
e1.record()
torch.cuda.synchronize()
import hashlib  # noqa: E402
run(0)
torch.cuda.synchronize()
digest = hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"rotary attention B={N} RGM_ATTN_QSPLIT={os.environ.get('RGM_ATTN_QSPLIT', 'auto')} output sha1 {digest}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch ({os.environ.get('RGM_LIB_PATH', 'librgm_hip.so')})")

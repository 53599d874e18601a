#!/usr/bin/env python3
"""Run-to-run determinism of the building blocks at the shapes of N half-windows (T = 128): attention, LN-modulate, GEMMs."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/rule-guided-music_amd")
from rgm import native as R
from rgm.synth import rotary_freqs
sys.path.insert(0, ROOT)
from oracle import dit_np as odit
N, T, heads, hd = int(sys.argv[1]), int(sys.argv[2]), 16, 72
D = heads * hd
M = N * T
st = R.current_stream()
g = torch.Generator(device="cuda").manual_seed(0)
def spread(fn, reps=15):
    outs = []
    for _ in range(reps):
        outs.append(fn().clone())
    torch.cuda.synchronize()
    return max(float((o - outs[0]).abs().max()) for o in outs), outs[0]
for prec in ("bf16x3", "fp32"):
    R.set_gemm_precision(prec)
    qkv = torch.randn(M, 3 * D, device="cuda", generator=g) * 1.5
    cos, sin = odit.rotary_tables(rotary_freqs(36), T)
    cd, sd_ = torch.from_numpy(cos).cuda(), torch.from_numpy(sin).cuda()
    def attn():
        o = torch.empty(M, D, device="cuda")
        R.check(R.lib.rgm_rotary_attention(R.ptr(qkv), R.ptr(o), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, 18, st))
        return o
    print(prec, "attention spread", spread(attn)[0], flush=True)
x = torch.randn(M, D, device="cuda", generator=g)
mod = torch.randn(N, 6 * D, device="cuda", generator=g)
def ln():
    o = torch.empty(M, D, device="cuda")
    R.check(R.lib.rgm_layernorm_modulate(R.ptr(x), R.ptr(o), M, D, 1e-6, None, None, mod.data_ptr() + 4 * D, mod.data_ptr() + 8 * D, 6 * D, T, st))
    return o
print("ln_mod spread", spread(ln)[0], flush=True)
for (Nn, K) in ((3456, 1152), (1152, 1152), (4608, 1152), (1152, 4608), (256, 32), (1152, 256)):
    a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(Nn, K, device="cuda", generator=g) * 0.03
    bias = torch.randn(Nn, device="cuda", generator=g)
    for prec in ("bf16x3", "fp32"):
        R.set_gemm_precision(prec)
        def gm():
            c = torch.empty(M, Nn, device="cuda")
            R.check(R.lib.rgm_gemm(R.ptr(a), K, R.ptr(b), K, R.ptr(c), Nn, M, Nn, K, R.ptr(bias), 0, 1.0, None, 0, 1, None, 0, st))
            return c
        print(f"gemm {prec} {M}x{Nn}x{K} spread", spread(gm)[0], flush=True)
    if K % 32 == 0 and Nn % 32 == 0:
        a2, b2 = torch.empty_like(a), torch.empty_like(b)
        R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st)); R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), Nn, K, st))
        need = max(int(R.lib.rgm_gemm_scratch_bytes(M, Nn)), 4096 + 8 * M * Nn * 4)
        ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
        def gs():
            c = torch.empty(M, Nn, device="cuda")
            R.check(R.lib.rgm_gemm_split_ws(R.ptr(a2), R.ptr(b2), R.ptr(c), M, Nn, K, R.ptr(bias), 0, 0, 0, R.ptr(ws), need, st))
            return c
        print(f"gemm presplit {M}x{Nn}x{K} spread", spread(gs)[0], flush=True)

#!/usr/bin/env python3
"""One DiT block composed from the C-ABI building blocks (bf16x3 on-the-fly arithmetic), repeated: which op is the first whose output
differs run to run?  usage: race_block.py N T"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/rule-guided-music_amd")
from rgm import native as R
from rgm.synth import rotary_freqs
from oracle import dit_np as odit
N, T = int(sys.argv[1]), int(sys.argv[2])
heads, hd = 16, 72
D = heads * hd
M = N * T
L = 6 * D
st = R.current_stream()
R.set_gemm_precision(os.environ.get("PREC", "bf16x3"))
g = torch.Generator(device="cuda").manual_seed(0)
x0 = torch.randn(M, D, device="cuda", generator=g)
mod = torch.randn(N, L, device="cuda", generator=g) * 0.3
Wqkv = torch.randn(3 * D, D, device="cuda", generator=g) * 0.03; bqkv = torch.randn(3 * D, device="cuda", generator=g) * 0.1
Wp = torch.randn(D, D, device="cuda", generator=g) * 0.03; bp = torch.randn(D, device="cuda", generator=g) * 0.1
W1 = torch.randn(4 * D, D, device="cuda", generator=g) * 0.03; b1 = torch.randn(4 * D, device="cuda", generator=g) * 0.1
W2 = torch.randn(D, 4 * D, device="cuda", generator=g) * 0.02; b2 = torch.randn(D, device="cuda", generator=g) * 0.1
cos, sin = odit.rotary_tables(rotary_freqs(36), T)
cd, sd_ = torch.from_numpy(cos).cuda(), torch.from_numpy(sin).cuda()
x, xm, qkv, ao, hid = (torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda"), torch.empty(M, 3 * D, device="cuda"),
                       torch.empty(M, D, device="cuda"), torch.empty(M, 4 * D, device="cuda"))
def mp(off): return mod.data_ptr() + 4 * off
def block():
    snaps = {}
    x.copy_(x0)
    R.check(R.lib.rgm_layernorm_modulate(R.ptr(x), R.ptr(xm), M, D, 1e-6, None, None, mp(0), mp(D), L, T, st)); snaps["ln1"] = xm.clone()
    R.check(R.lib.rgm_gemm(R.ptr(xm), D, R.ptr(Wqkv), D, R.ptr(qkv), 3 * D, M, 3 * D, D, R.ptr(bqkv), 0, 1.0, None, 0, 1, None, 0, st)); snaps["qkv"] = qkv.clone()
    R.check(R.lib.rgm_rotary_attention(R.ptr(qkv), R.ptr(ao), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, 18, st)); snaps["attn"] = ao.clone()
    R.check(R.lib.rgm_gemm(R.ptr(ao), D, R.ptr(Wp), D, R.ptr(x), D, M, D, D, R.ptr(bp), 0, 1.0, mp(2 * D), L, T, R.ptr(x), D, st)); snaps["proj"] = x.clone()
    R.check(R.lib.rgm_layernorm_modulate(R.ptr(x), R.ptr(xm), M, D, 1e-6, None, None, mp(3 * D), mp(4 * D), L, T, st)); snaps["ln2"] = xm.clone()
    R.check(R.lib.rgm_gemm(R.ptr(xm), D, R.ptr(W1), D, R.ptr(hid), 4 * D, M, 4 * D, D, R.ptr(b1), 2, 1.0, None, 0, 1, None, 0, st)); snaps["fc1"] = hid.clone()
    R.check(R.lib.rgm_gemm(R.ptr(hid), 4 * D, R.ptr(W2), 4 * D, R.ptr(x), D, M, D, 4 * D, R.ptr(b2), 0, 1.0, mp(5 * D), L, T, R.ptr(x), D, st)); snaps["fc2"] = x.clone()
    return snaps
ref = block()
bad = {}
for rep in range(12):
    cur = block()
    for k in cur:
        d = float((cur[k] - ref[k]).abs().max())
        if d > 0:
            rows = torch.unique(torch.nonzero((cur[k] - ref[k]).abs() > 0)[:, 0] // T).tolist()
            bad.setdefault(k, []).append((rep, d, rows[:8]))
torch.cuda.synchronize()
print(f"N={N} T={T} prec={os.environ.get('PREC', 'bf16x3')}:", "deterministic" if not bad else "")
for k in ("ln1", "qkv", "attn", "proj", "ln2", "fc1", "fc2"):
    if k in bad:
        print("  ", k, bad[k][:3])

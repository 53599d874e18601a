#!/usr/bin/env python3
"""s_memtime profile of the 128x144 kernel (gemm144.hip, middle workgroup): tools/g144_stamp.py M N K [M N K ...]

Per wave: consumers 0-3 -- prologue, K loop, cycles per K-tile, of which waiting at the barrier, epilogue; loaders 4-7 -- prologue,
K loop, per K-tile: piece issue, landing wait, barrier wait.  MFMA floor per K-tile: 54 x 17 = 918 cycles.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

args = [int(x) for x in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(1024, 4608, 1152), (4096, 1152, 1152), (4096, 1152, 4608)]
st = R.current_stream()
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") * 0.03
    c = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    a2, b2 = torch.empty_like(a), torch.empty_like(b)
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
    R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))

    def run():
        R.check(R.lib.rgm_gemm_split(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), 0, 81, 0, st))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    ref = a.double() @ b.double().t() + bias.double()
    err = ((c.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{M}x{N}x{K}: {us:.1f} us unstamped (warm, back to back) = {6.0 * M * N * K / us / 1e6:.0f} TF bf16-MFMA; max err {err:.2e}")
    R.check(R.lib.rgm_gemm144_dbg(1, None))
    if os.environ.get("COLD"):          # weights out of every cache, A just written by another kernel (as in the forward)
        big = torch.empty(1 << 28, device="cuda")
        big.zero_()
        a_src = a2.clone()
        big.zero_()
        a2.copy_(a_src)
        print("  (cold: 1 GiB written, then A rewritten by a copy kernel)")
    run()
    out = (C.c_longlong * 64)()
    R.check(R.lib.rgm_gemm144_dbg(2, out))
    R.check(R.lib.rgm_gemm144_dbg(0, None))
    for w in range(8):
        v = [out[w * 8 + i] for i in range(8)]
        kt = max(v[7], 1)
        if w < 4:
            print(f"  consumer {w}: prologue {v[2]:6d}  K loop {v[1]:7d} = {v[1] / kt:6.0f} / K-tile (barrier wait {v[0] / kt:5.0f})  epilogue {v[3]:6d}")
        else:
            print(f"  loader   {w}: prologue {v[2]:6d}  K loop {v[1]:7d} = {v[1] / kt:6.0f} / K-tile (issue {v[4] / kt:5.0f}, landing {v[5] / kt:5.0f}, barrier {v[0] / kt:5.0f})")

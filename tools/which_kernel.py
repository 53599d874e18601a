#!/usr/bin/env python3
"""Which kernels does the pre-split GEMM heuristic launch for a shape, and how long does each take?  (GPU box)
usage: python tools/which_kernel.py M N K [M N K ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

args = [int(x) for x in sys.argv[1:]]
zero = bool(os.environ.get("ZERO"))
for M, N, K in zip(args[0::3], args[1::3], args[2::3]):
    a = torch.zeros(M, K, device="cuda") if zero else torch.randn(M, K, device="cuda")
    b = torch.zeros(N, K, device="cuda") if zero else torch.randn(N, K, device="cuda") * 0.03
    a2, b2, c = torch.empty_like(a), torch.empty_like(b), torch.empty(M, N, device="cuda")
    st = R.current_stream()
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
    R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 8 * M * N * 4)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    tile = int(os.environ.get("TILE", "0"))
    for it in range(12):
        if it == 2:
            R.check(R.lib.rgm_prof_reset())
            R.check(R.lib.rgm_prof_enable(1))
        R.check(R.lib.rgm_gemm_split_ws(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, None, 0, tile, 0, R.ptr(ws), need, st))
    torch.cuda.synchronize()
    R.check(R.lib.rgm_prof_enable(0))
    out = []
    for kid in range(40, 140):
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        R.check(R.lib.rgm_prof_report(kid, C.byref(n), C.byref(ms), C.byref(fl)))
        if n.value:
            out.append(f"id {kid}: {n.value // 10} launch(es), {1e3 * ms.value / 10:.1f} us, {fl.value / ms.value / 1e9:.1f} TF")
    R.check(R.lib.rgm_prof_reset())
    print(f"M={M} N={N} K={K} tile={tile} zero={zero}: " + "; ".join(out), flush=True)

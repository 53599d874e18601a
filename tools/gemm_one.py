#!/usr/bin/env python3
"""Run ONE GEMM configuration a few times (for rocprofv3 --pmc passes): tools/gemm_one.py M N K tilecode [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

M, N, K, tile = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
a = torch.randn(M, K, device="cuda")
b = torch.randn(N, K, device="cuda") * 0.03
c = torch.empty(M, N, device="cuda")
bias = torch.randn(N, device="cuda")
st = R.current_stream()
if tile >= 100:                                       # pre-split operands, LDS-DMA kernel (gemm2), tile - 100
    a2, b2 = torch.empty_like(a), torch.empty_like(b)
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
    R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 8 * M * N * 4)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")         # the heuristic's K slices / stream-K need caller scratch
for _ in range(iters):
    if tile >= 100:
        R.check(R.lib.rgm_gemm_split_ws(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), 0, tile - 100, 0, R.ptr(ws), need, st))
    else:
        R.check(R.lib.rgm_gemm_tile(R.ptr(a), K, R.ptr(b), K, R.ptr(c), N, M, N, K, R.ptr(bias), 0, tile, st))
torch.cuda.synchronize()

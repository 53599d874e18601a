#!/usr/bin/env python3
"""SHA-1 of the 128x144 kernel's outputs on seeded operands, each shape run 20 times (a run that differs shows up as a second hash).  For A/B
builds of gemm144.hip that change WHEN a tile is multiplied but not in which order (round 6: LDS counters instead of the barrier per K-tile,
profiles/r06_g144_sync_counters_ab.txt): the outputs must be the same bytes in both builds."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

st = R.current_stream()
for M, N, K in [(1024, 4608, 1152), (4096, 1152, 1152), (4096, 1152, 4608), (1000, 1152, 1152), (1300, 4608, 64), (2500, 1152, 96), (128, 144, 64)]:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    b = torch.randn(N, K, device="cuda", generator=g) * 0.03
    bias = torch.randn(N, device="cuda", generator=g)
    a2, b2 = torch.empty_like(a), torch.empty_like(b)
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
    R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
    hs = set()
    for rep in range(20):
        c = torch.full((M, N), float("nan"), device="cuda")
        R.check(R.lib.rgm_gemm_split(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), 0, 81, 0, st))
        torch.cuda.synchronize()
        hs.add(hashlib.sha1(c.cpu().numpy().tobytes()).hexdigest())
    print(f"{M}x{N}x{K}: {sorted(hs)}")

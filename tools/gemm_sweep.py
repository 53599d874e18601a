#!/usr/bin/env python3
"""GEMM tile sweep on the GPU box: TFLOP/s of rgm_gemm_tile per (shape, tile), HIP-event timed.

usage (on the GPU box): python tools/gemm_sweep.py > gpurun_out/gemm_sweep.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

SHAPES = [("qkv", 4096, 3456, 1152), ("proj", 4096, 1152, 1152), ("fc1", 4096, 4608, 1152), ("fc2", 4096, 1152, 4608),
          ("qkv_b2", 512, 3456, 1152), ("fc2_b2", 512, 1152, 4608), ("fc1_scg", 16384, 4608, 1152), ("fc2_scg", 16384, 1152, 4608)]


def bench(M, N, K, tile, iters=20):
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") * 0.03
    c = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    st = R.current_stream()
    if tile >= 100:                                   # 100 + t: pre-split operands, LDS-DMA kernel (gemm2), tile t
        a2, b2 = torch.empty_like(a), torch.empty_like(b)
        R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
        R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))

        def run():
            R.check(R.lib.rgm_gemm_split(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), 0, tile - 100, 0, st))
    else:
        def run():
            R.check(R.lib.rgm_gemm_tile(R.ptr(a), K, R.ptr(b), K, R.ptr(c), N, M, N, K, R.ptr(bias), 0, tile, st))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / (ms * 1e-3) / 1e12, ms * 1e3


if __name__ == "__main__":
    # arguments: tile codes; add 32 for bf16x3 (e.g. 34 = 128x64 tile in bf16x3), 16 for explicit fp32
    tiles = [int(x) for x in sys.argv[1:]] or [2, 3, 33, 34, 35]
    bench(4096, 4608, 1152, 2, iters=50)      # warm the clocks before the first measured cell
    print(f"{'shape':10s} {'M':>6s} {'N':>5s} {'K':>5s} " + " ".join(f"tile{t}:TF/us".rjust(16) for t in tiles))
    for name, M, N, K in SHAPES:
        row = [bench(M, N, K, t) for t in tiles]
        print(f"{name:10s} {M:6d} {N:5d} {K:5d} " + " ".join(f"{tf:7.1f}/{us:8.1f}" for tf, us in row), flush=True)

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

if os.environ.get("SWEEP_SHAPES") == "big":
    SHAPES_OVERRIDE = [("fc1", 4096, 4608, 1152), ("fc1_scg", 16384, 4608, 1152), ("fc2_scg", 16384, 1152, 4608)]
elif os.environ.get("SWEEP_SHAPES") == "small":      # B = 2 / 4 / 8 latents
    SHAPES_OVERRIDE = [(f"{n}_b{b}", 256 * b, N, K) for b in (2, 4, 8)
                       for n, N, K in (("qkv", 3456, 1152), ("proj", 1152, 1152), ("fc1", 4608, 1152), ("fc2", 1152, 4608))]
elif os.environ.get("SWEEP_SHAPES") == "cls":        # DiTRotary-S/8 classifier (D = 384) at B = 32: forward and dgrad shapes
    SHAPES_OVERRIDE = [("qkv", 8224, 1152, 384), ("proj", 8224, 384, 384), ("fc1", 8224, 1536, 384), ("fc2", 8224, 384, 1536),
                       ("d_qkv", 8224, 384, 1152), ("d_fc2", 8224, 1536, 384), ("d_fc1", 8224, 384, 1536)]
elif os.environ.get("SWEEP_SHAPES") == "conv":       # dense stand-ins for the VAE's 3x3 convs (K = 9 Cin): huge M, N = Cout
    SHAPES_OVERRIDE = [("c128", 1 << 20, 128, 1152), ("c256", 1 << 19, 256, 2304), ("c512", 1 << 17, 512, 4608)]
elif os.environ.get("SWEEP_SHAPES") == "b32":        # XL backbone at B = 32 (C3)
    SHAPES_OVERRIDE = [("qkv", 8192, 3456, 1152), ("proj", 8192, 1152, 1152), ("fc1", 8192, 4608, 1152), ("fc2", 8192, 1152, 4608)]
elif os.environ.get("SWEEP_SHAPES") == "b64":        # XL backbone at n.B = 64 (C4's candidate batch)
    SHAPES_OVERRIDE = [("qkv", 16384, 3456, 1152), ("proj", 16384, 1152, 1152), ("fc1", 16384, 4608, 1152), ("fc2", 16384, 1152, 4608)]
elif os.environ.get("SWEEP_SHAPES") == "mid":        # B = 6 / 8 / 12 latents: M of a few thousand rows
    SHAPES_OVERRIDE = [(f"{n}_b{b}", 256 * b, N, K) for b in (6, 8, 12)
                       for n, N, K in (("qkv", 3456, 1152), ("proj", 1152, 1152), ("fc1", 4608, 1152), ("fc2", 1152, 4608))]
elif os.environ.get("SWEEP_SHAPES") == "ksweep":     # one full round of 256x256 tiles (16 x 14), K from 9 to 144 K-tiles: slope and intercept
    SHAPES_OVERRIDE = [(f"k{k}", 4096, 3584, k) for k in (288, 576, 1152, 2304, 4608)]
else:
    SHAPES_OVERRIDE = None
SHAPES = [("qkv", 4096, 3456, 1152), ("proj", 4096, 1152, 1152), ("fc1", 4096, 4608, 1152), ("fc2", 4096, 1152, 4608),
          ("qkv_b2", 512, 3456, 1152), ("fc2_b2", 512, 1152, 4608), ("fc1_scg", 16384, 4608, 1152), ("fc2_scg", 16384, 1152, 4608)]


ACT = int(os.environ.get('SWEEP_ACT', '0'))      # epilogue variants (gemm2 tiles only): 2 = GELU
SPLIT = int(os.environ.get('SWEEP_SPLIT', '0'))
PAD = int(os.environ.get('SWEEP_PAD', '0'))  # 1 = split-row output


def bench(M, N, K, tile, iters=20, check=True):
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") * 0.03
    # SWEEP_COLD=1: cycle through enough copies of the weight matrix that every launch reads it from HBM, as a model
    # with 28 different layers does (the default re-uses one B, which then sits in L2 / the 256 MB MALL)
    ncopy = max(2, int(600e6 // (N * K * 4))) if os.environ.get("SWEEP_COLD") else 1
    c = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    st = R.current_stream()
    if tile >= 100:                                   # 100 + t: pre-split operands, LDS-DMA kernel (gemm2), tile t
        ld = K + PAD                                  # SWEEP_PAD: row padding (elements) of the split operands
        a2, b2 = torch.zeros(M, ld, device="cuda"), torch.zeros(N, ld, device="cuda")
        R.check(R.lib.rgm_split_rows_ld(R.ptr(a), K, R.ptr(a2), ld, M, K, st))
        R.check(R.lib.rgm_split_rows_ld(R.ptr(b), K, R.ptr(b2), ld, N, K, st))
        bs = [b2] + [b2.clone() for _ in range(ncopy - 1)]
        cnt = [0]
        need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 16 * M * N * 4)
        ws = torch.zeros(need, dtype=torch.uint8, device="cuda")

        def run_ws():                                  # 100 = heuristic with scratch
            cnt[0] += 1
            R.check(R.lib.rgm_gemm_split_ws(R.ptr(a2), R.ptr(bs[cnt[0] % ncopy]), R.ptr(c), M, N, K, R.ptr(bias), ACT, tile - 100, SPLIT,
                                            R.ptr(ws), need, st))

        def run():
            if (tile in (100,) or 300 <= tile < 317) and not PAD:
                return run_ws()
            cnt[0] += 1
            R.check(R.lib.rgm_gemm_split_ld(R.ptr(a2), ld, R.ptr(bs[cnt[0] % ncopy]), ld, R.ptr(c), N, M, N, K, R.ptr(bias), ACT,
                                            tile - 100, SPLIT, st))
    else:
        bs = [b] + [b.clone() for _ in range(ncopy - 1)]
        cnt = [0]

        def run():
            cnt[0] += 1
            R.check(R.lib.rgm_gemm_tile(R.ptr(a), K, R.ptr(bs[cnt[0] % ncopy]), K, R.ptr(c), N, M, N, K, R.ptr(bias), 0, tile, st))
    for _ in range(3):
        run()
    if check and not os.environ.get('RGM_GEMM2_EXP') and not ACT and not SPLIT:                                         # max |c - (a b^T + bias)| relative to the output scale
        ref = torch.addmm(bias, a, b.t())
        err = ((c - ref).abs().max() / ref.abs().max()).item()
        assert err < 2e-4, f"tile {tile} wrong on {M}x{N}x{K}: rel err {err:.3e}"
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
    bench(4096, 4608, 1152, 2, iters=50, check=False)      # warm the clocks before the first measured cell
    print(f"{'shape':10s} {'M':>6s} {'N':>5s} {'K':>5s} " + " ".join(f"tile{t}:TF/us".rjust(16) for t in tiles))
    for name, M, N, K in (SHAPES_OVERRIDE or SHAPES):
        row = [bench(M, N, K, t) for t in tiles]
        print(f"{name:10s} {M:6d} {N:5d} {K:5d} " + " ".join(f"{tf:7.1f}/{us:8.1f}" for tf, us in row), flush=True)

#!/usr/bin/env python3
"""The forward's small dense GEMMs outside the blocks (x_embedder's two Linears, the final layer) on every tile of gemm.hip: microseconds per call,
back to back.  usage: small_gemm_time.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
st = R.current_stream()
for prec in ("bf16x3_presplit", "fp32"):
    R.set_gemm_precision(prec)
    for name, M, N, K, act in (("x_embed.0", 256 * B, 256, 32, 1), ("x_embed.2", 256 * B, 1152, 256, 0), ("final", 256 * B, 32, 1152, 0)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        c = torch.empty(M, N, device="cuda")
        out = []
        for tile in (0, 1, 2, 3, 4, 5):
            def run():
                R.check(R.lib.rgm_gemm_tile(R.ptr(a), K, R.ptr(w), K, R.ptr(c), N, M, N, K, R.ptr(b), act, tile, st))
            try:
                for _ in range(5):
                    run()
            except Exception as e:
                out.append(f"{tile}: n/a")
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            out.append(f"{tile}: {e0.elapsed_time(e1) * 20:.1f}")
        print(f"{prec:16s} {name:10s} {M}x{N}x{K}  us per call by tile  " + "  ".join(out))
R.set_gemm_precision("fp32")

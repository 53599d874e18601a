#!/usr/bin/env python3
"""s_memtime profile of the gemm2 K loop (middle workgroup): tools/gemm_stamp.py M N K tile [tile ...]

Needs the stamped kernels: (cd rule-guided-music_amd/csrc && touch gemm2.hip && make EXTRA=-DRGM_GEMM2_STAMPS)

Prints, per wave, the average cycles per K-tile spent in each segment of the loop body:
  dma_wait | barrier | dma_issue | read0 (8 ds_read + wait) | mfma0 (issue) | read1 | mfma1
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
tiles = [int(x) for x in sys.argv[4:]]
ACT, SPLIT = int(os.environ.get("ACT", "0")), int(os.environ.get("SPLIT", "0"))   # epilogue variant: activation (1 SiLU, 2 GELU), split-row output
a = torch.randn(M, K, device="cuda")
b = torch.randn(N, K, device="cuda") * 0.03
c = torch.empty(M, N, device="cuda")
bias = torch.randn(N, device="cuda")
st = R.current_stream()
a2, b2 = torch.empty_like(a), torch.empty_like(b)
R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
dbg = R.lib.rgm_gemm2_dbg
dbg.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
names = ["dma_wait", "barrier", "dma_iss", "read0", "mfma0", "read1", "mfma1"]
print(f"{M}x{N}x{K}  cycles per K-tile (s_memtime ticks = shader cycles)")
for t in tiles:
    for _ in range(3):
        R.check(R.lib.rgm_gemm_split(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), ACT, t, SPLIT, st))
    torch.cuda.synchronize()
    R.check(dbg(1, None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R.check(R.lib.rgm_gemm_split(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), ACT, t, SPLIT, st))
    e1.record()
    out = (C.c_longlong * 64)()
    R.check(dbg(2, out))
    us = e0.elapsed_time(e1) * 1e3
    R.check(dbg(0, None))
    print(f"tile {t}: stamped launch {us:.1f} us = {2.0 * M * N * K / us / 1e6:.1f} TF")
    print("  wave " + " ".join(f"{n:>9s}" for n in names) + "     total")
    for w in range(8):
        kt = out[w * 8 + 7]
        if kt == 0:
            continue
        v = [out[w * 8 + i] / kt for i in range(7)]
        if t >= 20:                                  # PIPE kernels: slots 2 / 5 hold the WHOLE prologue / epilogue (cycles)
            extra = f"   prologue {out[w * 8 + 2]} epilogue {out[w * 8 + 5]} cyc (K loop {sum(out[w * 8 + i] for i in (0, 1, 3, 4, 6))})"
            v[2] = v[5] = 0.0
        else:
            extra = ""
        print(f"  {w:4d} " + " ".join(f"{x:9.0f}" for x in v) + f" {sum(v):9.0f}" + extra)
    if any(out[32 + i] for i in range(32)):          # experiment builds: epilogue sub-stamps of waves 0-3 (cycles after the K loop)
        for w in range(4):
            print(f"  epilogue wave {w}: " + " ".join(f"{out[32 + w * 8 + i]:7d}" for i in range(8)))

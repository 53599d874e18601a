"""Do the small-batch GEMMs of the DiT forward (B = 4 / 8: the per-rank batches of the 8-GPU SCG step) run faster when their weights are
already on the die?  Each shape is timed (heuristic tile, pre-split operands) with the weights rotating over R copies: R = 1 stays in
L2, R = 4 (<= 100 MB) in the Infinity Cache, R = 16 (> 256 MB) comes from HBM every time -- what a prefetch of block i + 1's weights beside
block i could buy is the gap between R = 16 and R = 4.     python tools/cold_warm_probe.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M, D = 256 * B, 1152
st = R.current_stream()
R.set_gemm_precision("bf16x3_presplit")
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    a = torch.randn(M, K, device="cuda")
    a2 = torch.empty_like(a)
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
    c = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 8 * M * N * 4)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    res = []
    for copies in (1, 4, 16):
        bs = []
        for _ in range(copies):
            b = torch.randn(N, K, device="cuda") * 0.03
            b2 = torch.empty_like(b)
            R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
            bs.append(b2)

        def run(i):
            R.check(R.lib.rgm_gemm_split_ws(R.ptr(a2), R.ptr(bs[i % copies]), R.ptr(c), M, N, K, R.ptr(bias), 0, 0, 0, R.ptr(ws), need, st))
        for i in range(2 * copies + 4):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 64
        for i in range(n):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    print(f"B={B} {name:4s} M={M} N={N} K={K}: weights in L2 {res[0]:6.1f} us | in the Infinity Cache {res[1]:6.1f} us | from HBM {res[2]:6.1f} us")

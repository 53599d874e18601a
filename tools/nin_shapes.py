import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch
from rgm import native as R
st = R.current_stream()
for (M, N, K) in ((2097152, 128, 256), (524288, 256, 512), (131072, 512, 512)):
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda") * 0.03
    c = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    a2, b2 = torch.empty_like(a), torch.empty_like(b)
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st)); R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
    out = []
    for tile in (33, 34, 35, 36, 1, 143, 144):
        def run():
            if tile >= 100:
                R.check(R.lib.rgm_gemm_split(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), 0, tile - 100, 0, st))
            else:
                R.check(R.lib.rgm_gemm_tile(R.ptr(a), K, R.ptr(b), K, R.ptr(c), N, M, N, K, R.ptr(bias), 0, tile, st))
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gb = (M * K + M * N) * 4 / 1e9
        out.append(f"t{tile}: {ms*1e3:7.0f} us {gb/ms:6.2f} TB/s")
    print(M, N, K, " | ".join(out), flush=True)

#!/usr/bin/env python3
"""How launch-bound is a small-batch DiT forward?  Prints the wall time per forward; run under rocprofv3 --kernel-trace --stats
(tools/prof_small.sh) the summed kernel time / forwards gives the busy time to compare.  tools/launch_bound.py [B] [graph]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from guided_diffusion.dit import DiT_models  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
GRAPH = len(sys.argv) > 2 and sys.argv[2] == "graph"
R.set_gemm_precision("bf16x3_presplit")
dev = "cuda"
m = DiT_models["DiTRotary_XL_8"](input_size=[128, 16], in_channels=4, num_classes=3, learn_sigma=False)
arch = dict(depth=m.depth, hidden=m.hidden_size, heads=m.num_heads, patch=m.patch_size, in_ch=4, out_ch=m.out_channels,
            num_classes=m._n_embed, class_dropout=False)
m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / m.hidden_size ** 0.5, device=dev, **arch))
m = m.to(dev).eval()
x = torch.randn(B, 4, 128, 16, device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.int64)
y = torch.zeros(B, device=dev, dtype=torch.int64)
N = 40
for _ in range(3):
    ref = m(x, t, y)
torch.cuda.synchronize()
if GRAPH:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(x, t, y)
    g.replay()
    torch.cuda.synchronize()
    print("graph max diff", float((out - ref).abs().max()))
    fn = g.replay
else:
    fn = lambda: m(x, t, y)  # noqa: E731
t0 = time.perf_counter()
for _ in range(N):
    fn()
torch.cuda.synchronize()
print(f"B={B} graph={GRAPH} forwards={N + 3} wall_ms_per_forward={(time.perf_counter() - t0) / N * 1e3:.3f}")

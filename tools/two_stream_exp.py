#!/usr/bin/env python3
"""Experiment: does splitting the batch over two HIP streams (two half-batch forwards in flight) recover the tile-quantisation
tails of the single-stream forward?  tools/two_stream_exp.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from guided_diffusion.dit import DiT_models  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R.set_gemm_precision("bf16x3_presplit")
dev = "cuda"


def make():
    m = DiT_models["DiTRotary_XL_8"](input_size=[128, 16], in_channels=4, num_classes=3, learn_sigma=False)
    arch = dict(depth=m.depth, hidden=m.hidden_size, heads=m.num_heads, patch=m.patch_size, in_ch=4, out_ch=m.out_channels,
                num_classes=m._n_embed, class_dropout=False)
    m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / m.hidden_size ** 0.5, device=dev, **arch))
    return m.to(dev).eval()


m1, m2 = make(), make()        # two module instances = two workspaces (weights duplicated: experiment only)
x = torch.randn(B, 4, 128, 16, device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.int64)
y = torch.zeros(B, device=dev, dtype=torch.int64)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


single = timed(lambda: m1(x, t, y))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = B // 2


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        m1(x[:h], t[:h], y[:h])
    with torch.cuda.stream(s2):
        m2(x[h:], t[h:], y[h:])
    cur.wait_stream(s1)
    cur.wait_stream(s2)


print(f"B={B}: single stream {single:.3f} ms/forward, two half-batch streams {timed(two):.3f} ms/forward")

#!/usr/bin/env python3
"""Time of one classifier value-and-gradient call (DiTRotary-S/8-cls, depth 12) at the samplers' batches, per-(sample, head) attention
workgroups (rgm_set_attn_split 0) against per-tile ones (1).  usage: cls_time.py [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rule-guided-music_amd")]
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from guided_diffusion.dit import DiT_models  # noqa: E402

clf = DiT_models["DiTRotary-S/8-cls"](input_size=[128, 16], in_channels=4, num_classes=16)
clf.load_state_dict(synth.dit_state_dict(3, device="cuda", depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16))
clf = clf.cuda().eval()
R.set_gemm_precision("bf16x3_presplit")
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 32]:
    x = torch.randn(B, 4, 128, 16, device="cuda")
    t = torch.full((B,), 300, dtype=torch.int64, device="cuda")
    tgt = torch.full((B, 16), 3.0, device="cuda")
    row = []
    for mode in (0, 1):
        R.check(R.lib.rgm_set_attn_split(mode))
        for _ in range(3):
            clf.value_and_grad(x, t, tgt, "mse", 10.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            clf.value_and_grad(x, t, tgt, "mse", 10.0)
        b.record()
        torch.cuda.synchronize()
        row.append(a.elapsed_time(b) / 20)
    R.check(R.lib.rgm_set_attn_split(-1))
    print(f"B={B:3d}: value-and-grad {row[0]:.3f} ms per (sample, head) workgroups, {row[1]:.3f} ms per tile", flush=True)

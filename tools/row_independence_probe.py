import os, sys, numpy as np, torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/rule-guided-music_amd"); sys.path.insert(0, ROOT + "/tests")
from rgm import synth, native as R
from guided_diffusion.dit import DiTRotary
R.set_gemm_precision("bf16x3_presplit")
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
for depth in (2, 8, 28):
    arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3, learn_sigma=False)
    m.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch)); m = m.cuda().eval()
    for N, H in ((28, 128), (24, 64), (16, 128), (32, 128)):
        g = torch.Generator(device="cuda").manual_seed(N + H)
        x = torch.randn(N, 4, H, 16, device="cuda", generator=g); t = torch.full((N,), 500, dtype=torch.int64, device="cuda"); y = torch.ones(N, dtype=torch.int64, device="cuda")
        big = m(x, t, y).cpu().numpy()
        small = torch.cat([m(x[i:i+2].contiguous(), t[i:i+2].contiguous(), y[i:i+2].contiguous()) for i in range(0, N, 2)]).cpu().numpy()
        R.set_gemm_precision("fp32")
        exact = m(x, t, y).cpu().numpy()
        R.set_gemm_precision("bf16x3_presplit")
        print(f"depth {depth} N={N} H={H}: big vs small {rel(big, small):.2e}   big vs fp32 {rel(big, exact):.2e}   small vs fp32 {rel(small, exact):.2e}", flush=True)

"""Debugging aid for csrc/chain.hip: run the persistent launch truncated after k ops of block 0 (RGM_CHAIN_STOP_OP, one process per k) and
report NaNs / run-to-run differences of the activation strips.   python tools/chain_phases.py K [B]"""
import os
import sys

K = int(sys.argv[1])
os.environ["RGM_CHAIN_STOP_OP"] = str(K)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from gpu_util import load_module  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402

B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
D, T = 1152, 256
R.set_gemm_precision("bf16x3_presplit")
arch = dict(depth=1, hidden=D, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=D, depth=1, num_heads=16, num_classes=3,
                          learn_sigma=False), synth.dit_state_dict(3, final_std=0.3 / D ** 0.5, device="cuda", **arch))
rng = np.random.RandomState(5)
x = torch.from_numpy(rng.randn(B, 4, 128, 16).astype(np.float32)).cuda()
t = torch.from_numpy(rng.randint(0, 1000, size=B).astype(np.int64)).cuda()
y = torch.from_numpy(rng.randint(0, 3, size=B).astype(np.int64)).cuda()
M = B * T
sizes = [("tok_in", M * 32), ("h1", M * 256), ("x", M * D), ("xm", M * D), ("qkv", M * 3 * D), ("ao", M * D), ("hid", M * 4 * D)]


def strips():
    ws, out, off = m._ws, {}, 0
    for name, n in sizes:
        out[name] = ws[off:off + n * 4].clone().view(torch.int32)
        off += (n * 4 + 255) // 256 * 256
    return out


R.check(R.lib.rgm_set_dit_chain(1, None))
runs = []
for k in range(3):
    m._ws.zero_() if m._ws is not None else None
    m(x, t, y)
    torch.cuda.synchronize()
    runs.append(strips())
for name, n in sizes[2:]:
    f = runs[0][name].view(torch.float32)
    d01 = int((runs[0][name] != runs[1][name]).sum())
    d02 = int((runs[0][name] != runs[2][name]).sum())
    print(f"K={K} {name:4s} nan(as f32)={int(torch.isnan(f).sum())} nonzero={int((runs[0][name] != 0).sum())} of {n}  run0!=run1: {d01}  run0!=run2: {d02}", flush=True)
if K in (2, 3):
    a, b = runs[0]["ao"], runs[1]["ao"]
    idx = (a != b).nonzero().flatten().cpu().numpy()
    rows, cols = idx // D, idx % D
    import collections
    print("ao differing rows:", sorted(collections.Counter(rows.tolist()).items())[:40])
    print("ao differing col blocks (col // 64 = 32-element block):", sorted(collections.Counter((cols // 64).tolist()).items()))
    print("ao differing col % 64:", sorted(collections.Counter((cols % 64).tolist()).items()))
    for i in idx[:8]:
        print("  dword", int(i // D), int(i % D), hex(int(a[i]) & 0xffffffff), hex(int(b[i]) & 0xffffffff))
if K == 3:
    f = runs[0]["x"].view(torch.float32).view(M, D)
    nanrows = torch.isnan(f).any(dim=1).nonzero().flatten().tolist()
    print("x NaN rows:", nanrows, "NaN count per such row:", [int(torch.isnan(f[r]).sum()) for r in nanrows[:20]])

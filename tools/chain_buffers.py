"""Debugging aid for csrc/chain.hip: at depth 1 every activation strip of the workspace still holds block 0's values after a forward --
compare them between the launch-per-GEMM forward and the persistent launch (split-row strips as raw bits).
    python tools/chain_buffers.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from gpu_util import load_module  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D, T = 1152, 256
R.set_gemm_precision("bf16x3_presplit")
arch = dict(depth=1, hidden=D, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=D, depth=1, num_heads=16, num_classes=3,
                          learn_sigma=False), synth.dit_state_dict(3, final_std=0.3 / D ** 0.5, device="cuda", **arch))
rng = np.random.RandomState(5)
x = torch.from_numpy(rng.randn(B, 4, 128, 16).astype(np.float32)).cuda()
t = torch.from_numpy(rng.randint(0, 1000, size=B).astype(np.int64)).cuda()
y = torch.from_numpy(rng.randint(0, 3, size=B).astype(np.int64)).cuda()


def al(n):
    return (n * 4 + 255) // 256 * 256


M = B * T
sizes = [("tok_in", M * 32), ("h1", M * 256), ("x", M * D), ("xm", M * D), ("qkv", M * 3 * D), ("ao", M * D), ("hid", M * 4 * D)]


def strips():
    ws = m._ws
    out, off = {}, 0
    for name, n in sizes:
        out[name] = ws[off:off + n * 4].clone().view(torch.int32)
        off += al(n)
    return out


ref = m(x, t, y).clone()
torch.cuda.synchronize()
a = strips()
R.check(R.lib.rgm_set_dit_chain(1, None))
out = m(x, t, y).clone()
torch.cuda.synchronize()
b = strips()
for name, n in sizes[2:]:
    fa, fb = a[name].view(torch.float32), b[name].view(torch.float32)
    same = bool(torch.equal(a[name], b[name]))
    bad = (a[name] != b[name])
    nb = int(bad.sum())
    msg = f"{name:4s} bit-identical={same} differing dwords={nb} of {n}"
    if nb:
        idx = bad.nonzero().flatten()
        cols = {"x": D, "xm": D, "qkv": 3 * D, "ao": D, "hid": 4 * D}[name]
        rows = (idx // cols)
        msg += f" rows {int(rows.min())}..{int(rows.max())} cols {int((idx % cols).min())}..{int((idx % cols).max())}"
        if name in ("x", "qkv"):
            msg += f" nan={int(torch.isnan(fb).sum())} max|d|={float((fa - fb).abs().nan_to_num(1e9).max()):.3e}"
        first = idx[:6].tolist()
        msg += f" first {[(i // cols, i % cols) for i in first]}"
    print(msg, flush=True)
print("out nan", int(torch.isnan(out).sum()), "rel", float((out - ref).abs().max() / ref.abs().max()))

"""Per-phase spans of the persistent DiT forward (csrc/chain.hip) from its per-item time stamps (RGM_CHAIN_TIMES=1, rgm_dit_chain_times).
For every phase (qkv, attn, proj, ln2, fc1, fc2, red) over all blocks: items, body time (dependencies met -> finished) mean / p90, wait
(claimed -> dependencies met) mean, and for one block in the middle the wall-clock span of each phase; plus per-workgroup busy fraction.
    RGM_CHAIN_TIMES=1 python tools/chain_spans.py [B] [depth] [order]  > gpurun_out/chain_spans.txt"""
import ctypes as C
import os
import sys

os.environ.setdefault("RGM_CHAIN_TIMES", "1")
if len(sys.argv) > 3:
    os.environ["RGM_DIT_CHAIN_ORDER"] = sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from gpu_util import load_module  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 28
R.set_gemm_precision("bf16x3_presplit")
arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=0)
m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=0,
                          learn_sigma=False), synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))
rng = np.random.RandomState(5)
x = torch.from_numpy(rng.randn(B, 4, 128, 16).astype(np.float32)).cuda()
t = torch.from_numpy(rng.randint(0, 1000, size=B).astype(np.int64)).cuda()
R.check(R.lib.rgm_set_dit_chain(1, None))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(5):
    m(x, t)
torch.cuda.synchronize()
ev[0].record()
for _ in range(10):
    m(x, t)
ev[1].record()
torch.cuda.synchronize()
print(f"B={B} depth={depth} order={os.environ.get('RGM_DIT_CHAIN_ORDER', '0')}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms per forward (with time stamps on)")
nn = C.c_int(0)
R.check(R.lib.rgm_dit_chain_times(m._handle, None, None, 0, C.byref(nn)))
n = nn.value
tb = np.zeros((n, 8), dtype=np.uint64)
ib = np.zeros((n, 4), dtype=np.uint32)
R.check(R.lib.rgm_dit_chain_times(m._handle, tb.ctypes.data_as(C.c_void_p), ib.ctypes.data_as(C.c_void_p), n, C.byref(nn)))
t0 = tb[:, 0].min()
claim, ready, end, wg = [(tb[:, k].astype(np.int64) - (int(t0) if k < 3 else 0)) * (0.01 if k < 3 else 1) for k in range(4)]   # us
op = (ib[:, 0] & 0xffff).astype(int)
blk, ph = op // 7, op % 7
names = ["qkv", "attn", "proj", "ln2", "fc1", "fc2", "red"]
print(f"items {n}; launch span {end.max():.1f} us = {end.max() / depth:.1f} us per block")
w0done = (tb[:, 4].astype(np.int64) - int(t0)) * 0.01
w0ret = (tb[:, 5].astype(np.int64) - int(t0)) * 0.01
print("phase  items  body_mean  body_p10  body_p90   wait_mean  wait_p90   sum_body_ms   wave0: compute  drain  barrier+next")
for p in range(7):
    s = ph == p
    if not s.any():
        continue
    body, wait = end[s] - ready[s], ready[s] - claim[s]
    print(f"{names[p]:5s} {s.sum():6d} {body.mean():9.1f} {np.percentile(body, 10):9.1f} {np.percentile(body, 90):9.1f} {wait.mean():10.1f} {np.percentile(wait, 90):9.1f} {body.sum() / 1000:12.2f}   {(w0done[s] - ready[s]).mean():14.1f} {(w0ret[s] - w0done[s]).mean():6.1f} {(end[s] - w0ret[s]).mean():8.1f}")
tot_body = (end - ready).sum()
tot_wait = (ready - claim).sum()
print(f"sum over items: body {tot_body / 1000:.1f} ms, wait {tot_wait / 1000:.1f} ms; 256 workgroups x span = {256 * end.max() / 1000:.1f} ms "
      f"-> busy {tot_body / (256 * end.max()):.3f}, waiting {tot_wait / (256 * end.max()):.3f}")
for b in sorted({min(depth - 1, 1), depth // 2}):
    print(f"block {b}: phase  first_ready  last_end   (us from the block's first claim)")
    s0 = claim[blk == b].min()
    for p in range(7):
        s = (blk == b) & (ph == p)
        if not s.any():
            continue
        print(f"   {names[p]:5s} {ready[s].min() - s0:10.1f} {end[s].max() - s0:10.1f}   median end {np.median(end[s]) - s0:8.1f}")
    if b + 1 < depth:
        print(f"   next block's first claim at {claim[blk == b + 1].min() - s0:.1f}")

"""Debugging aid for the persistent DiT forward (csrc/chain.hip): run ONE chained forward without waiting for it and watch its control
words from a private stream (rgm_dit_chain_peek; RGM_CHAIN_TRACE=1 adds where every workgroup is).
    RGM_CHAIN_TRACE=1 timeout 120 python tools/chain_diag.py [B] [depth] > gpurun_out/chain_diag.txt"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rgm import native as R, synth  # noqa: E402
from gpu_util import load_module  # noqa: E402
from guided_diffusion.dit import DiTRotary  # noqa: E402


def log(*a):
    print(*a, flush=True)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
R.set_gemm_precision("bf16x3_presplit")
arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3,
                          learn_sigma=False), synth.dit_state_dict(3, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))
rng = np.random.RandomState(5)
x = torch.from_numpy(rng.randn(B, 4, 128, 16).astype(np.float32)).cuda()
t = torch.from_numpy(rng.randint(0, 1000, size=B).astype(np.int64)).cuda()
y = torch.from_numpy(rng.randint(0, 3, size=B).astype(np.int64)).cuda()
ref = m(x, t, y).clone()
torch.cuda.synchronize()
log("launch-per-GEMM forward done", float(ref.abs().max()))
R.check(R.lib.rgm_set_dit_chain(1, None))
per_sample = depth * (14 + 16 + 5 + 16 + 18 + 15 + 16)
log("items per sample", per_sample, "total", per_sample * B)
out = m(x, t, y)                       # enqueued; the call does not wait
WORDS = 8 + 4096 + 2 * 512
buf = (C.c_uint * WORDS)()
done = False
for k in range(24):
    time.sleep(0.25)
    R.check(R.lib.rgm_dit_chain_peek(m._handle, buf, WORDS))
    head, err = buf[0], buf[1]
    prog = [buf[8 + g] for g in range(B)]
    states = {}
    waiting = []
    for w in range(256):
        it, st = buf[8 + 4096 + 2 * w], buf[8 + 4096 + 2 * w + 1]
        states[st] = states.get(st, 0) + 1
        if st in (1, 2, 3) and len(waiting) < 12:
            waiting.append((w, it - 1, st))
    log(f"t={0.25 * (k + 1):.2f}s head={head} error={err} progress={prog[:8]} states={states} some={waiting}")
    if all(p == per_sample for p in prog) or err:
        done = True
        break
if not done:
    log("NOT FINISHED after 6 s -- giving up without synchronising")
    os._exit(3)
torch.cuda.synchronize()
e = float((out - ref).abs().max() / ref.abs().max())
log("chained forward done; rel err vs launch-per-GEMM", e, "status words", buf[0], buf[1])

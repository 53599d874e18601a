#!/usr/bin/env python3
"""Socket power and shader clock (rocm-smi, sampled from a thread) while the 256x256 pre-split GEMM runs back to back for ~8 s, on random
and on all-zero operands: is the kernel power-limited?   usage: python tools/power_probe.py"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rule-guided-music_amd"))
import torch
from rgm import native as R

M, N, K = 16384, 4608, 1152
st = R.current_stream()
need = int(R.lib.rgm_gemm_scratch_bytes(M, N))
ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
c = torch.empty(M, N, device="cuda")
bias = torch.zeros(N, device="cuda")


def sampler(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = [l.split(":")[-1].strip() for l in t.splitlines() if "Power (W)" in l]
            ck = [l.split("(")[-1].split(")")[0] for l in t.splitlines() if "sclk" in l]
            out.append((pw[0] if pw else "?", ck[0] if ck else "?"))
        except Exception as e:
            out.append(("err", str(e)[:40]))


for name, scale in (("random", 1.0), ("zeros", 0.0)):
    a = torch.randn(M, K, device="cuda") * scale
    b = torch.randn(N, K, device="cuda") * 0.03 * scale
    a2, b2 = torch.empty_like(a), torch.empty_like(b)
    R.check(R.lib.rgm_split_rows(R.ptr(a), R.ptr(a2), M, K, st))
    R.check(R.lib.rgm_split_rows(R.ptr(b), R.ptr(b2), N, K, st))
    run = lambda: R.check(R.lib.rgm_gemm_split_ws(R.ptr(a2), R.ptr(b2), R.ptr(c), M, N, K, R.ptr(bias), 0, 71, 0, R.ptr(ws), need, st))
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out)); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while time.time() - t0 < 8.0:
        for _ in range(200):
            run()
        n += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name}: {us:.1f} us per launch = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s fp32-equivalent;  (W, sclk) samples: {out[1:-1]}")

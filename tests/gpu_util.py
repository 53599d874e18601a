"""Helpers shared by the -m gpu tests: every call goes through the C ABI (ctypes), never torch math."""
import ctypes as C
import numpy as np
import torch

from rgm import native as R

DEV = "cuda"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def rel(a, b):
    """NORM-WISE relative error max|a - b| / max|b| -- the measure every "rel" tolerance of the GPU tests (and the north
    star's "latents within 1e-3 rel") is stated in.  It bounds the error relative to the tensor's scale; entries much smaller
    than the largest one are constrained absolutely (to tol * max|b|), not relatively."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def u8_flip_report(u8, golden_u8, roll, tol=1e-4):
    """uint8 piano rolls (B,128,T,3) of this implementation and of the reference + this implementation's float roll (B,3,128,T).
    -> (mismatches, mismatches that are NOT explained by fp32 re-association across a quantisation boundary, largest distance
    of a mismatching value to its boundary).  A flip is explained when the float value sits within `tol` of the -0.95
    background threshold (0 <-> 3 snap) or of a value where (x + 1) * 63.5 crosses an integer (one grey level)."""
    u8, golden_u8 = np.asarray(u8), np.asarray(golden_u8)
    v = np.asarray(roll, dtype=np.float64).transpose(0, 2, 3, 1)
    bad = u8 != golden_u8
    if not bad.any():
        return 0, 0, 0.0
    vb = v[bad]
    q = (vb + 1.0) * 63.5
    d_int = np.abs(q - np.round(q)) / 63.5                    # distance to the nearest truncation boundary, in roll units
    d_thr = np.abs(vb + 0.95)
    dist = np.minimum(d_int, d_thr)
    step = np.abs(u8[bad].astype(np.int32) - golden_u8[bad].astype(np.int32))
    explained = (dist < tol) & ((step == 1) | (d_thr < tol))
    return int(bad.sum()), int((~explained).sum()), float(dist.max())


def gemm(A, B, bias=None, act=0, alpha=1.0, gate=None, rows_per_gate=1, res=None, tile=None, prec=None):
    """C = epi(alpha * A @ B.T) through rgm_gemm; numpy in, numpy out."""
    M, K = A.shape
    N = B.shape[0]
    a, b = dev(A), dev(B)
    c = torch.full((M, N), float("nan"), device=DEV)
    bb = dev(bias) if bias is not None else None
    gg = dev(gate) if gate is not None else None
    if res is not None:
        c.copy_(dev(res))
    st = R.current_stream()
    if prec is not None:
        tile = (tile or 0) | ((R.PRECISIONS[prec] + 1) << 4)
    if tile is None:
        R.check(R.lib.rgm_gemm(R.ptr(a), K, R.ptr(b), K, R.ptr(c), N, M, N, K, R.ptr(bb), act, alpha,
                               R.ptr(gg), gate.shape[1] if gate is not None else 0, rows_per_gate,
                               R.ptr(c) if res is not None else None, N, st))
    else:
        R.check(R.lib.rgm_gemm_tile(R.ptr(a), K, R.ptr(b), K, R.ptr(c), N, M, N, K, R.ptr(bb), act, tile, st))
    torch.cuda.synchronize()
    return c.cpu().numpy()


def load_module(module, sd_np_or_torch):
    sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd_np_or_torch.items()}
    module.load_state_dict(sd, strict=True)
    return module.to(DEV).eval()


def fake_chord_backend(pr, given_key=None, return_key=False, fs=100., window_size=1.28):
    """Stand-in for the reference's music21 analyser with its signature (module level: the worker pool pickles it).  Encodes
    what it was given into its answer so that tests can check the plumbing: chords[w] = (sum of window w) % 7."""
    import numpy as np
    import torch
    assert pr.shape[0] == 128 and pr.dtype == np.intc and pr.min() >= 0 and pr.max() <= 127
    nw = int(pr.shape[-1] / fs / window_size)
    w = int(window_size * fs)
    out = {"chords": torch.tensor([int(pr[:, i * w:(i + 1) * w].sum()) % 7 for i in range(nw)], dtype=torch.long)}
    if return_key:
        out.update(key=int(pr.sum()) % 24, correlationCoefficient=float(pr.mean()))
    return out


def split_torch_dtype():
    """torch dtype of the hi / lo halves of a split row: bfloat16 in the default build, float16 in the RGM_SPLIT_F16 build"""
    from rgm import native as R
    return torch.float16 if int(R.lib.rgm_split_dtype()) == 1 else torch.bfloat16


SLOW_CHORD_SLEEP = 0.0      # seconds per excerpt of slow_chord_backend (set by the test that uses it; in-process backend only)


def slow_chord_backend(pr, given_key=None, return_key=False, fs=100., window_size=1.28):
    """fake_chord_backend that takes SLOW_CHORD_SLEEP seconds per excerpt, like a real symbolic analyser would (the sleep releases the GIL,
    as the reference's worker processes leave the sampling process free)."""
    import time
    time.sleep(SLOW_CHORD_SLEEP)
    return fake_chord_backend(pr, given_key=given_key, return_key=return_key, fs=fs, window_size=window_size)

"""-m gpu: building-block HIP kernels (through the C ABI) against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import dit_np as odit

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"


def _ref_gemm(A, B, bias, act, alpha, gate, rpg, res):
    y = alpha * (A.astype(np.float64) @ B.astype(np.float64).T)
    if bias is not None:
        y = y + bias
    if act == 1:
        y = y / (1 + np.exp(-y))
    elif act == 2:
        y = 0.5 * y * (1 + np.tanh(np.sqrt(2 / np.pi) * (y + 0.044715 * y ** 3)))
    if gate is not None:
        y = y * gate[np.arange(A.shape[0]) // rpg]
    if res is not None:
        y = y + res
    return y


@pytest.mark.parametrize("tile", [None, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(200, 96, 64), (513, 1152, 1152), (1000, 3456, 384), (16, 700, 256), (4096, 32, 1152)])
def test_gemm_asymmetric_data_all_tiles(tile, M, N, K):
    from gpu_util import gemm, rel
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K).astype(F32)
    B = (rng.randn(N, K) * (1 + np.arange(N)[:, None] / N)).astype(F32)     # rows of B differ in scale: catches transposes
    bias = rng.randn(N).astype(F32)
    out = gemm(A, B, bias=bias, act=0, tile=tile)
    assert rel(out, _ref_gemm(A, B, bias, 0, 1.0, None, 1, None)) < 2e-6


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(200, 96, 64), (513, 1152, 1152), (4096, 32, 1152), (777, 3456, 384)])
def test_gemm_bf16x3_split_precision(tile, M, N, K):
    """bf16x3 mode: three bf16 MFMAs per product, fp32 accumulate -- ~1e-5 of the output scale, not bf16's 3e-3."""
    from gpu_util import gemm, rel
    rng = np.random.RandomState(M + N + K + 1)
    A = rng.randn(M, K).astype(F32)
    B = (rng.randn(N, K) * (1 + np.arange(N)[:, None] / N)).astype(F32)
    bias = rng.randn(N).astype(F32)
    out = gemm(A, B, bias=bias, act=0, tile=tile, prec="bf16x3")
    ref = _ref_gemm(A, B, bias, 0, 1.0, None, 1, None)
    assert rel(out, ref) < 3e-5
    assert rel(out, ref) > 1e-8 or K < 64        # and it really is the split path, not the fp32 one


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(3 * 128 + 7, 1152, 1152), (1024, 1152, 1152), (1024, 3456, 1152), (4096, 1152, 1152), (2100, 3456, 384)])
def test_gemm_fused_epilogues(act, M, N, K):
    """bias, activation, alpha, per-sample gate and residual in the GEMM epilogue; the shapes walk the heuristic through its tile
    choices in pre-split mode (64x64 / 128x64 / 128x128 deep-ring loader-consumer kernels, 128x64 and 128x128 cross-iteration
    kernels), act 0 takes the 'linear' row body (gate / residual reads hoisted), 1 and 2 the general rolled one (gemm2.hip)."""
    from gpu_util import gemm, rel
    rng = np.random.RandomState(act + M)
    T = 128
    A = rng.randn(M, K).astype(F32) * 0.5
    B = rng.randn(N, K).astype(F32) * 0.05
    bias = rng.randn(N).astype(F32)
    gate = rng.randn((M + T - 1) // T, N).astype(F32)
    res = rng.randn(M, N).astype(F32)
    out = gemm(A, B, bias=bias, act=act, alpha=0.7, gate=gate, rows_per_gate=T, res=res)
    assert rel(out, _ref_gemm(A, B, bias, act, 0.7, gate, T, res)) < 3e-6


@pytest.mark.parametrize("D,affine,mod", [(1152, False, True), (384, False, True), (384, True, False), (768, True, True)])
def test_layernorm_modulate(D, affine, mod):
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(D)
    N, T = 3, 37
    x = (rng.randn(N * T, D) * 3 + 1).astype(F32)
    w, b = (1 + 0.1 * rng.randn(D)).astype(F32), rng.randn(D).astype(F32)
    modbuf = rng.randn(N, 6 * D).astype(F32)
    xd, od = dev(x), torch.empty(N * T, D, device="cuda")
    wd, bd, md = dev(w), dev(b), dev(modbuf)
    sh = md[:, D:] if mod else None         # views into a strided modulation buffer, like the DiT's
    R.check(R.lib.rgm_layernorm_modulate(R.ptr(xd), R.ptr(od), N * T, D, 1e-6, R.ptr(wd) if affine else None,
                                         R.ptr(bd) if affine else None,
                                         md.data_ptr() + 4 * D if mod else None, md.data_ptr() + 8 * D if mod else None,
                                         6 * D, T, R.current_stream()))
    torch.cuda.synchronize()
    ref = odit.layernorm(x, 1e-6, w if affine else None, b if affine else None).reshape(N, T, D)
    if mod:
        ref = ref * (1 + modbuf[:, None, 2 * D:3 * D]) + modbuf[:, None, D:2 * D]
    assert rel(od.cpu().numpy(), ref.reshape(N * T, D)) < 2e-6


@pytest.mark.parametrize("N,T,heads,hd", [(2, 256, 16, 72), (3, 128, 16, 72), (2, 257, 6, 64), (2, 129, 6, 64), (1, 256, 12, 64), (2, 200, 6, 64)])
def test_rotary_attention(N, T, heads, hd):
    from gpu_util import dev, rel
    from rgm import native as R
    from rgm.synth import rotary_freqs
    rng = np.random.RandomState(T + hd)
    D = heads * hd
    rot = int(hd * 0.5)
    qkv = (rng.randn(N * T, 3 * D) * 1.5).astype(F32)
    cos, sin = odit.rotary_tables(rotary_freqs(rot), T)
    od = torch.full((N * T, D), float("nan"), device="cuda")
    qd, cd, sd_ = dev(qkv), dev(cos), dev(sin)
    R.check(R.lib.rgm_rotary_attention(R.ptr(qd), R.ptr(od), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, rot // 2, R.current_stream()))
    torch.cuda.synchronize()
    r = qkv.reshape(N, T, 3, heads, hd)
    q, k, v = (r[:, :, i].transpose(0, 2, 1, 3).astype(np.float64) for i in range(3))
    q = odit.apply_rotary(q.astype(F32), cos, sin).astype(np.float64)
    k = odit.apply_rotary(k.astype(F32), cos, sin).astype(np.float64)
    s = q @ k.transpose(0, 1, 3, 2) * hd ** -0.5
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(N * T, D)
    assert rel(od.cpu().numpy(), ref) < 3e-6


@pytest.mark.parametrize("N,T", [(1, 256), (2, 256), (4, 256), (3, 200), (8, 256)])
def test_attention_queries_split_over_workgroups_give_identical_rows(N, T):
    """csrc/attention_x3.hip, key-blocked kernel (128 < T <= 256; ref guided_diffusion/dit.py:263-288): at small batches the queries of a
    (sample, head) are split over 2 or 4 workgroups (each stages all of K / V, its first 8 / qsplit waves own 32 queries each) so that
    the launch is not 16 N workgroups on 256 CUs.  Per query nothing changes -- the rows must be IDENTICAL to the unsplit launch, for
    every split (RGM_ATTN_QSPLIT is read per launch) and for the heuristic's own choice; a ragged T leaves the last waves without queries."""
    import os
    from gpu_util import dev
    from rgm import native as R
    from rgm.synth import rotary_freqs
    heads, hd = 16, 72
    rng = np.random.RandomState(N * 1000 + T)
    D = heads * hd
    rot = hd // 2
    qkv = dev((rng.randn(N * T, 3 * D) * 1.5).astype(F32))
    cos, sin = odit.rotary_tables(rotary_freqs(rot), T)
    cd, sd_ = dev(cos), dev(sin)
    R.set_gemm_precision("bf16x3_presplit")
    outs = {}
    old = os.environ.get("RGM_ATTN_QSPLIT")
    try:
        for qs in ("1", "2", "4", None):
            if qs is None:
                os.environ.pop("RGM_ATTN_QSPLIT", None)
            else:
                os.environ["RGM_ATTN_QSPLIT"] = qs
            od = torch.full((N * T, D), float("nan"), device="cuda")
            R.check(R.lib.rgm_rotary_attention(R.ptr(qkv), R.ptr(od), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, rot // 2, R.current_stream()))
            torch.cuda.synchronize()
            outs[qs] = od
    finally:
        if old is None:
            os.environ.pop("RGM_ATTN_QSPLIT", None)
        else:
            os.environ["RGM_ATTN_QSPLIT"] = old
        R.set_gemm_precision("fp32")
    assert bool(torch.isfinite(outs["1"]).all())
    for qs in ("2", "4", None):
        assert torch.equal(outs[qs], outs["1"]), qs


def _split(x):
    """numpy fp32 (R,K) -> split-row image via the device kernel, returned as a device tensor of the same shape."""
    from gpu_util import dev
    from rgm import native as R
    xd = dev(x)
    out = torch.empty_like(xd)
    R.check(R.lib.rgm_split_rows(R.ptr(xd), R.ptr(out), x.shape[0], x.shape[1], R.current_stream()))
    return out


def _unsplit(t):
    """split-row device tensor (R,K) -> numpy float64 hi+lo; a row is K/32 lines of [32 bf16 hi | 32 bf16 lo] (common.h)"""
    R_, K = t.shape
    raw = t.contiguous().view(__import__("gpu_util").split_torch_dtype()).view(R_, K // 32, 2, 32).float().cpu().numpy().astype(np.float64)
    return (raw[:, :, 0, :] + raw[:, :, 1, :]).reshape(R_, K)


def test_split_rows_format():
    rng = np.random.RandomState(0)
    x = (rng.randn(37, 96) * np.exp(rng.randn(37, 96) * 3)).astype(F32)
    back = _unsplit(_split(x))
    assert np.abs(back - x).max() / np.abs(x).max() < 2 ** -16
    assert (np.abs(back - x) <= np.abs(x) * 2.0 ** -15.5 + 1e-38).all()


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 21, 22, 43, 44, 45, 46, 51, 52, 53, 54, 55, 56, 57, 58, 71, 72, 73])
@pytest.mark.parametrize("M,N,K", [(200, 96, 64), (513, 1152, 1152), (4096, 36, 1152), (777, 3456, 384), (256, 128, 32)])
def test_gemm_split_dma_kernel(tile, M, N, K):
    """Pre-split operands + LDS-DMA staging (gemm2): same contraction, bf16x3 accuracy, every tile shape, ragged edges."""
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(M + N + K + tile)
    A = rng.randn(M, K).astype(F32)
    B = (rng.randn(N, K) * (1 + np.arange(N)[:, None] / N)).astype(F32)
    bias = rng.randn(N).astype(F32)
    As, Bs, bd = _split(A), _split(B), dev(bias)
    ref = _ref_gemm(A, B, bias, 0, 1.0, None, 1, None)
    c = torch.full((M, N), float("nan"), device="cuda")
    R.check(R.lib.rgm_gemm_split(R.ptr(As), R.ptr(Bs), R.ptr(c), M, N, K, R.ptr(bd), 0, tile, 0, R.current_stream()))
    torch.cuda.synchronize()
    assert rel(c.cpu().numpy(), ref) < 3e-5
    if N % 32 == 0:                                                  # split output feeds the next GEMM directly
        cs = torch.zeros((M, N), device="cuda")
        R.check(R.lib.rgm_gemm_split(R.ptr(As), R.ptr(Bs), R.ptr(cs), M, N, K, R.ptr(bd), 2, tile, 1, R.current_stream()))
        torch.cuda.synchronize()
        assert rel(_unsplit(cs), _ref_gemm(A, B, bias, 2, 1.0, None, 1, None)) < 3e-5


def test_split_rows_and_gemm_with_padded_row_strides():
    """rgm_split_rows_ld / rgm_gemm_split_ld: operands and output with padded rows (ld > K, ldc > N), ragged M and N."""
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(9)
    M, N, K, lda, ldb, ldc = 333, 160, 192, 192 + 32, 192 + 64, 160 + 32
    A, B, bias = rng.randn(M, K).astype(F32), rng.randn(N, K).astype(F32), rng.randn(N).astype(F32)
    As, Bs = torch.zeros(M, lda, device="cuda"), torch.zeros(N, ldb, device="cuda")
    st = R.current_stream()
    R.check(R.lib.rgm_split_rows_ld(R.ptr(dev(A)), K, R.ptr(As), lda, M, K, st))
    R.check(R.lib.rgm_split_rows_ld(R.ptr(dev(B)), K, R.ptr(Bs), ldb, N, K, st))
    assert rel(_unsplit(As[:, :K].contiguous()), A) < 2 ** -16
    C = torch.full((M, ldc), 7.0, device="cuda")
    for tile in (0, 43, 44, 52, 54, 56, 57, 71, 72, 73):
        C.fill_(7.0)
        R.check(R.lib.rgm_gemm_split_ld(R.ptr(As), lda, R.ptr(Bs), ldb, R.ptr(C), ldc, M, N, K, R.ptr(dev(bias)), 0, tile, 0, st))
        torch.cuda.synchronize()
        assert rel(C[:, :N].cpu().numpy(), _ref_gemm(A, B, bias, 0, 1.0, None, 1, None)) < 3e-5, tile
        assert bool((C[:, N:] == 7.0).all()), f"tile {tile} wrote into the row padding"


@pytest.mark.parametrize("tile", [48, 49])
@pytest.mark.parametrize("M,N,K,act,split", [(4096, 4608, 1152, 2, 1), (4096, 4608, 1152, 0, 0), (8192, 2304, 256, 1, 0)])
def test_gemm_big_and_small_tiles_in_one_launch(tile, M, N, K, act, split):
    """gemm2_dual_kernel (tiles 48 / 49; an explicit choice, see gemm2_launch): whole CU-rounds of 128x128 tiles plus the
    leftover columns on 128x64 / 64x64 tiles in the same launch -- every element is still ONE workgroup's K loop, so the result
    is bit-identical to the one-shape kernels wherever the per-element arithmetic is (same K order)."""
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(M + N + K + tile)
    A = rng.randn(M, K).astype(F32)
    B = (rng.randn(N, K) * 0.05).astype(F32)
    bias = rng.randn(N).astype(F32)
    As, Bs, bd = _split(A), _split(B), dev(bias)
    st = R.current_stream()
    c = torch.full((M, N), float("nan"), device="cuda")
    R.check(R.lib.rgm_gemm_split(R.ptr(As), R.ptr(Bs), R.ptr(c), M, N, K, R.ptr(bd), act, tile, split, st))
    c2 = torch.full((M, N), float("nan"), device="cuda")
    R.check(R.lib.rgm_gemm_split(R.ptr(As), R.ptr(Bs), R.ptr(c2), M, N, K, R.ptr(bd), act, 43, split, st))
    torch.cuda.synchronize()
    got = _unsplit(c) if split else c.cpu().numpy()
    assert rel(got, _ref_gemm(A, B, bias, act, 1.0, None, 1, None)) < 3e-5
    assert np.array_equal(got, _unsplit(c2) if split else c2.cpu().numpy())       # same per-element K order as the 128x128 kernel

"""-m gpu: parity at the sizes BASELINE configs 3-5 really run (round-2 review, "what's weak" 1 and 2).

The goldens pin batches of 2; the kernels that carry C3 (B = 32), C4 (n.B = 64) and C5 (14-window batches at XL width) are chosen by
M: the one-wave-per-SIMD kernels (gemm2.hip PIPE 5: 256x256 / 512x128 / 128x256 tiles, profiler ids 111..113 and 121 / 122 for the
implicit convs) only run at M >= 2048.  Covered here: their fused gate / residual epilogues against an fp64 product, the XL model at B = 32 / 64
against batches of 2 (with proof, from the library's own launch records, that those kernels ran), one SCG search step at
n.B = 64 against its 'rank r of 2' replay, and one guided DiffCollage step of config 5 at XL width."""
import ctypes as C
import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
XL2 = dict(depth=2, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)


def _ref(A, B, bias, act, alpha, gate, rpg, res):
    y = alpha * (A.astype(np.float64) @ B.astype(np.float64).T) + bias
    if act == 1:
        y = y / (1 + np.exp(-y))
    elif act == 2:
        y = 0.5 * y * (1 + np.tanh(np.sqrt(2 / np.pi) * (y + 0.044715 * y ** 3)))
    if gate is not None:
        y = y * gate[np.arange(A.shape[0]) // rpg]
    if res is not None:
        y = y + res
    return y


def _split(x):
    from gpu_util import dev
    from rgm import native as R
    xd = dev(x)
    out = torch.empty_like(xd)
    R.check(R.lib.rgm_split_rows(R.ptr(xd), R.ptr(out), x.shape[0], x.shape[1], R.current_stream()))
    return out


def _launches(ids):
    """{kernel id: launches} from the library's per-launch records (rgm_prof_*)."""
    from rgm import native as R
    out = {}
    for k in ids:
        n, ms, fl = C.c_int(0), C.c_double(0), C.c_double(0)
        R.check(R.lib.rgm_prof_report(k, C.byref(n), C.byref(ms), C.byref(fl)))
        out[k] = n.value
    return out


class _Recorded:
    def __enter__(self):
        from rgm import native as R
        R.check(R.lib.rgm_prof_reset())
        R.check(R.lib.rgm_prof_enable(1))
        return self

    def __exit__(self, *a):
        from rgm import native as R
        torch.cuda.synchronize()
        R.check(R.lib.rgm_prof_enable(0))
        self.n = _launches([83, 84, 111, 112, 113, 121, 122, 135])
        self.big = self.n[111] + self.n[112] + self.n[113]
        R.check(R.lib.rgm_prof_reset())


BIG_TILES = [71, 72, 73]


@pytest.mark.parametrize("tile", BIG_TILES)
@pytest.mark.parametrize("M,N,K,T", [(8192, 1152, 4608, 256), (16384, 1152, 1152, 256), (8192 + 200, 1152, 1152, 128), (4096, 1152, 4608, 256)])
def test_big_tile_kernels_gate_and_residual_in_place(tile, M, N, K, T):
    """proj / fc2 of a DiT block (dit.py:332-336: x = x + gate_b * (h W^T + bias), per-sample adaLN gate, residual read from and
    written to C) through the one-wave-per-SIMD kernels (71 / 72 / 73) on C3 / C4's shapes and one ragged M, against the fp64
    product; run twice on the same (uninitialised) scratch, bit-identical."""
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(M + N + K + tile)
    A, B = rng.randn(M, K).astype(F32), (rng.randn(N, K) * 0.03).astype(F32)
    bias, res = rng.randn(N).astype(F32), rng.randn(M, N).astype(F32)
    gate = rng.randn((M + T - 1) // T, N + 64).astype(F32)          # gate rows are strided like the modulation buffer's
    As, Bs, bd, gd = _split(A), _split(B), dev(bias), dev(gate)
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 4 * M * N * 4)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ws.fill_(0xAB)
    st = R.current_stream()
    outs = []
    for rep in range(2):
        x = dev(res)
        R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(x), N, M, N, K, R.ptr(bd), 0, 0.7, R.ptr(gd), N + 64, T,
                                         R.ptr(x), N, tile, 0, R.ptr(ws), need, st))
        torch.cuda.synchronize()
        outs.append(x.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert rel(outs[0], _ref(A, B, bias, 0, 0.7, gate[:, :N], T, res)) < 3e-5
    # the GELU + split-row output epilogue of fc1 through the same kernels
    if K == 1152 and M % 128 == 0:
        W = (rng.randn(2304, K) * 0.03).astype(F32)
        b2 = rng.randn(2304).astype(F32)
        h = torch.zeros(M, 2304, device="cuda")
        R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(_split(W)), K, R.ptr(h), 2304, M, 2304, K, R.ptr(dev(b2)), 2, 1.0, None, 0, 1,
                                         None, 0, tile, 1, R.ptr(ws), need, st))
        torch.cuda.synchronize()
        raw = h.view(__import__("gpu_util").split_torch_dtype()).view(M, 2304 // 32, 2, 32).float().cpu().numpy().astype(np.float64)
        assert rel((raw[:, :, 0] + raw[:, :, 1]).reshape(M, 2304), _ref(A, W, b2, 2, 1.0, None, 1, None)) < 3e-5


@pytest.mark.parametrize("M,N,K,T", [(1024, 4608, 1152, 256), (1000, 1152, 1152, 128), (512, 3456, 1152, 256), (768, 1152, 4608, 256),
                                     (4096, 1152, 1152, 256), (4096, 1152, 4608, 256),      # proj / fc2 at B = 16: one round, 8-row sweeps
                                     (2500, 1152, 1152, 128), (1300, 4608, 64, 128)])        # ragged last sweep; fewer K-tiles than the prefetch distance
def test_tile_144_kernel_epilogues_and_its_place_in_the_heuristic(M, N, K, T):
    """csrc/gemm144.hip (tile 81: 128x144 output tiles on v_mfma_f32_16x16x32, the product computed transposed so that a lane owns four
    consecutive columns): proj / fc2's gated in-place residual and fc1's GELU + split-row output against the fp64 product on the shapes of
    B = 2 .. 4 (one ragged M), bit-identical run to run; and through the heuristic (tile 0) -- fc1 at B = 4 is one round of 256 of
    these tiles, fc2 at B = 3 / 4 runs as K slices ON them, proj and fc2 at B = 16 are one round each (round 5: 8-row sweeps of the raster,
    operand lines prefetched into L2 by the consumer waves) -- the launch records prove the kernel took the call."""
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(M + N + K + 81)
    A, B = rng.randn(M, K).astype(F32), (rng.randn(N, K) * 0.03).astype(F32)
    bias, res = rng.randn(N).astype(F32), rng.randn(M, N).astype(F32)
    gate = rng.randn((M + T - 1) // T, N + 64).astype(F32)
    As, Bs, bd, gd = _split(A), _split(B), dev(bias), dev(gate)
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 8 * M * N * 4)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ws.fill_(0xAB)
    st = R.current_stream()
    want = _ref(A, B, bias, 0, 0.7, gate[:, :N], T, res)
    for tile in (81, 0):
        outs = []
        for rep in range(2):
            x = dev(res)
            R.check(R.lib.rgm_prof_reset())
            R.check(R.lib.rgm_prof_enable(1))
            R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(x), N, M, N, K, R.ptr(bd), 0, 0.7, R.ptr(gd), N + 64, T,
                                             R.ptr(x), N, tile, 0, R.ptr(ws), need, st))
            torch.cuda.synchronize()
            R.check(R.lib.rgm_prof_enable(0))
            n144 = _launches([135])[135]
            outs.append(x.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
        assert rel(outs[0], want) < 3e-5, (tile, rel(outs[0], want))
        if tile == 81:
            assert n144 == 1
        elif (M, N, K) in ((1024, 4608, 1152), (768, 1152, 4608), (4096, 1152, 1152), (4096, 1152, 4608)):
            assert n144 == 1, n144                                   # the heuristic's choice: one round / K slices of 128x144 tiles
    R.check(R.lib.rgm_prof_reset())
    h = torch.zeros(M, N, device="cuda")
    R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(h), N, M, N, K, R.ptr(bd), 2, 1.0, None, 0, 1, None, 0, 81, 1,
                                     R.ptr(ws), need, st))
    torch.cuda.synchronize()
    raw = h.view(__import__("gpu_util").split_torch_dtype()).view(M, N // 32, 2, 32).float().cpu().numpy().astype(np.float64)
    assert rel((raw[:, :, 0] + raw[:, :, 1]).reshape(M, N), _ref(A, B, bias, 2, 1.0, None, 1, None)) < 3e-5
    # SiLU, plain output, no bias (the remaining branches of the epilogue)
    c = torch.zeros(M, N, device="cuda")
    R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(c), N, M, N, K, None, 1, 1.0, None, 0, 1, None, 0, 81, 0,
                                     R.ptr(ws), need, st))
    assert rel(c.cpu().numpy(), _ref(A, B, np.zeros(N), 1, 1.0, None, 1, None)) < 3e-5


@pytest.mark.parametrize("M,N,K,T", [(3072, 1152, 4608, 128), (7168, 3456, 1152, 256), (3072, 4608, 1152, 128), (4096, 4608, 1152, 256),
                                     (4096, 1152, 4608, 256), (16384, 1152, 1152, 256), (7168, 1152, 4608, 256), (2048 + 256, 3456, 1152, 256),
                                     (2048, 4608, 1152, 256),      # fc1 at B = 8: 144 tiles, one partial round of the 256x256 kernel
                                     (28672, 1152, 1152, 256), (28672 + 128, 1152, 4608, 128)])   # C5's 112 windows: whole rounds of row tiles + the leftover rows
def test_heuristic_decompositions_of_the_big_tile_kernel(M, N, K, T):
    """gemm2_launch with tile 0 on the shapes the samplers produce (B = 16 / 64, the 28 + 24 windows of config 5): whole launches
    of 256x256 tiles, K slices + reduce, whole rounds of column tiles + the leftover columns through the heuristic again -- each with
    bias, alpha, per-sample gate and in-place residual -- against the fp64 product, and bit-identical run to run."""
    from gpu_util import dev, rel
    from rgm import native as R
    rng = np.random.RandomState(M + N + K)
    A, B = rng.randn(M, K).astype(F32), (rng.randn(N, K) * 0.03).astype(F32)
    bias, res = rng.randn(N).astype(F32), rng.randn(M, N).astype(F32)
    gate = rng.randn((M + T - 1) // T, N + 64).astype(F32)
    As, Bs, bd, gd = _split(A), _split(B), dev(bias), dev(gate)
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096 + 8 * M * N * 4)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = R.current_stream()
    outs = []
    for rep in range(2):
        x = dev(res)
        with _Recorded() as rec:
            R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(x), N, M, N, K, R.ptr(bd), 0, 0.7, R.ptr(gd), N + 64, T,
                                             R.ptr(x), N, 0, 0, R.ptr(ws), need, st))
        outs.append(x.cpu().numpy())
    if (M, N, K) == (4096, 1152, 4608):                             # fc2 at B = 16 (round 5): ONE round of 256 tiles of 128 x 144, unsliced
        assert rec.n[135] == 1 and rec.n[111] == 0, rec.n
    elif (M, N, K) not in ((7168, 1152, 4608), (2304, 3456, 1152)):   # (those two stay on the 128-row kernels: no full round of 256x256 tiles)
        assert rec.n[111] >= 1, rec.n                               # the 256x256 kernel took part
    assert np.array_equal(outs[0], outs[1])
    err = rel(outs[0], _ref(A, B, bias, 0, 0.7, gate[:, :N], T, res))
    assert err < 3e-5, (err, rec.n)
    # GELU + split-row output through the same decomposition (fc1)
    h = torch.zeros(M, N, device="cuda")
    R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(h), N, M, N, K, R.ptr(bd), 2, 1.0, None, 0, 1, None, 0, 0, 1,
                                     R.ptr(ws), need, st))
    torch.cuda.synchronize()
    raw = h.view(__import__("gpu_util").split_torch_dtype()).view(M, N // 32, 2, 32).float().cpu().numpy().astype(np.float64)
    assert rel((raw[:, :, 0] + raw[:, :, 1]).reshape(M, N), _ref(A, B, bias, 2, 1.0, None, 1, None)) < 3e-5


@pytest.mark.parametrize("tile", [71, 72, 73])
@pytest.mark.parametrize("rpg,gate_only,split_out", [(48, False, False), (40, True, False), (129, False, True), (32, False, False)])
def test_big_tile_epilogue_with_gate_rows_that_cut_through_slabs(tile, rpg, gate_only, split_out):
    """The gate / residual epilogue of the big tiles works slab by slab (32 rows) with the slab's two possible gate rows loaded up front and
    the residual rows brought in by LDS-DMA: gate periods that are not multiples of 32 (48, 40, the classifier's 129 tokens) put the switch
    inside slabs; also without a residual (zero-page DMA), with split-row output, and on a ragged M and N (partial tiles: guarded stores)."""
    from gpu_util import dev, rel
    from rgm import native as R
    M, N, K = 2048 + 72, 1152, 1152
    rng = np.random.RandomState(tile + rpg)
    A, B = rng.randn(M, K).astype(F32), (rng.randn(N, K) * 0.03).astype(F32)
    bias = rng.randn(N).astype(F32)
    res = None if gate_only else rng.randn(M, N).astype(F32)
    gate = rng.randn((M + rpg - 1) // rpg, N).astype(F32)
    As, Bs, bd, gd = _split(A), _split(B), dev(bias), dev(gate)
    st = R.current_stream()
    x = torch.zeros(M, N, device="cuda") if gate_only else dev(res)
    R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(x), N, M, N, K, R.ptr(bd), 0, 0.7, R.ptr(gd), N, rpg,
                                     None if gate_only else R.ptr(x), N, tile, 1 if split_out else 0, None, 0, st))
    torch.cuda.synchronize()
    ref = _ref(A, B, bias, 0, 0.7, gate, rpg, res)
    if split_out:
        raw = x.view(__import__("gpu_util").split_torch_dtype()).view(M, N // 32, 2, 32).float().cpu().numpy().astype(np.float64)
        got = (raw[:, :, 0] + raw[:, :, 1]).reshape(M, N)
    else:
        got = x.cpu().numpy()
    assert rel(got, ref) < 3e-5


def test_fine_grained_gate_stays_off_the_big_tiles():
    """The big tiles' gate / residual epilogue loads two gate rows per 32-row slab: a gate finer than 32 rows must keep the heuristic on the
    128-row kernels (right result, no 256x256 launch) and make an explicit big tile refuse."""
    from gpu_util import dev, rel
    from rgm import native as R
    M, N, K, rpg = 4096, 1152, 1152, 8
    rng = np.random.RandomState(11)
    A = rng.randn(M, K).astype(F32)
    B = (rng.randn(N, K) * 0.03).astype(F32)
    bias = rng.randn(N).astype(F32)
    gate = rng.randn(M // rpg, N).astype(F32)
    res = rng.randn(M, N).astype(F32)
    As, Bs = _split(A), _split(B)
    bd, gd = dev(bias), dev(gate)
    need = int(R.lib.rgm_gemm_scratch_bytes(M, N))
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    st = R.current_stream()
    x = dev(res)
    with _Recorded() as rec:
        R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(x), N, M, N, K, R.ptr(bd), 0, 1.0, R.ptr(gd), N, rpg,
                                         R.ptr(x), N, 0, 0, R.ptr(ws), need, st))
    assert rec.n[111] == 0, rec.n
    assert rel(x.cpu().numpy(), _ref(A, B, bias, 0, 1.0, gate, rpg, res)) < 3e-5
    status = R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(x), N, M, N, K, R.ptr(bd), 0, 1.0, R.ptr(gd), N, rpg,
                                      R.ptr(x), N, 71, 0, R.ptr(ws), need, st)
    assert status != 0 and b"rows_per_gate" in R.lib.rgm_last_error()
    torch.cuda.synchronize()


def _dit(arch, seed):
    from gpu_util import load_module
    from guided_diffusion.dit import DiTRotary
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=arch["hidden"], depth=arch["depth"],
                  num_heads=arch["heads"], num_classes=arch["num_classes"], learn_sigma=False)
    return load_module(m, synth.dit_state_dict(seed, final_std=0.3 / arch["hidden"] ** 0.5, device="cuda", **arch))


@pytest.mark.parametrize("B,depth", [(32, 2), (64, 2), (32, 28)])
def test_xl_model_at_c3_and_c4_batch_sizes_is_row_independent(B, depth, precision):
    """C3 runs the XL eps-network at B = 32 (M = 8192 rows), C4 at n.B = 64 (M = 16384): the sizes at which the heuristic hands the
    backbone GEMMs to the big-tile kernels.  Every sample of the big batch must equal the same sample in a batch of 2 (what the
    reference-generated goldens pin); in pre-split mode the launch records must show that the big-tile kernels really ran."""
    from gpu_util import dev, rel
    m = _dit(dict(XL2, depth=depth), 1)
    rng = np.random.RandomState(B + depth)
    x = dev(rng.randn(B, 4, 128, 16).astype(F32))
    t = dev(rng.randint(0, 1000, size=B).astype(np.int64))
    y = dev(rng.randint(0, 3, size=B).astype(np.int64))
    with _Recorded() as rec:
        big = m(x, t, y)
    if precision == "bf16x3_presplit":
        assert rec.big >= 2 * depth, rec.n
    assert bool(torch.isfinite(big).all())
    tol = 3e-5 if precision == "bf16x3_presplit" else 2e-6
    for i in (0, B // 2 - 1, B - 2):
        small = m(x[i:i + 2].contiguous(), t[i:i + 2].contiguous(), y[i:i + 2].contiguous())
        assert rel(big[i:i + 2].cpu().numpy(), small.cpu().numpy()) < tol * (4 if depth == 28 else 1), (i, rec.n)


@pytest.mark.parametrize("B,depth", [(2, 2), (5, 2), (8, 28), (19, 2), (32, 28), (64, 2)])
def test_blocks_as_two_half_batches_on_two_streams_equal_the_single_stream_forward(B, depth, precision):
    """rgm_set_dit_halves: the blocks of an eps-network forward as two half batches, the second on the handle's side stream (forked from /
    joined to the caller's stream by events).  Same values as the single-stream forward up to the tile choice (K slices of fc2 differ with
    M), identical from call to call, and ordered for the caller: the output is consumed on the caller's stream right behind the call."""
    from gpu_util import dev, rel
    from rgm import native as R
    m = _dit(dict(XL2, depth=depth), 3)
    rng = np.random.RandomState(100 + B)
    x = dev(rng.randn(B, 4, 128, 16).astype(F32))
    t = dev(rng.randint(0, 1000, size=B).astype(np.int64))
    y = dev(rng.randint(0, 3, size=B).astype(np.int64))
    prev = C.c_int(0)
    R.check(R.lib.rgm_set_dit_halves(0, C.byref(prev)))
    try:
        assert prev.value == -1        # the default: the measured batch sizes
        one = m(x, t, y).clone()
        R.check(R.lib.rgm_set_dit_halves(2, None))
        two = [(m(x, t, y) * 1.0).clone() for _ in range(3)]
    finally:
        R.check(R.lib.rgm_set_dit_halves(prev.value, None))
    assert bool(torch.isfinite(two[0]).all())
    assert torch.equal(two[0], two[1]) and torch.equal(two[0], two[2])
    tol = 3e-5 if precision == "bf16x3_presplit" else 2e-6
    assert rel(two[0].cpu().numpy(), one.cpu().numpy()) < tol * (4 if depth == 28 else 1)
    if precision != "bf16x3_presplit":      # the other arithmetics keep one stream: bit-identical
        assert torch.equal(two[0], one)


@pytest.mark.parametrize("B", [4, 8, 12, 16])
def test_fc2_reduce_with_the_next_layernorm_equals_the_two_kernels(B):
    """fc2 of a block runs as K slices at B = 4 / 8 / 12 (M = 1024 .. 3072 rows; B = 16 went to ONE round of 128 x 144 tiles in round 5 and
    keeps its LayerNorm apart: the counter must not move there); the kernel that reduces the slices holds whole rows
    and also writes the next block's adaLN-LayerNorm (GemmParams::ln_out, csrc/gemm2.hip; ref guided_diffusion/dit.py:334-336).
    Same arithmetic in the same order as the separate LayerNorm launch: the outputs must be IDENTICAL with the fusion on and off,
    and the library's counter must show that the fused route really ran (depth - 1 launches per forward and half batch)."""
    from gpu_util import dev
    from rgm import native as R
    R.set_gemm_precision("bf16x3_presplit")          # K slices exist in the pre-split arithmetic only
    depth = 4
    m = _dit(dict(XL2, depth=depth), 3)
    rng = np.random.RandomState(B)
    x = dev(rng.randn(B, 4, 128, 16).astype(F32))
    t = dev(rng.randint(0, 1000, size=B).astype(np.int64))
    y = dev(rng.randint(0, 3, size=B).astype(np.int64))
    try:
        R.check(R.lib.rgm_set_fuse_reduce_ln(0))
        n0 = R.lib.rgm_fused_reduce_ln_launches()
        apart = m(x, t, y).clone()
        assert R.lib.rgm_fused_reduce_ln_launches() == n0
        R.check(R.lib.rgm_set_fuse_reduce_ln(1))
        fused = m(x, t, y).clone()
        parts = {8: 2, 16: 0}.get(B, 1)  # B = 8 runs as two half batches (rgm_set_dit_halves' default rule): every half's fc2 reduces its own rows
        proj = depth if B == 4 else 0    # round 6: at B = 4 proj is K-sliced too (64 tiles x 4 slices) and its reduce writes the block's SECOND LayerNorm
        assert R.lib.rgm_fused_reduce_ln_launches() == n0 + parts * (depth - 1) + proj, "fc2 / proj did not take the K-slice route with the fused LayerNorm"
    finally:
        R.check(R.lib.rgm_set_fuse_reduce_ln(1))
        R.set_gemm_precision("fp32")
    assert bool(torch.isfinite(fused).all())
    assert torch.equal(fused, apart), float((fused - apart).abs().max())


def _vae(seed=2):
    from gpu_util import load_module
    from taming.models.klvae_pedal import AutoencoderKL
    return load_module(AutoencoderKL(), synth.vae_state_dict(seed, encoder=True))


def _diffusion(rs=""):
    from guided_diffusion.script_util import create_diffusion
    return create_diffusion(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing=rs,
                            use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def _targets(B, nw):
    ph = torch.tensor([0.5, 0, 0, 0, 0.25, 0, 0, 0.25, 0, 0, 0, 0], device="cuda").repeat(B, 1)
    nd = torch.tensor([3.] * (2 * nw), device="cuda").repeat(B, 1)
    return {"pitch_hist": ph, "note_density": nd}


def _c4_classifiers():
    """the pitch-histogram (12 outputs) and note-density (16) DiTRotary-S/8-cls of scg_classifier_all.yml, synthetic weights (seeds 5 / 3
    as in tests/golden/make_golden.py g_round4 and bench.py), scales 400 / 10"""
    from functools import partial
    from gpu_util import load_module
    from rgm import synth
    from guided_diffusion.dit import DiTRotaryClassifier
    from guided_diffusion.condition_functions import composite_nn_zt
    clfs = []
    for k, seed in ((12, 5), (16, 3)):
        arch = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=k)
        clfs.append(load_module(DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=12, num_heads=6,
                                                    num_classes=k), synth.dit_state_dict(seed, **arch)))
    return partial(composite_nn_zt, fns=["grad_nn_zt_mse", "grad_nn_zt_mse"], classifier_scales=[400., 10.], classifiers=clfs,
                   rule_names=["pitch_hist", "note_density"])


def test_scg_search_step_at_c4_size_matches_its_two_rank_replay(monkeypatch, precision):
    """BASELINE config 4 as scg_classifier_all.yml defines it (minus the chord rule) at its real candidate batch: classifier guidance
    with the pitch and note-density classifiers AND SCG, B = 4, n = 16 -> the eps-network scores 64 rows (M = 16384), the decoder 512
    squares.  The step is run unsharded and replayed as 'rank r of 2' (32 candidates each, other half of the log-prob table from a
    stand-in all-gather): same per-sample winners, log-probs equal up to the batch-size dependence of the GEMM tiles, and the
    rebuilt winner bit-identical.  (The same step against the REFERENCE's values: tests/test_gpu_round4.py.)"""
    from functools import partial
    from types import SimpleNamespace
    from gpu_util import rel
    from rgm import scg_shard
    from guided_diffusion.condition_functions import model_fn
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    B, n = 4, 16
    m, vae = _dit(XL2, 1), _vae(2)
    fn = partial(model_fn, model=m, num_classes=3, class_cond=True, cfg=False, w=0.)
    kw = {"y": torch.ones(B, dtype=torch.int64, device="cuda"), "rule": _targets(B, 8)}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="classifier_guidance")
    cond = _c4_classifiers()
    scg = {"num_samples": n, "pitch_hist": 40., "note_density": 1.}
    x = torch.randn(B, 4, 128, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    t = torch.full((B,), 600, dtype=torch.int64, device="cuda")

    def run():
        d = _diffusion("")
        d.t_end = 0
        d.noise = PhiloxNoise(seed=99)
        out = d.p_sample(fn, x, t, clip_denoised=False, cond_fn=cond, model_kwargs=kw, embed_model=vae, scale_factor=1.2465,
                         guidance_kwargs=guid, scg_kwargs=scg)
        return out["sample"], d.last_scg["total_log_prob"].clone(), d.last_scg["max_ind"].clone()

    with _Recorded() as rec:
        ref_sample, ref_total, ref_idx = run()
    if precision == "bf16x3_presplit":
        assert rec.big >= 4, rec.n
    assert ref_total.shape == (n, B) and bool(torch.isfinite(ref_total).all())
    spread = float((ref_total.max(0).values - ref_total.min(0).values).min())
    assert spread > 1e-3 * float(ref_total.abs().max()), "degenerate search: every candidate scores the same"
    h = n // 2
    for rank in (0, 1):
        monkeypatch.setattr(scg_shard, "partition", lambda n_, r=rank: (r * n_ // 2, n_ // 2, True))

        def fake_gather(local, r=rank):
            assert rel(local.cpu().numpy(), ref_total[r * h:(r + 1) * h].cpu().numpy()) < 1e-4
            parts = [ref_total[:h], ref_total[h:]]
            parts[r] = local
            return torch.cat(parts, dim=0)
        monkeypatch.setattr(scg_shard, "gather_totals", fake_gather)
        s, total, idx = run()
        assert torch.equal(idx, ref_idx), f"rank {rank}: other winners"
        assert torch.equal(s, ref_sample), f"rank {rank}: rebuilt winner differs"


@pytest.fixture(params=[("fp32", 2), ("bf16x3", 2), ("bf16x3_presplit", 2), ("bf16x3_presplit", 28)], ids=lambda p: f"{p[0]}-d{p[1]}")
def prec_depth(request):
    """every arithmetic at depth 2, depth 28 once in the arithmetic the bench runs"""
    from rgm import native as R
    R.set_gemm_precision(request.param[0])
    yield request.param
    R.set_gemm_precision("fp32")


def test_long_sequence_guided_step_at_xl_width(monkeypatch, prec_depth):
    """BASELINE config 5 (diff_collage/condind_long.py:24-51 + gaussian_diffusion.py:562-592): a 4 x 512 x 16 latent = 7 windows + 6
    overlap halves through the XL eps-network per evaluation, 32 squares per candidate through the decoder, segment-wise SCG
    (dc.base 128), B = 1, n = 4.  (a) the collage eps of the 13-window batch equals the same windows evaluated one pair at a time
    (row independence at M = 13 x 256 .. 4 x 13 x 256 rows); (b) the guided step equals its 'rank r of 2' replay."""
    precision, depth = prec_depth
    from functools import partial
    from types import SimpleNamespace
    from gpu_util import rel
    import diff_collage as dc
    from rgm import scg_shard
    from guided_diffusion.condition_functions import dc_model_fn
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    B, n = 1, 4
    m, vae = _dit(dict(XL2, depth=depth), 1), _vae(2)
    calls = []

    def eps_fn(x, t, y=None):
        calls.append(tuple(x.shape))
        return m(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)

    def eps_pairs(x, t, y=None):                        # the same windows, two at a time
        outs = []
        for i in range(0, x.shape[0], 2):
            outs.append(m(x[i:i + 2].permute(0, 1, 3, 2).contiguous(), t[i:i + 2].contiguous(),
                          y=None if y is None else y[i:i + 2].contiguous()).permute(0, 1, 3, 2))
        return torch.cat(outs, 0)
    worker = dc.CondIndSimple((4, 16, 128), eps_fn, 7, overlap_size=64)
    small = dc.CondIndSimple((4, 16, 128), eps_pairs, 7, overlap_size=64)
    assert worker.shape == (4, 16, 512)
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(4 * B, 4, 16, 512, device="cuda", generator=g)
    tt = torch.full((4 * B,), 500, dtype=torch.int64, device="cuda")
    yy = torch.ones(4 * B, dtype=torch.int64, device="cuda")
    e_big = worker.eps_scalar_t_fn(w, tt, y=yy)
    e_small = small.eps_scalar_t_fn(w, tt, y=yy)
    assert max(c[0] for c in calls) >= 4 * 7, calls                    # the windows of all samples really went through as one batch
    tol = (3e-5 if precision == "bf16x3_presplit" else 2e-6) * (4 if depth == 28 else 1)
    assert rel(e_big.cpu().numpy(), e_small.cpu().numpy()) < tol

    fn = partial(dc_model_fn, model=worker.eps_scalar_t_fn, num_classes=3, class_cond=True, cfg=False, w=0.)
    x = torch.randn(B, 4, 512, 16, device="cuda", generator=g)
    t = torch.full((B,), 600, dtype=torch.int64, device="cuda")
    kw = {"y": torch.ones(B, dtype=torch.int64, device="cuda"), "rule": _targets(B, 32)}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance", dc=SimpleNamespace(base=128))
    scg = {"num_samples": n, "pitch_hist": 40., "note_density": 1.}

    def run():
        d = _diffusion("")
        d.t_end = 0
        d.noise = PhiloxNoise(seed=17)
        out = d.p_sample(fn, x, t, clip_denoised=False, model_kwargs=kw, embed_model=vae, scale_factor=1.2465,
                         guidance_kwargs=guid, scg_kwargs=scg)
        return out["sample"], d.last_scg["total_log_prob"].clone(), d.last_scg["max_ind"].clone()

    ref_sample, ref_total, ref_idx = run()
    S = 512 // 128
    assert ref_sample.shape == (B, 4, 512, 16) and ref_total.shape == (n, S, B) and ref_idx.shape == (S, B)
    assert bool(torch.isfinite(ref_sample).all())
    h = n // 2
    for rank in (0, 1):
        monkeypatch.setattr(scg_shard, "partition", lambda n_, r=rank: (r * n_ // 2, n_ // 2, True))

        def fake_gather(local, r=rank):
            mine = ref_total[r * h:(r + 1) * h].reshape(h, -1)
            assert rel(local.cpu().numpy(), mine.cpu().numpy()) < 1e-4
            parts = [ref_total[:h].reshape(h, -1), ref_total[h:].reshape(h, -1)]
            parts[r] = local
            return torch.cat(parts, dim=0)
        monkeypatch.setattr(scg_shard, "gather_totals", fake_gather)
        s, total, idx = run()
        # winners must agree wherever the unsharded table separates its two best candidates by more than the batch-size dependence of
        # the scores (a rank scores 2 candidates per forward, the unsharded step 4: other GEMM tiles, ~1e-5 relative); a segment whose
        # two best candidates tie within that is allowed to flip (depth 28 with random weights produces such ties)
        top2 = torch.topk(ref_total, 2, dim=0).values
        clear = (top2[0] - top2[1]) > 2e-4 * ref_total.abs().max()
        assert torch.equal(idx[clear], ref_idx[clear]), f"rank {rank}: other per-segment winners"
        assert int(clear.sum()) >= clear.numel() // 2
        # the rebuilt sample, segment by segment (128 latent rows each): bit-identical wherever the winners agree -- also when some OTHER
        # segment flipped on a tie; a flipped segment must hold exactly the other candidate (not a mixture): it differs there
        same = (idx == ref_idx)
        for seg in range(S):
            for b in range(B):
                a, r_ = s[b, :, seg * 128:(seg + 1) * 128], ref_sample[b, :, seg * 128:(seg + 1) * 128]
                if bool(same[seg, b]):
                    assert torch.equal(a, r_), f"rank {rank}: segment {seg} of sample {b} has the same winner but other values"
                else:
                    assert not bool(clear[seg, b]) and not torch.equal(a, r_)


_DECODE_SNIPPET = """
import sys, numpy as np, torch
sys.path[:0] = [{pkg!r}, {tests!r}]
from rgm import native as R, synth
from gpu_util import load_module
from taming.models.klvae_pedal import AutoencoderKL
from guided_diffusion.gaussian_diffusion import _decode
R.set_gemm_precision("bf16x3_presplit")
R.check(R.lib.rgm_set_big_tiles(1, {min_tiles}))
vae = load_module(AutoencoderKL(), synth.vae_state_dict(2, encoder=True))
z = torch.from_numpy(np.random.RandomState(5).randn({n}, 4, 128, 16).astype(np.float32) * 0.8).cuda()
np.save({out!r}, _decode(z, vae, scale_factor=1.2465).cpu().numpy())
"""


@pytest.mark.parametrize("n,min_tiles", [(2, 256), (2, 1)])
def test_decoder_1x1_convs_on_the_presplit_gemm_match_the_fp32_operand_kernel(tmp_path, n, min_tiles):
    """Pre-split arithmetic: the decoder's 1x1 convs (attention q / k / v / proj_out, nin_shortcut) run on the LDS-DMA GEMM from split-row
    weight copies, a channel-changing ResnetBlock computes its shortcut first, in x's own buffer (csrc/vae.hip; ref taming model.py:117-137,
    :140-192).  RGM_VAE_SPLIT_1X1=0 keeps them on the fp32-operand kernel and the old order: the decoded rolls must agree -- with the
    heuristic tiles of a small input and with the big-tile kernels forced (rgm_set_big_tiles(1, 1))."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"roll_{flag}.npy")
        code = _DECODE_SNIPPET.format(pkg=os.path.join(root, "rule-guided-music_amd"), tests=os.path.join(root, "tests"), n=n,
                                      min_tiles=min_tiles, out=out)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, RGM_VAE_SPLIT_1X1=flag))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    from gpu_util import rel
    assert np.isfinite(outs[1]).all()
    assert rel(outs[1], outs[0]) < 2e-5, rel(outs[1], outs[0])


def test_two_rank_scg_bench_control_flow_on_one_device():
    """The command the 2-GPU test runs, with both ranks on this one device over gloo (RGM_BENCH_ONE_DEVICE=1: plumbing, never a
    measurement): the sharded search step, both collectives, the same-winners check and the JSON line's shape."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGM_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-extras", "--workload", "scg", "--steps", "2",
                          "--warmup", "1", "--repeats", "1"], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "PLUMBING" in line["data"]
    assert line["config"]["same_winners_on_every_rank"] is True, line


def test_forward_time_has_no_cliff_between_neighbouring_batch_sizes():
    """gemm2_launch's tile choice is a hand-tuned ladder fitted to B = 16 / 32 / 64 and C5's window batches: the
    XL-28 forward is timed over the batch sizes in between and the cost PER SAMPLE may not rise by more than 10 % from one batch size to
    the next larger one (tools/batch_sweep.py prints the table; round 4: the worst step is B = 24 -> 28 at +9 %).  A pair over the bar is
    measured once more before it counts (boxes drift by a few percent within a run)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import batch_sweep
    Bs = [2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64]
    rows = dict(batch_sweep.sweep(Bs))
    per = {B: rows[B] / B for B in Bs}
    for a, b in zip(Bs[:-1], Bs[1:]):
        ratio = per[b] / per[a]
        for _ in range(4):              # a pair over the bar is measured again, up to four times (a box right after other tests drifts by a few percent: round 5 saw one in-suite failure of a pair that reads x1.00 alone)
            if ratio <= 1.10:
                break
            again = dict(batch_sweep.sweep([a, b], reps=15))
            ratio = min(ratio, (again[b] / b) / (again[a] / a))
        assert ratio <= 1.10, f"per-sample cost rises x{ratio:.3f} from B = {a} to B = {b}: {rows}"
    assert per[64] < per[16] < per[4] < per[2]


def test_eight_rank_scg_bench_control_flow_on_one_device():
    """The north star's topology -- n = 16 candidates over EIGHT ranks (2 each), B = 4 < R: the x_t forward and the two classifiers'
    gradients shared out one row per rank (`R % B == 0` rule of batch_shard.partition_rows), one all-gather of eps + gradient rows, one
    of the (2, 4) log-prob tables -- run as eight processes on this one device over gloo (plumbing, never a measurement).  The step is
    C4 as the reference's scg_classifier_all.yml defines it (classifier guidance AND SCG); every rank must pick the same winners."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGM_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--no-extras", "--workload", "scg", "--steps", "2",
                          "--warmup", "1", "--repeats", "1"], capture_output=True, text=True, timeout=2400, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and "PLUMBING" in line["data"]
    assert "classifier-guided" in line["config"]["workload"]
    assert line["config"]["same_winners_on_every_rank"] is True, line


def test_eight_rank_long_sequence_bench_control_flow_on_one_device():
    """BASELINE config 5's topology: ONE 4 x 512 x 16 sample, n = 16 candidates over eight ranks.  The x_t forward has no rows to share
    out, so the linear collage shares out its 7 + 6 windows (batch_shard.WINDOW_SHARD, one all-reduce of the window eps); candidates two
    per rank, per-segment winners from one all-gather.  Eight processes on this one device over gloo (plumbing, never a measurement):
    every rank must pick the same per-segment winners."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGM_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--no-extras", "--workload", "long", "--batch", "1", "--steps", "1",
                          "--warmup", "1", "--repeats", "1"], capture_output=True, text=True, timeout=2400, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and "PLUMBING" in line["data"]
    assert line["config"]["same_winners_on_every_rank"] is True, line


def test_two_gpu_scg_bench_runs_over_rccl_when_the_box_has_two_gpus():
    """Multi-GPU readiness: on a box with >= 2 GPUs this runs the sharded SCG bench over RCCL (bench.py spawns its own ranks) and
    requires the same winners on every rank; on the 1-GPU boxes of this round it is skipped -- the only test that may skip."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL); the gloo two-rank tests in test_host_logic.py cover the protocol on CPU")
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-extras", "--workload", "scg", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=1500, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["same_winners_on_every_rank"] is True, line


@pytest.mark.parametrize("N,T,heads,hd", [(48, 128, 16, 72), (40, 96, 16, 72), (64, 64, 6, 64), (48, 129, 6, 64)])
def test_short_sequence_attention_is_right_in_every_launch(N, T, heads, hd, precision):
    """Config 5's half windows are 128-token sequences, dozens of them per batch: K and V of a head then take <= 80 KiB of LDS and a
    second workgroup used to move in beside the first one's last waves -- and the one running in the upper half of the LDS sporadically
    returned wrong rows (found in round 3 through the row-independence test of the 13-window batch; tools/race_block.py).  The
    launchers now keep one workgroup per CU.  30 launches on the same input: bit-identical, and right against the fp64 product."""
    from gpu_util import dev
    from rgm import native as R
    from rgm.synth import rotary_freqs
    from oracle import dit_np as odit
    rng = np.random.RandomState(N + T)
    D = heads * hd
    rot = hd // 2
    qkv = (rng.randn(N * T, 3 * D) * 1.5).astype(F32)
    cos, sin = odit.rotary_tables(rotary_freqs(rot), T)
    qd, cd, sd_ = dev(qkv), dev(cos), dev(sin)
    outs = []
    for _ in range(30):
        od = torch.full((N * T, D), float("nan"), device="cuda")
        R.check(R.lib.rgm_rotary_attention(R.ptr(qd), R.ptr(od), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, rot // 2, R.current_stream()))
        outs.append(od)
    torch.cuda.synchronize()
    worst = max(float((o - outs[0]).abs().max()) for o in outs)
    assert worst == 0.0, f"launches of one input differ by {worst}"
    r = qkv.reshape(N, T, 3, heads, hd)
    q, k, v = (r[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    q = odit.apply_rotary(q.astype(F32), cos, sin).astype(np.float64)
    k = odit.apply_rotary(k.astype(F32), cos, sin).astype(np.float64)
    s = q @ k.transpose(0, 1, 3, 2) * hd ** -0.5
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v.astype(np.float64)).transpose(0, 2, 1, 3).reshape(N * T, D)
    err = np.abs(outs[0].cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < (3e-6 if precision == "fp32" else 3e-5), err


@pytest.mark.parametrize("N,T,heads,hd,rot", [(40, 160, 6, 64, 4), (24, 256, 6, 64, 8), (24, 130, 16, 72, 4), (3, 257, 6, 64, 32)])
def test_every_attention_launcher_keeps_one_workgroup_per_cu(N, T, heads, hd, rot, precision):
    """two workgroups of an attention kernel on one CU are a correctness hazard, so every launcher goes through
    attn_prepare_kernel (common.h), which pads the LDS request past half of the 160 KiB and REQUIRES an occupancy of one from the
    runtime -- a launch that could co-reside returns an error instead of running.  The key-blocked kernel's request shrinks with the rotary
    table (head_dim 64 with 4 rotary channels at T = 160 would fit twice without the padding: the advisor's case); the backward kernels
    take the same helper.  Here: those shapes run, repeat bit for bit, and are right against the fp64 product."""
    from gpu_util import dev
    from rgm import native as R
    from rgm.synth import rotary_freqs
    from oracle import dit_np as odit
    rng = np.random.RandomState(N * 1000 + T)
    D = heads * hd
    qkv = (rng.randn(N * T, 3 * D) * 1.2).astype(F32)
    cos, sin = odit.rotary_tables(rotary_freqs(rot), T)
    qd, cd, sd_ = dev(qkv), dev(cos), dev(sin)
    outs = []
    for _ in range(8):
        od = torch.full((N * T, D), float("nan"), device="cuda")
        R.check(R.lib.rgm_rotary_attention(R.ptr(qd), R.ptr(od), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, rot // 2, R.current_stream()))
        outs.append(od)
    torch.cuda.synchronize()
    assert max(float((o - outs[0]).abs().max()) for o in outs) == 0.0
    r = qkv.reshape(N, T, 3, heads, hd)
    q, k, v = (r[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    q = odit.apply_rotary(q.astype(F32), cos, sin).astype(np.float64)
    k = odit.apply_rotary(k.astype(F32), cos, sin).astype(np.float64)
    s = q @ k.transpose(0, 1, 3, 2) * hd ** -0.5
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v.astype(np.float64)).transpose(0, 2, 1, 3).reshape(N * T, D)
    err = np.abs(outs[0].cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < (3e-6 if precision == "fp32" else 3e-5), err
    # the backward launchers (fp32 MFMA kernels) on the same shape: the helper's occupancy requirement must hold for them too
    lse = torch.zeros(N * heads * T, device="cuda")
    do = dev(rng.randn(N * T, D).astype(F32))
    dq = torch.empty(N * T, 3 * D, device="cuda")
    R.check(R.lib.rgm_rotary_attention_bwd(R.ptr(qd), R.ptr(outs[0]), R.ptr(do), R.ptr(lse), R.ptr(dq), R.ptr(cd), R.ptr(sd_), N, T, heads, hd,
                                           rot // 2, R.current_stream()))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dq).all())


@pytest.mark.parametrize("N,T", [(96, 128), (48, 128), (40, 96), (13, 128)])
def test_short_sequence_attention_with_two_workgroups_per_cu(N, T):
    """rgm_set_attn_pairs(1): the head_dim-72 attention at T <= 128 with TWO workgroups per CU -- the co-residency under which the round-3
    kernel returned wrong rows and which the two-phase Q prologue of round 5 makes safe (tools/ubench/attn_hazard: 0 wrong
    workgroups of 4.6 million).  30 launches per shape, every one bit-identical to the guarded (one per CU) launch."""
    from gpu_util import dev
    from rgm import native as R
    from rgm.synth import rotary_freqs
    from oracle import dit_np as odit
    R.set_gemm_precision("bf16x3_presplit")
    try:
        heads, hd, rot = 16, 72, 36
        rng = np.random.RandomState(N + T)
        D = heads * hd
        qd = dev((rng.randn(N * T, 3 * D) * 1.2).astype(F32))
        cos, sin = odit.rotary_tables(rotary_freqs(rot), T)
        cd, sd_ = dev(cos), dev(sin)

        def run():
            od = torch.full((N * T, D), float("nan"), device="cuda")
            R.check(R.lib.rgm_rotary_attention(R.ptr(qd), R.ptr(od), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, rot // 2, R.current_stream()))
            return od
        guarded = run()
        prev = R.lib.rgm_set_attn_pairs(1)
        try:
            assert prev == 0
            for k in range(30):
                assert torch.equal(run(), guarded), k
        finally:
            R.lib.rgm_set_attn_pairs(prev)
    finally:
        R.set_gemm_precision("fp32")


@pytest.mark.parametrize("N,T,heads,hd", [(4, 257, 6, 64), (32, 257, 6, 64), (3, 256, 16, 72), (5, 128, 16, 72), (2, 161, 6, 64)])
def test_per_tile_attention_workgroups_match_the_per_head_ones(N, T, heads, hd, precision):
    """rgm_set_attn_split: the classifier path's attention (257 tokens = 9 tiles on 8 waves, 6 x B (sample, head) pairs on 256 CUs) runs
    one workgroup per (sample, head, tile) in the backward -- the eight waves share the tile's key / query loop, partial dQ / dK / dV tiles
    meet in LDS and are summed in wave order -- and two workgroups per (sample, head) in the forward.  Same products, another summation
    order of the eight partials: forward bit-identical, backward to 3e-6 of the gradient's scale (4e-5 in the bf16x3 modes, see below); both repeat bit for bit.  At T = 257 the
    per-head launch (mode 0) also takes the lone-token paths -- the 257th token's rows as plain FMA sums of 257 terms and a rank-1 update
    instead of a ninth MFMA tile -- which this comparison covers against the all-MFMA per-tile kernels."""
    from gpu_util import dev
    from rgm import native as R
    from rgm.synth import rotary_freqs
    from oracle import dit_np as odit
    rng = np.random.RandomState(N + T)
    D = heads * hd
    rot = hd // 2
    qkv = dev((rng.randn(N * T, 3 * D) * 1.2).astype(F32))
    do = dev(rng.randn(N * T, D).astype(F32))
    cos, sin = odit.rotary_tables(rotary_freqs(rot), T)
    cd, sd_ = dev(cos), dev(sin)
    st = R.current_stream()
    res = {}
    try:
        for mode in (0, 1, 1):
            R.check(R.lib.rgm_set_attn_split(mode))
            o = torch.full((N * T, D), float("nan"), device="cuda")
            lse = torch.full((N * heads * T,), float("nan"), device="cuda")
            R.check(R.lib.rgm_rotary_attention_lse(R.ptr(qkv), R.ptr(o), R.ptr(lse), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, rot // 2, st))
            dq = torch.full((N * T, 3 * D), float("nan"), device="cuda")
            R.check(R.lib.rgm_rotary_attention_bwd(R.ptr(qkv), R.ptr(o), R.ptr(do), R.ptr(lse), R.ptr(dq), R.ptr(cd), R.ptr(sd_), N, T, heads, hd,
                                                   rot // 2, st))
            torch.cuda.synchronize()
            res.setdefault(mode, []).append((o, lse, dq))
    finally:
        R.check(R.lib.rgm_set_attn_split(-1))
    (o0, l0, g0), (o1, l1, g1), (o2, l2, g2) = res[0][0], res[1][0], res[1][1]
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(g1, g2)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    assert bool(torch.isfinite(g1).all())
    # fp32 mode: the same exact products in another order.  bf16x3 modes (round 6: the backward's products are split hi + lo like the forward's):
    # at T = 257 the per-head launch computes the lone token's rows with plain fp32 FMAs where the per-tile kernels run a ninth x3 MFMA tile
    # -- those terms differ by the split's 2^-16 per product
    assert float((g0 - g1).abs().max()) <= (3e-6 if precision == "fp32" else 4e-5) * float(g0.abs().max())


def test_half_window_batches_through_the_xl_model_are_deterministic(precision):
    """The same hazard at the model level: 48 half windows (H = 64 -> 128 tokens) through XL depth 4, twelve times: identical outputs,
    equal to the batches-of-2 evaluation."""
    from gpu_util import dev, rel
    m = _dit(dict(XL2, depth=4), 1)
    rng = np.random.RandomState(48)
    N = 48
    x = dev(rng.randn(N, 4, 64, 16).astype(F32))
    t = dev(np.full((N,), 500, dtype=np.int64))
    y = dev(np.ones((N,), dtype=np.int64))
    outs = [m(x, t, y).clone() for _ in range(12)]
    worst = max(float((o - outs[0]).abs().max()) for o in outs)
    assert worst == 0.0, f"identical calls differ by {worst}"
    small = torch.cat([m(x[i:i + 2].contiguous(), t[i:i + 2].contiguous(), y[i:i + 2].contiguous()) for i in range(0, N, 2)])
    assert rel(outs[0].cpu().numpy(), small.cpu().numpy()) < (3e-5 if precision == "bf16x3_presplit" else 2e-6)


@pytest.mark.parametrize("n", [8, 12])
def test_groupnorm_inside_the_conv_launch_equals_the_separate_pass(n):
    """rgm_set_gn_fuse: conv1 of a decoder ResnetBlock normalises its own output (the tiles of an image meet at a counter, bounded wait) and
    writes swish(norm2(.)) as split rows -- no separate pass over the tensor (ref taming model.py:117-126).  Same partial sums in the same
    order, same element arithmetic: the roll must be IDENTICAL to the separate pass, the library's counter must show the route ran, and
    the fallback (mode 2: every tile writes raw rows, gn_fixup_kernel converts them in place) must give the same roll again.  n = 12 latents
    (96 squares) does not fill whole XCD chunks with whole images at the 128x128 level: those launches must stay on the separate pass."""
    from gpu_util import dev
    from rgm import native as R
    from guided_diffusion.gaussian_diffusion import _decode
    R.set_gemm_precision("bf16x3_presplit")
    vae = _vae()
    z = dev(np.random.RandomState(n).randn(n, 4, 128, 16).astype(F32))
    prev = C.c_int(0)
    R.check(R.lib.rgm_set_gn_fuse(0, C.byref(prev)))
    try:
        assert prev.value == 1
        n0 = R.lib.rgm_gn_fused_launches()
        apart = _decode(z, vae, 1.0).clone()
        assert R.lib.rgm_gn_fused_launches() == n0
        R.check(R.lib.rgm_set_gn_fuse(1, None))
        R.lib.rgm_gn_fallback_tiles(1)
        fused = [_decode(z, vae, 1.0).clone() for _ in range(2)]
        launches = R.lib.rgm_gn_fused_launches() - n0
        assert R.lib.rgm_gn_fallback_tiles(1) == 0               # an idle device: no tile ever gives up its wait
        R.check(R.lib.rgm_set_gn_fuse(2, None))
        fallback = _decode(z, vae, 1.0).clone()
        assert R.lib.rgm_gn_fallback_tiles(1) > 0                # forced: every tile of every fused launch is counted
    finally:
        R.check(R.lib.rgm_set_gn_fuse(prev.value, None))
        R.set_gemm_precision("fp32")
    assert launches >= 2 * (5 if n == 8 else 2), launches       # per decode: the 16x16 level always qualifies; n = 8: every level
    assert bool(torch.isfinite(fused[0]).all())
    assert torch.equal(fused[0], fused[1])
    assert torch.equal(fused[0], apart), float((fused[0] - apart).abs().max())
    assert torch.equal(fallback, apart), float((fallback - apart).abs().max())


def test_fused_groupnorm_decode_beside_a_stream_that_holds_cus():
    """The image-level wait of the fused GroupNorm assumes an image's tiles are resident together; another stream can take CUs away
    (the library itself runs classifier chains and half batches on side streams).  64 latents are decoded while a second stream runs
    XL-28 forwards at B = 16 back to back: the rolls must equal the quiet decode bit for bit (a tile whose 1 ms wait runs out takes the
    raw-row fallback: same values), rgm_gn_fallback_tiles reports how many did, and the pair may not take longer than 1.5 x the two
    run one after the other (a stalled decode would: every fallback tile costs up to 1 ms)."""
    import time
    from gpu_util import dev
    from rgm import native as R
    from guided_diffusion.gaussian_diffusion import _decode
    R.set_gemm_precision("bf16x3_presplit")
    try:
        vae = _vae()
        m = _dit(dict(XL2, depth=28), 1)
        z = dev(np.random.RandomState(64).randn(64, 4, 128, 16).astype(F32))
        rng = np.random.RandomState(3)
        x = dev(rng.randn(16, 4, 128, 16).astype(F32))
        t = dev(rng.randint(0, 1000, size=16).astype(np.int64))
        y = dev(rng.randint(0, 3, size=16).astype(np.int64))
        quiet = _decode(z, vae, 1.0).clone()
        m(x, t, y)
        torch.cuda.synchronize()

        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        NF = 12                                                      # ~12 x 12.8 ms of foreign forwards beside a ~165 ms decode
        t_dec = min(timed(lambda: _decode(z, vae, 1.0)) for _ in range(2))
        t_for = min(timed(lambda: [m(x, t, y) for _ in range(NF)]) for _ in range(2))
        side = torch.cuda.Stream()
        R.lib.rgm_gn_fallback_tiles(1)
        outs = []

        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(NF):
                    m(x, t, y)
            outs.append(_decode(z, vae, 1.0))
            torch.cuda.current_stream().wait_stream(side)
        t_both = timed(both)
        fallbacks = R.lib.rgm_gn_fallback_tiles(1)
        print(f"decode {t_dec * 1e3:.1f} ms, {NF} forwards {t_for * 1e3:.1f} ms, together {t_both * 1e3:.1f} ms, fallback tiles {fallbacks}")
        assert fallbacks >= 0
        assert torch.equal(outs[0], quiet)
        assert t_both < 1.5 * (t_dec + t_for), (t_dec, t_for, t_both, fallbacks)
    finally:
        R.set_gemm_precision("fp32")


def test_vae_decoder_golden_through_the_big_tile_conv_kernels():
    """The VAE's 3x3 convs take the 256x256 / 512x128 one-wave-per-SIMD kernels (implicit-GEMM loader, GroupNorm sums in the epilogue at
    256- / 512-row granularity) once a launch fills the chip -- far above the golden fixtures' two squares.  rgm_set_big_tiles lowers that
    threshold to 1 tile so that the reference's decoder golden (and the decode -> uint8 path) runs through exactly those kernels."""
    from gpu_util import dev, rel
    from rgm import native as R
    from guided_diffusion.gaussian_diffusion import _decode
    g = load_golden("vae_decoder")
    R.set_gemm_precision("bf16x3_presplit")
    try:
        vae = _vae(int(g["seed"]))
        base = vae.decode(dev(g["z"]))                                   # heuristic: 128-row kernels on this small input
        R.check(R.lib.rgm_set_big_tiles(1, 1))
        with _Recorded() as rec:
            out = vae.decode(dev(g["z"]))
        assert rec.n[121] >= 10 and rec.n[122] >= 5, rec.n           # 256x256 (N = 256 / 512) and 512x128 (N = 128) conv launches
        roll = _decode(dev(g["lat"]), vae, scale_factor=1.2465)
    finally:
        R.check(R.lib.rgm_set_big_tiles(1, 256))
        R.set_gemm_precision("fp32")
    assert rel(out.cpu().numpy(), g["out"]) < 5e-5
    assert rel(out.cpu().numpy(), base.cpu().numpy()) < 2e-5
    assert bool(torch.isfinite(roll).all())

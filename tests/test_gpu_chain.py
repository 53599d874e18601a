"""-m gpu: the blocks of a DiTRotary forward as ONE persistent launch (csrc/chain.hip, rgm_set_dit_chain; ref guided_diffusion/dit.py:332-336,
618-634) against the launch-per-GEMM forward, the reference's goldens, and itself over many launches -- every work item of the chain
consumes what another workgroup of the SAME launch published, so a stale read anywhere shows up as a value that differs from run to run."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
XL = dict(hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)


@pytest.fixture(autouse=True)
def presplit():
    from rgm import native as R
    R.set_gemm_precision("bf16x3_presplit")          # the chain is a schedule of the pre-split arithmetic
    yield
    R.set_gemm_precision("fp32")


def _model(depth, seed=1):
    from gpu_util import load_module
    from guided_diffusion.dit import DiTRotary
    arch = dict(XL, depth=depth)
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3, learn_sigma=False)
    return load_module(m, synth.dit_state_dict(seed, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))


class _Chain:
    """with _Chain(min_batch): forwards of at least min_batch samples take the persistent launch; .launches counts them"""
    def __init__(self, min_batch):
        self.min_batch = min_batch

    def __enter__(self):
        from rgm import native as R
        self.prev = C.c_int(0)
        R.check(R.lib.rgm_set_dit_chain(self.min_batch, C.byref(self.prev)))
        self.n0 = R.lib.rgm_dit_chain_launches()
        return self

    def __exit__(self, *a):
        from rgm import native as R
        self.launches = R.lib.rgm_dit_chain_launches() - self.n0
        R.check(R.lib.rgm_set_dit_chain(self.prev.value, None))


def _status(m):
    from rgm import native as R
    st = C.c_int(-1)
    R.check(R.lib.rgm_dit_chain_status(m._handle, C.byref(st)))
    return st.value


def _inputs(B, seed):
    from gpu_util import dev
    rng = np.random.RandomState(seed)
    return (dev(rng.randn(B, 4, 128, 16).astype(F32)), dev(rng.randint(0, 1000, size=B).astype(np.int64)),
            dev(rng.randint(0, 3, size=B).astype(np.int64)))


@pytest.mark.parametrize("B,depth", [(16, 2), (16, 28), (5, 2), (19, 2), (32, 28), (1, 2)])
def test_chained_forward_equals_the_launch_per_gemm_forward(B, depth):
    """Same weights, same inputs: the persistent launch against the forward it replaces.  The GEMM / LayerNorm items run the one-launch
    kernels' own bodies (bit-identical sums); the attention item is the single-pass kernel where the launch-per-GEMM forward at T = 256
    runs the key-blocked one (running-maximum softmax) -- equal to ~1e-6 per block.  Identical from call to call; every item ran."""
    from gpu_util import rel
    m = _model(depth, 3)
    x, t, y = _inputs(B, 100 + B)
    with _Chain(0):
        one = m(x, t, y).clone()
    with _Chain(1) as ch:
        outs = [m(x, t, y).clone() for _ in range(3)]
    assert ch.launches == 3 and _status(m) == 0
    assert bool(torch.isfinite(outs[0]).all())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel(outs[0].cpu().numpy(), one.cpu().numpy()) < 3e-5 * (4 if depth == 28 else 1)


@pytest.mark.parametrize("tag,depth", [("xl_d2", 2), ("xl_d28", 28)])
def test_chained_forward_matches_the_reference_golden(tag, depth):
    """the reference's own outputs (tests/golden/dit_xl_*.npz: DiTRotary_XL_8 at depth 2 / 28, batch 2) through the persistent launch"""
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotary
    g = load_golden(f"dit_{tag}")
    arch = dict(XL, depth=depth)
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=depth, num_heads=16, num_classes=3,
                              learn_sigma=False), synth.dit_state_dict(int(g["seed"]), device="cuda", **arch))
    with _Chain(1) as ch:
        out = m(dev(g["x128"]), dev(g["t128"]), dev(g["y128"]))
        short = m(dev(g["x64"]), dev(g["t64"]), dev(g["y64"]))           # T = 128: not a chain shape, the launch-per-GEMM forward
    assert ch.launches == 1 and _status(m) == 0
    assert rel(out.cpu().numpy(), g["out128"]) < 2e-4
    assert rel(short.cpu().numpy(), g["out64"]) < 2e-4


def test_chained_xl28_forward_at_batch_32_against_the_reference():
    """C3's batch: the reference's own output for 32 seeded samples (round4.npz) through the persistent launch"""
    from gpu_util import dev, rel
    g = load_golden("round4")
    rb = np.random.RandomState(int(g["xl28_b32.x_seed"]))
    x = rb.randn(32, 4, 128, 16).astype(F32)
    t = rb.randint(0, 1000, size=32).astype(np.int64)
    y = rb.randint(0, 4, size=32).astype(np.int64)
    m = _model(28, 1)
    with _Chain(1) as ch:
        out = m(dev(x), dev(t), dev(y)).cpu().numpy()
    assert ch.launches == 1 and _status(m) == 0
    assert rel(out, g["xl28_b32.out"]) < 2e-4


def test_chained_forward_is_identical_over_many_launches_beside_foreign_work():
    """40 back-to-back chained XL-28 forwards at B = 16, half of them while a second stream keeps CUs busy with GEMMs (uneven load: fewer
    than 256 chain workgroups resident at a time, hand-offs at other moments): all 40 bit-identical to the first; no item ever gave up."""
    m = _model(28, 1)
    x, t, y = _inputs(16, 7)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    with _Chain(1) as ch:
        first = m(x, t, y).clone()
        for k in range(40):
            if k % 2:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(4):
                        a = (a @ a) * 1e-3
            out = m(x, t, y)
            assert torch.equal(out, first), k
        torch.cuda.current_stream().wait_stream(side)
    assert ch.launches == 41 and _status(m) == 0


def test_chain_only_takes_the_shapes_it_was_built_for():
    """fp32 / on-the-fly bf16x3 arithmetic, the classifiers (T = 257) and half windows (T = 128) keep the launch-per-GEMM forward"""
    from rgm import native as R
    m = _model(2, 3)
    x, t, y = _inputs(4, 1)
    with _Chain(1) as ch:
        m(x[:, :, :64].contiguous(), t, y)
        R.set_gemm_precision("fp32")
        m(x, t, y)
        R.set_gemm_precision("bf16x3")
        m(x, t, y)
        R.set_gemm_precision("bf16x3_presplit")
        torch.cuda.synchronize()
    assert ch.launches == 0
    with _Chain(8) as ch:
        m(x, t, y)                                     # 4 < min_batch
    assert ch.launches == 0


def test_kernels_with_the_load_wait_use_pattern_are_bit_stable():
    """tools/isa_lint.py finds the instruction pattern of the round-4 attention hazard in most kernels -- it is the plain way to
    consume a vector load.  Those that share their CUs with lock-step twins of themselves are soaked here: 12 runs per case on fixed
    inputs (tools/hazard_soak.py; the round's record holds 100), every run bit-identical to the first."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hazard_soak
    lines = []
    assert hazard_soak.soak(12, log=lines.append) == 0, lines

"""-m gpu: round-5 pin against tests/golden/round5.npz (make_golden.py round5, generated from the imported reference): ONE guided step of
BASELINE config 5 at its size -- a 4 x 512 x 16 latent through CondIndSimple(7 windows, overlap 64) of DiTRotary_XL_8 (depth 28), SCG with
n = 16 candidates (16 x 13 window forwards, 512 decoder squares), selection per segment of dc.base = 128 latent rows
(reference guided_diffusion/gaussian_diffusion.py:562-592, diff_collage/condind_long.py:24-51)."""
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_sampler import _inject

pytestmark = pytest.mark.gpu
F32 = np.float32


def test_c5_guided_step_against_the_reference(precision):
    """The reference's own run of the step: the four (16, 1) log-probability tables it hands its per-segment argmax (:587) within 2e-3 of
    each segment's spread, the SAME winner in every segment (the reference's best and second-best candidates are 2 .. 17 % of the spread
    apart: every segment is clearly separated, none is allowed to flip), the selected sample <= 2e-4."""
    import diff_collage as dc
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import dc_model_fn
    from test_gpu_fullsize import _diffusion, _vae
    from test_gpu_round4 import _xl28
    g = load_golden("round5")
    B, n, S = 1, 16, 4
    x = np.random.RandomState(int(g["c5.x_seed"])).randn(B, 4, 512, 16).astype(F32)
    nz = np.random.RandomState(int(g["c5.noise_seed"])).randn(n, B, 4, 512, 16).astype(F32)
    m, vae = _xl28(), _vae(2)

    def eps_fn(xx, tt, y=None):
        return m(xx.permute(0, 1, 3, 2).contiguous(), tt, y=y).permute(0, 1, 3, 2)
    worker = dc.CondIndSimple((4, 16, 128), eps_fn, 7, overlap_size=64)
    fn = partial(dc_model_fn, model=worker.eps_scalar_t_fn, num_classes=3, class_cond=True, cfg=False, w=0.)
    kw = {"y": torch.ones(B, dtype=torch.int64, device="cuda"),
          "rule": {"pitch_hist": dev(g["c5.target.pitch_hist"]), "note_density": dev(g["c5.target.note_density"])}}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance", dc=SimpleNamespace(base=128))
    d = _diffusion("")
    d.t_end = 0
    _inject(d, nz)
    out = d.p_sample(fn, dev(x), dev(g["c5.t"]), clip_denoised=False, model_kwargs=kw, embed_model=vae, scale_factor=1.2465,
                     guidance_kwargs=guid, scg_kwargs={"num_samples": n, "pitch_hist": 40., "note_density": 1.})
    table, ref = d.last_scg["total_log_prob"].cpu().numpy(), g["c5.total_log_prob"]
    assert table.shape == ref.shape == (n, S, B)
    spread = ref.max(0) - ref.min(0)                                   # (S, B)
    srt = np.sort(ref, axis=0)
    assert ((srt[-1] - srt[-2]) > 0.015 * spread).all()                # the fixture's segments are all clearly separated
    err = np.abs(table - ref).max(0)
    assert (err < 2e-3 * spread + 1e-4 * np.abs(ref).max()).all(), (err, spread)
    assert np.array_equal(d.last_scg["max_ind"].cpu().numpy(), g["c5.max_ind"])
    assert rel(out["sample"].cpu().numpy(), g["c5.sample"]) < (5e-5 if precision == "fp32" else 2e-4)
    assert rel(out["pred_xstart"].cpu().numpy(), g["c5.pred_xstart"]) < (5e-5 if precision == "fp32" else 3e-4)


def test_kernels_with_the_load_wait_use_pattern_are_bit_stable():
    """tools/isa_lint.py finds the instruction pattern of the round-4 attention hazard in most kernels -- it is the plain way to
    consume a vector load.  Those that share their CUs with lock-step twins of themselves are soaked here: 12 runs per case on fixed
    inputs (tools/hazard_soak.py; the round's record holds 100), every run bit-identical to the first."""
    import os, sys
    from rgm import native as R
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hazard_soak
    prev = R.lib.rgm_get_gemm_precision()
    lines = []
    try:
        assert hazard_soak.soak(12, log=lines.append) == 0, lines
    finally:
        R.lib.rgm_set_gemm_precision(prev)

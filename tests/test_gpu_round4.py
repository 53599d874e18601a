"""-m gpu: round-4 pins against tests/golden/round4.npz (make_golden.py round4, generated from the imported reference): values the
REFERENCE computes at the sizes the configs run -- the XL-28 forward at B = 32, one guided step of cond_table/all/scg_classifier_all.yml
(classifier guidance AND SCG, B = 4, n = 16, 512 decoder squares), condind_long's 13-window collage eps at XL-28 -- and
ModelMeanType.PREVIOUS_X together with learned-range variances; and against round4b.npz: one classifier-guided step of config[2] at B = 32
and one DDIM step of config[1] at B = 16."""
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth
from test_gpu_sampler import SM, _inject, _model_fn

pytestmark = pytest.mark.gpu
F32 = np.float32
XL28 = dict(depth=28, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)


def _xl28(num_classes=3):
    from gpu_util import load_module
    from guided_diffusion.dit import DiTRotary
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=28, num_heads=16, num_classes=num_classes,
                  learn_sigma=False)
    return load_module(m, synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **dict(XL28, num_classes=num_classes)))


@pytest.mark.parametrize("tag,clip,use_dfn", [("prevx_lr_clip", True, False), ("prevx_lr_dfn", False, True)])
def test_previous_x_with_learned_range_variances(tag, clip, use_dfn, precision):
    """Reference :299-313 + :331-338: with ModelMeanType.PREVIOUS_X the posterior mean is the network's raw output -- whatever
    clip_denoised / denoised_fn do to pred_xstart -- also when the variance is the learned interpolation (round 3 restored the mean
    only behind the fixed-variance kernel: ADVICE r3).  Golden: the reference's p_sample on a learn_sigma=True network."""
    from gpu_util import dev, load_module, rel
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.dit import DiTRotary
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    g = load_golden("round4")
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=3,
                              learn_sigma=True), synth.dit_state_dict(int(g["prevx_lr.seed"]), **dict(SM, out_ch=8)))
    d = SpacedDiffusion(use_timesteps=space_timesteps(1000, [1000]), betas=gd.get_named_beta_schedule("linear", 1000),
                        model_mean_type=gd.ModelMeanType.PREVIOUS_X, model_var_type=gd.ModelVarType.LEARNED_RANGE,
                        loss_type=gd.LossType.MSE, rescale_timesteps=False)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])

    def dfn(v):
        return v.clamp(-0.5, 0.5) * 0.9
    out = d.p_sample(_model_fn(m), dev(g["prevx_lr.x"]), dev(g[f"{tag}.t"]), clip_denoised=clip, denoised_fn=dfn if use_dfn else None,
                     model_kwargs={"y": dev(g["prevx_lr.y"])})
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4
    # pred_xstart = process_xstart(x_prev / coef1 - ...): 1 / coef1 amplifies the network's arithmetic noise (see test_gpu_round3);
    # the processed estimate is bounded, so compare on the scale of the bound
    a, b = out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]
    inv_c1 = 1.0 / float(d.posterior_mean_coef1[int(g[f"{tag}.t"][0])])
    net = 3e-5 if precision == "fp32" else 2e-4
    assert np.abs(a - b).max() < max(inv_c1 * net, 2e-4)
    assert float(np.abs(a).max()) <= (0.45 if use_dfn else 1.0) + 1e-6


def test_xl28_forward_at_batch_32_against_the_reference():
    """C3's batch through DiTRotary_XL_8 (depth 28): the reference's own output for 32 seeded samples, in the arithmetic the bench runs and
    in exact fp32 -- direct, not chained through batches of 2; the launch records prove the big-tile kernels carried it."""
    from gpu_util import dev, rel
    from rgm import native as R
    from test_gpu_fullsize import _Recorded
    g = load_golden("round4")
    rb = np.random.RandomState(int(g["xl28_b32.x_seed"]))
    x = rb.randn(32, 4, 128, 16).astype(F32)
    t = rb.randint(0, 1000, size=32).astype(np.int64)
    y = rb.randint(0, 4, size=32).astype(np.int64)
    m = _xl28()
    try:
        for prec, tol in (("bf16x3_presplit", 2e-4), ("fp32", 5e-5)):
            R.set_gemm_precision(prec)
            with _Recorded() as rec:
                out = m(dev(x), dev(t), dev(y)).cpu().numpy()
            if prec == "bf16x3_presplit":
                assert rec.big >= 28, rec.n
            e = rel(out, g["xl28_b32.out"])
            assert e < tol, (prec, e)
    finally:
        R.set_gemm_precision("fp32")


def test_c5_collage_eps_at_xl28_against_the_reference(precision):
    """condind_long over a 4 x 16 x 512 latent at XL depth 28: 7 full windows + 6 overlap halves (T = 128 -- the short-sequence attention
    launch) merged by the reference's own CondIndSimple."""
    import diff_collage as dc
    from gpu_util import dev, rel
    g = load_golden("round4")
    w = np.random.RandomState(int(g["c5.w_seed"])).randn(1, 4, 16, 512).astype(F32)
    m = _xl28()

    def eps_fn(x, t, y=None):
        return m(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
    worker = dc.CondIndSimple((4, 16, 128), eps_fn, 7, overlap_size=64)
    out = worker.eps_scalar_t_fn(dev(w), dev(g["c5.t"]), y=dev(g["c5.y"])).cpu().numpy()
    assert rel(out, g["c5.eps"]) < (5e-5 if precision == "fp32" else 3e-4)


def test_c4_step_with_classifiers_and_scg_against_the_reference(precision):
    """ONE guided step of scg_classifier_all.yml (minus the chord rule) at its real size, against the reference's own run of it:
    B = 4, n = 16, XL-28 on 4 + 64 rows, the pitch / note-density classifiers' gradients (scales 400 / 10) shifting the mean, 512 decoder
    squares, two rules.  Same winners; the (16, 4) log-probability table the reference handed its argmax (:540) within 2e-3 of its
    spread; the selected sample <= 2e-4."""
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import model_fn
    from test_gpu_fullsize import _c4_classifiers, _diffusion, _vae
    g = load_golden("round4")
    B, n = 4, 16
    x = np.random.RandomState(int(g["c4.x_seed"])).randn(B, 4, 128, 16).astype(F32)
    nz = np.random.RandomState(int(g["c4.noise_seed"])).randn(n, B, 4, 128, 16).astype(F32)
    m, vae = _xl28(), _vae(2)
    fn = partial(model_fn, model=m, num_classes=3, class_cond=True, cfg=False, w=0.)
    kw = {"y": torch.ones(B, dtype=torch.int64, device="cuda"),
          "rule": {"pitch_hist": dev(g["c4.target.pitch_hist"]), "note_density": dev(g["c4.target.note_density"])}}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="classifier_guidance")
    d = _diffusion("")
    d.t_end = 0
    _inject(d, nz)
    out = d.p_sample(fn, dev(x), dev(g["c4.t"]), clip_denoised=False, cond_fn=_c4_classifiers(), model_kwargs=kw, embed_model=vae,
                     scale_factor=1.2465, guidance_kwargs=guid, scg_kwargs={"num_samples": n, "pitch_hist": 40., "note_density": 1.})
    table, ref = d.last_scg["total_log_prob"].cpu().numpy(), g["c4.total_log_prob"]
    assert table.shape == ref.shape == (n, B)
    spread = float((ref.max(0) - ref.min(0)).min())
    assert np.abs(table - ref).max() < 2e-3 * spread + 1e-4 * np.abs(ref).max(), (np.abs(table - ref).max(), spread)
    assert np.array_equal(d.last_scg["max_ind"].cpu().numpy().reshape(-1), g["c4.max_ind"])
    assert rel(out["sample"].cpu().numpy(), g["c4.sample"]) < (5e-5 if precision == "fp32" else 2e-4)
    assert rel(out["pred_xstart"].cpu().numpy(), g["c4.pred_xstart"]) < (5e-5 if precision == "fp32" else 3e-4)


def test_c3_step_at_batch_32_against_the_reference(precision):
    """ONE classifier-guided DDPM step of BASELINE config[2] at its batch, against the reference's own run of it (round4b.npz): B = 32 on
    the '250' chain with a different timestep per row, unconditional XL-28, the note-density DiTRotary-S/8-cls (depth 12) through
    grad_nn_zt_mse x 10.  The error must stay far below what the guidance itself moved (stored per row by the generator)."""
    from gpu_util import dev, load_module, rel
    from guided_diffusion.condition_functions import composite_nn_zt, model_fn
    from guided_diffusion.dit import DiTRotaryClassifier
    from test_gpu_fullsize import _diffusion
    g = load_golden("round4b")
    B = 32
    x = np.random.RandomState(int(g["c3.x_seed"])).randn(B, 4, 128, 16).astype(F32)
    nz = np.random.RandomState(int(g["c3.noise_seed"])).randn(B, 4, 128, 16).astype(F32)
    arch = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    clf = load_module(DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=12, num_heads=6,
                                          num_classes=16), synth.dit_state_dict(3, **arch))
    cond = partial(composite_nn_zt, fns=["grad_nn_zt_mse"], classifier_scales=[10.], classifiers=[clf], rule_names=["note_density"])
    fn = partial(model_fn, model=_xl28(0), num_classes=0, class_cond=False, cfg=False, w=0.)
    d = _diffusion("250")
    d.t_end = 0
    _inject(d, nz)
    out = d.p_sample(fn, dev(x), dev(g["c3.t"]), clip_denoised=False, cond_fn=cond, model_kwargs={"rule": {"note_density": dev(g["c3.target"])}},
                     guidance_kwargs=SimpleNamespace(schedule=False, method="classifier_guidance"))
    a, b = out["sample"].cpu().numpy(), g["c3.sample"]
    assert rel(a, b) < (5e-5 if precision == "fp32" else 2e-4)
    assert rel(out["pred_xstart"].cpu().numpy(), g["c3.pred_xstart"]) < (5e-5 if precision == "fp32" else 3e-4)
    # per row: the deviation is a small fraction of the shift the classifier gradient produced in the reference
    dev_row = np.abs(a - b).reshape(B, -1).max(1)
    assert (dev_row < 0.05 * g["c3.guidance_shift"] + 2e-5).all(), (dev_row / g["c3.guidance_shift"]).max()


def test_c2_ddim_step_at_batch_16_against_the_reference(precision):
    """ONE DDIM step (eta = 1, 'ddim50' chain) of BASELINE config[1] at its batch of 16, unconditional XL-28, against the reference's
    ddim_sample on the same seeded x_t and noise -- the step bench.py times, in the arithmetic it times it in."""
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import model_fn
    from test_gpu_fullsize import _diffusion
    g = load_golden("round4b")
    B = 16
    x = np.random.RandomState(int(g["c2.x_seed"])).randn(B, 4, 128, 16).astype(F32)
    nz = np.random.RandomState(int(g["c2.noise_seed"])).randn(B, 4, 128, 16).astype(F32)
    fn = partial(model_fn, model=_xl28(0), num_classes=0, class_cond=False, cfg=False, w=0.)
    d = _diffusion("ddim50")
    d.t_end = 0
    _inject(d, nz)
    out = d.ddim_sample(fn, dev(x), dev(g["c2.t"]), clip_denoised=False, model_kwargs={}, eta=1.0)
    assert rel(out["sample"].cpu().numpy(), g["c2.sample"]) < (5e-5 if precision == "fp32" else 2e-4)
    assert rel(out["pred_xstart"].cpu().numpy(), g["c2.pred_xstart"]) < (5e-5 if precision == "fp32" else 3e-4)


def test_side_stream_classifiers_and_gradient_give_the_same_bits(precision):
    """Round 4 runs independent chains of small launches on side streams: the classifiers of a composite cond_fn (composite_nn_zt) and the
    guidance gradient beside the eps-network forward (p_sample, _search_step_inputs).  Forked from and joined to the caller's stream, summed
    in the reference's order: a classifier-guided step and a classifier-guided SCG step must come out bit-identical with the streams on
    and off."""
    from gpu_util import dev
    from guided_diffusion import condition_functions as cf, gaussian_diffusion as gd
    from guided_diffusion.condition_functions import model_fn
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    from test_gpu_fullsize import XL2, _c4_classifiers, _diffusion, _dit, _vae
    B, n = 4, 4
    g = load_golden("round4")
    x = dev(np.random.RandomState(11).randn(B, 4, 128, 16).astype(F32))
    m, vae = _dit(XL2, 1), _vae(2)
    fn = partial(model_fn, model=m, num_classes=3, class_cond=True, cfg=False, w=0.)
    kw = {"y": torch.ones(B, dtype=torch.int64, device="cuda"),
          "rule": {"pitch_hist": dev(g["c4.target.pitch_hist"]), "note_density": dev(g["c4.target.note_density"])}}
    cond = _c4_classifiers()
    outs = {}
    keep = (cf.CONCURRENT_CLASSIFIERS, gd._CONCURRENT_GRAD)
    try:
        for on in (False, True):
            cf.CONCURRENT_CLASSIFIERS = on
            gd._CONCURRENT_GRAD = on
            res = []
            for scg in (None, {"num_samples": n, "pitch_hist": 40., "note_density": 1.}):
                d = _diffusion("")
                d.t_end = 0
                d.noise = PhiloxNoise(seed=5)
                t = torch.full((B,), 400, dtype=torch.int64, device="cuda")
                guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="classifier_guidance")
                out = d.p_sample(fn, x, t, clip_denoised=False, cond_fn=cond, model_kwargs=kw, embed_model=vae if scg else None,
                                 scale_factor=1.2465, guidance_kwargs=guid, scg_kwargs=scg)
                res.append(out["sample"].clone())
            torch.cuda.synchronize()
            outs[on] = res
    finally:
        cf.CONCURRENT_CLASSIFIERS, gd._CONCURRENT_GRAD = keep
    for a, b in zip(outs[False], outs[True]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)

"""-m gpu: the editing path (scripts/edit.py) -- VAE encoder, _encode, replacement-conditioned steps, the loop start --
against goldens produced by the reference (tests/golden/make_golden.py edit)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth
from test_gpu_sampler import SM, _diffusion, _dit, _inject, _model_fn

pytestmark = pytest.mark.gpu
F32 = np.float32


def _vae_full(seed=2):
    from gpu_util import load_module
    from taming.models.klvae_pedal import AutoencoderKL
    return load_module(AutoencoderKL(), synth.vae_state_dict(seed, encoder=True))


def _edit_kwargs(g, full=False):
    from gpu_util import dev
    mask = np.zeros_like(g["mask"]) if full else g["mask"]
    return {"gt": dev(g["gt"]), "mask": dev(mask), "l_start": 0 if full else int(g["l_start"]),
            "l_end": 128 if full else int(g["l_end"]), "noise_level": 3}


def test_vae_encoder_moments_and_encode_latent(precision):
    from gpu_util import dev, rel
    from guided_diffusion.gaussian_diffusion import _encode
    g = load_golden("edit")
    vae = _vae_full(int(g["seed"]))
    mom = vae.encode_save(dev(g["tiles"]))
    assert mom.shape == (2, 8, 16, 16)
    tol = 2e-5 if precision == "fp32" else 2e-4
    assert rel(mom.cpu().numpy(), g["moments"]) < tol
    lat = _encode(dev(g["roll"]), vae, scale_factor=1.2465)
    assert lat.shape == (1, 4, 32, 16)
    assert rel(lat.cpu().numpy(), g["latent"]) < tol
    post = vae.encode(dev(g["tiles"]))
    assert torch.equal(post.mode(), mom[:, :4]) and post.sample().shape == (2, 4, 16, 16)
    # encode -> decode round trip stays finite and in range on a synthetic-weight model (no trained weights here)
    assert torch.isfinite(vae.decode(mom[:, :4].contiguous())).all()


def test_encoder_needs_its_parameters():
    """A decoder-only state dict still decodes; encode() fails loudly until the encoder weights are set."""
    from gpu_util import dev
    from rgm import native as R
    from taming.models.klvae_pedal import AutoencoderKL
    vae = AutoencoderKL().to("cuda").eval()
    vae._ensure_native()
    assert R.lib.rgm_vae_missing_params(vae._handle) == 0 and R.lib.rgm_vae_encoder_missing_params(vae._handle) == 0
    import ctypes as C
    h = C.c_void_p()
    R.check(R.lib.rgm_vae_create(C.byref(h)))
    assert R.lib.rgm_vae_encoder_missing_params(h) > 0
    x = torch.zeros(1, 3, 128, 128, device="cuda")
    out = torch.empty(1, 8, 16, 16, device="cuda")
    need = R.lib.rgm_vae_workspace_bytes(h, 1)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc = R.lib.rgm_vae_encode(h, R.ptr(x), R.ptr(out), 1, R.ptr(ws), need, R.current_stream())
    assert rc != 0 and b"encoder parameters not set" in R.lib.rgm_last_error()
    R.lib.rgm_vae_destroy(h)


@pytest.mark.parametrize("tag,rs,ddim,clip", [("ddpm", "", False, True), ("ddim", "ddim50", True, False)])
def test_replacement_conditioned_step_matches_reference(tag, rs, ddim, clip, precision):
    from gpu_util import dev, rel
    g = load_golden("edit")
    m = _dit(SM, 11)
    d = _diffusion(rs)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])
    kw = dict(clip_denoised=clip, model_kwargs={"y": dev(g["y"])}, edit_kwargs=_edit_kwargs(g))
    x, t = dev(g["x"]), dev(g[f"{tag}.t"])
    out = d.ddim_sample(_model_fn(m), x, t, eta=1.0, **kw) if ddim else d.p_sample(_model_fn(m), x, t, **kw)
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4
    # x0 = c1*x - c2*eps' cancels ~6x the eps error at these timesteps (the numpy oracle itself: 3.7e-4 on this golden)
    assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 2e-3
    # where the mask is 1 the x0 estimate IS the ground truth (clipped like every x0 when clip_denoised)
    gt = np.clip(g["gt"], -1, 1) if clip else g["gt"]
    keep = g["mask"] == 1
    assert np.abs(out["pred_xstart"].cpu().numpy()[keep] - gt[keep]).max() < 2e-4


def test_classifier_guided_edit_step_matches_reference(precision):
    from functools import partial
    from types import SimpleNamespace
    from gpu_util import dev, load_module, rel
    from guided_diffusion.condition_functions import composite_nn_zt
    from guided_diffusion.dit import DiTRotaryClassifier
    g = load_golden("edit")
    m = _dit(SM, 11)
    cls_arch = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    cm = load_module(DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6,
                                         num_classes=16), synth.dit_state_dict(4, **cls_arch))
    d = _diffusion("250")
    d.t_end = 0
    _inject(d, g["cg.noise"])
    cond = partial(composite_nn_zt, fns=["grad_nn_zt_mse"], classifier_scales=[10.], classifiers=[cm], rule_names=["note_density"])
    out = d.p_sample(_model_fn(m), dev(g["x"]), dev(g["cg.t"]), clip_denoised=False, cond_fn=cond,
                     model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["cg.rule"])}},
                     guidance_kwargs=SimpleNamespace(schedule=False, method="classifier_guidance"),
                     edit_kwargs=_edit_kwargs(g, full=True))
    assert rel(out["sample"].cpu().numpy(), g["cg.sample"]) < 5e-4


def test_scg_edit_step_scores_only_the_editable_rows(precision):
    from types import SimpleNamespace
    from gpu_util import dev, rel
    g = load_golden("edit")
    m, vae = _dit(SM, 11), _vae_full(2)
    d = _diffusion("")
    d.t_end = 0
    _inject(d, g["scg.noise"])
    tgt = {"pitch_hist": dev(g["scg.target.pitch_hist"]), "note_density": dev(g["scg.target.note_density"])}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance")
    out = d.p_sample(_model_fn(m), dev(g["x"]), dev(g["scg.t"]), clip_denoised=False,
                     model_kwargs={"y": dev(g["y"]), "rule": tgt}, embed_model=vae, scale_factor=1.2465,
                     guidance_kwargs=guid, scg_kwargs={"num_samples": 3, "pitch_hist": 40., "note_density": 1.},
                     edit_kwargs=_edit_kwargs(g))
    assert np.array_equal(d.last_scg["max_ind"].cpu().numpy(), g["scg.max_ind"])
    assert rel(d.last_scg["total_log_prob"].cpu().numpy(), g["scg.total_log_prob"]) < 1e-4
    assert rel(out["sample"].cpu().numpy(), g["scg.sample"]) < 2e-4


def test_edit_loop_starts_from_the_noised_ground_truth(precision):
    from gpu_util import dev, rel
    g = load_golden("edit")
    m = _dit(SM, 11)
    d = _diffusion("")
    _inject(d, g["loop.init_noise"], *g["loop.noise"])
    out = d.p_sample_loop(_model_fn(m), (2, 4, 128, 16), clip_denoised=False, model_kwargs={"y": dev(g["y"])}, device="cuda",
                          edit_kwargs=_edit_kwargs(g))
    assert rel(out.cpu().numpy(), g["loop.sample"]) < 2e-4


@pytest.mark.parametrize("tag,rs", [("dps250", "250"), ("dps", "")])
def test_dps_guided_step_matches_reference(tag, rs, precision):
    """DPS guidance (reference condition_mean :415-465 with nn_z0_mse_dummy): the reference differentiates
    x0_hat(x_t) -> classifier with autograd; here the classifier's fused gradient feeds the eps-network's VJP."""
    from functools import partial
    from types import SimpleNamespace
    from gpu_util import dev, load_module, rel
    from guided_diffusion.condition_functions import composite_nn_zt
    from guided_diffusion.dit import DiTRotaryClassifier
    g = load_golden("dps")
    m = _dit(SM, 11)
    cls_arch = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    cm = load_module(DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6,
                                         num_classes=16), synth.dit_state_dict(4, **cls_arch))
    d = _diffusion(rs)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])
    cond = partial(composite_nn_zt, fns=["nn_z0_mse_dummy"], classifier_scales=[1.], classifiers=[cm], rule_names=["note_density"])
    gk = SimpleNamespace(schedule=False, method="dps", step_size=1.5, nn=True, vae=False)
    out = d.p_sample(_model_fn(m), dev(g["x"]), dev(g[f"{tag}.t"]), clip_denoised=False, cond_fn=cond,
                     model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["rule"])}}, guidance_kwargs=gk)
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 5e-4
    assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 5e-4

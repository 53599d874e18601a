"""-m gpu: DPS through rule(decode(x0)) (SURVEY 8f.1, configs cond_table/single/dps_rule) -- the VAE decoder's input gradient
(rgm_vae_decode_latent_save / _vjp), the pitch histogram's value-and-gradient kernel and full guided steps, against goldens
the reference's autograd produced (tests/golden/make_golden.py dpsrule) plus size-independent adjoint properties."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth
from test_gpu_sampler import SM, _diffusion, _dit, _inject, _model_fn

pytestmark = pytest.mark.gpu
F32 = np.float32


def _vae(seed=2):
    from gpu_util import load_module
    from taming.models.klvae_pedal import AutoencoderKL
    return load_module(AutoencoderKL(), synth.vae_state_dict(seed, encoder=True))


def test_decoder_vjp_matches_reference_autograd(precision):
    from gpu_util import dev, rel
    g = load_golden("dps_rule")
    vae = _vae()
    lat = dev(g["vjp.lat"])
    roll = vae.decode_latent_save(lat, scale_factor=1.2465)
    plain = vae.decode_latent(lat, scale_factor=1.2465)
    assert torch.equal(roll, plain)                                    # the saving forward is the same kernel sequence
    assert rel(roll.sum(dim=(2, 3)).cpu().numpy(), g["vjp.roll_sum"]) < 1e-3
    cot = np.random.RandomState(int(g["vjp.gseed"])).randn(2, 3, 128, 256).astype(F32)
    dl = vae.decode_latent_vjp(dev(cot))
    assert dl.shape == lat.shape
    tol = 3e-5 if precision == "fp32" else 3e-4
    assert rel(dl.cpu().numpy(), g["vjp.dlat"]) < tol
    # linear in the cotangent.  A power-of-two factor commutes with every rounding; what is left is the summation order of the
    # GroupNorm reductions' fp64 LDS atomics (last-bit differences, which a bf16 split can amplify to 2^-17 of an element)
    dl2 = vae.decode_latent_vjp(dev(cot * -2.0))
    assert rel(dl2.cpu().numpy(), -2.0 * dl.cpu().numpy()) < (1e-6 if precision == "fp32" else 1e-4)


@pytest.mark.parametrize("N,H", [(1, 16), (2, 128)])
def test_decoder_vjp_is_the_adjoint_of_the_decode(N, H):
    """<J v, u> == <v, J^T u> with J v from central differences of the forward in fp32 arithmetic -- no reference needed, any size
    (H = 128 is the sampling shape: 8 squares per sample)."""
    from rgm import native as R
    R.set_gemm_precision("fp32")
    try:
        vae = _vae(5)
        gen = torch.Generator(device="cuda").manual_seed(7)
        lat = torch.randn(N, 4, H, 16, device="cuda", generator=gen)
        v = torch.randn(N, 4, H, 16, device="cuda", generator=gen)
        u = torch.randn(N, 3, 128, 8 * H, device="cuda", generator=gen)
        h = 2e-2
        jv = (vae.decode_latent(lat + h * v, scale_factor=1.3).double() - vae.decode_latent(lat - h * v, scale_factor=1.3).double()) / (2 * h)
        vae.decode_latent_save(lat, scale_factor=1.3)
        jtu = vae.decode_latent_vjp(u)
        lhs = float((jv * u.double()).sum())
        rhs = float((v.double() * jtu.double()).sum())
        scale = float(jv.norm() * u.double().norm())
        assert abs(lhs - rhs) / scale < 2e-3, (lhs, rhs, scale)
    finally:
        R.set_gemm_precision("fp32")


def test_decoder_grad_state_errors():
    """The grad entry points fail loudly without rgm_vae_enable_grad, with a short workspace, and set_param invalidates."""
    import ctypes as C
    from rgm import native as R
    vae = _vae()
    vae._ensure_native()
    h = vae._handle
    lat = torch.zeros(1, 4, 16, 16, device="cuda")
    roll = torch.empty(1, 3, 128, 128, device="cuda")
    need = R.lib.rgm_vae_grad_workspace_bytes(h, 1)
    assert need > R.lib.rgm_vae_workspace_bytes(h, 1)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = R.lib.rgm_vae_decode_latent_save(h, R.ptr(lat), 1.0, R.ptr(roll), 1, 16, R.ptr(ws), need, R.current_stream())
    assert st != 0 and b"rgm_vae_enable_grad" in R.lib.rgm_last_error()
    R.check(R.lib.rgm_vae_enable_grad(h))
    st = R.lib.rgm_vae_decode_latent_save(h, R.ptr(lat), 1.0, R.ptr(roll), 1, 16, R.ptr(ws), need - 1, R.current_stream())
    assert st != 0 and b"workspace" in R.lib.rgm_last_error()
    R.check(R.lib.rgm_vae_decode_latent_save(h, R.ptr(lat), 1.0, R.ptr(roll), 1, 16, R.ptr(ws), need, R.current_stream()))
    w = torch.zeros(4, device="cuda")
    shape = (C.c_int64 * 1)(4)
    R.check(R.lib.rgm_vae_set_param(h, b"post_quant_conv.bias", R.ptr(w), shape, 1))
    st = R.lib.rgm_vae_decode_latent_vjp(h, R.ptr(roll), 1.0, R.ptr(lat), 1, 16, R.ptr(ws), need, R.current_stream())
    assert st != 0 and b"rgm_vae_enable_grad" in R.lib.rgm_last_error()


def test_pitch_hist_value_and_grad_matches_reference_autograd():
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import _rule_x0_vag, rule_x0_mse_dummy
    g = load_golden("dps_rule")
    r = (np.random.RandomState(int(g["ph.rseed"])).rand(2, 3, 128, 256).astype(F32) * 2 - 1) * 0.8
    roll = dev(r)
    lp, d = _rule_x0_vag(roll, dev(g["ph.target"]), "pitch_hist", 1.0)
    assert rel(lp.cpu().numpy(), g["ph.logp"]) < 1e-5
    d = d.cpu().numpy()
    assert np.abs(d - d[:, :, :, :1]).max() == 0                       # constant along time
    assert rel(d[:, :, :, 0], g["ph.grad_rows"]) < 2e-5
    assert (d[:, 1:] == 0).all() and (d[:, 0, :21] == 0).all() and (d[:, 0, 109:] == 0).all()
    # the value-only cond_fn agrees, and a scale multiplies both
    v = rule_x0_mse_dummy(dev(r), None, rule=dev(g["ph.target"]), rule_name="pitch_hist")
    assert rel(v.cpu().numpy(), g["ph.logp"]) < 1e-5
    lp3, d3 = _rule_x0_vag(dev(r), dev(g["ph.target"]), "pitch_hist", 3.0)
    assert rel(lp3.cpu().numpy(), 3 * g["ph.logp"]) < 1e-5 and rel(d3.cpu().numpy(), 3 * d) < 1e-5


def test_pitch_hist_value_and_grad_matches_oracle_at_sampling_size():
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import _rule_x0_vag
    from oracle import rules_np as orl
    rng = np.random.RandomState(77)
    r = np.tanh(rng.randn(5, 3, 128, 1024)).astype(F32)
    tgt = rng.rand(5, 12).astype(F32)
    ologp, ograd = orl.pitch_hist_logp_grad(r.copy(), tgt, scale=40.0)
    roll = dev(r)
    lp, d = _rule_x0_vag(roll, dev(tgt), "pitch_hist", 40.0)
    assert rel(lp.cpu().numpy(), ologp) < 1e-5 and rel(d.cpu().numpy(), ograd) < 2e-5
    assert (roll[:, 0, :21] == -1).all() and (roll[:, 0, 109:] == -1).all()      # piano_like wrote through, like the reference


def _dps_rule_step(d, m, vae, g, tag, rule_names=("pitch_hist",), rule=None, step_size=100.0):
    from functools import partial
    from types import SimpleNamespace
    from gpu_util import dev
    from guided_diffusion.condition_functions import composite_rule
    cond = partial(composite_rule, fns=["rule_x0_mse_dummy"] * len(rule_names), classifier_scales=[1.] * len(rule_names),
                   rule_names=list(rule_names))
    gk = SimpleNamespace(schedule=False, method="dps", step_size=step_size, nn=False, vae=True)
    return d.p_sample(_model_fn(m), dev(g["x"]), dev(g[f"{tag}.t"]), clip_denoised=False, cond_fn=cond,
                      model_kwargs={"y": dev(g["y"]), "rule": rule if rule is not None else {"pitch_hist": dev(g["rule"])}},
                      guidance_kwargs=gk, embed_model=vae, scale_factor=1.2465)


@pytest.mark.parametrize("tag,rs", [("dpsr250", "250"), ("dpsr", "")])
def test_dps_rule_guided_step_matches_reference(tag, rs, precision):
    """condition_mean's dps branch with guidance.nn False (reference :415-465): x0_hat -> _decode -> pitch_hist -> -MSE, gradient back
    through the decoder and the eps-network.  `shift` = guided - unguided sample isolates the guidance term."""
    from gpu_util import dev, rel
    g = load_golden("dps_rule")
    m, vae = _dit(SM, 11), _vae()
    d = _diffusion(rs)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])
    out = _dps_rule_step(d, m, vae, g, tag)
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 5e-5
    assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 5e-5
    d0 = _diffusion(rs)
    d0.t_end = 0
    _inject(d0, g[f"{tag}.noise"])
    plain = d0.p_sample(_model_fn(m), dev(g["x"]), dev(g[f"{tag}.t"]), clip_denoised=False, model_kwargs={"y": dev(g["y"])})
    shift = (out["sample"] - plain["sample"]).cpu().numpy()
    assert rel(shift, g[f"{tag}.shift"]) < (2e-3 if precision == "fp32" else 5e-3)


def test_dps_rule_with_a_hard_threshold_rule_is_unguided():
    """note_density counts behind hard thresholds: zero gradient (in the reference's autograd too), so the step equals the plain one."""
    from gpu_util import dev
    g = load_golden("dps_rule")
    m, vae = _dit(SM, 11), _vae()
    outs = []
    for guided in (True, False):
        d = _diffusion("250")
        d.t_end = 0
        _inject(d, g["dpsr250.noise"])
        if guided:
            rule = {"note_density": torch.full((2, 16), 2.0, device="cuda")}
            outs.append(_dps_rule_step(d, m, vae, g, "dpsr250", rule_names=("note_density",), rule=rule)["sample"])
        else:
            outs.append(d.p_sample(_model_fn(m), dev(g["x"]), dev(g["dpsr250.t"]), clip_denoised=False,
                                   model_kwargs={"y": dev(g["y"])})["sample"])
    assert torch.equal(outs[0], outs[1])


def test_dps_rule_with_a_user_torch_rule():
    """A rule a user registers in FUNC_DICT as plain torch code is differentiated by autograd on the decoded roll only; the decoder and
    the eps-network still go through the native VJPs.  mean-velocity rule: d/d roll is a constant, so the guided shift must be
    the decoder/eps pull-back of that constant -- compared with the pitch-hist path's machinery through linearity."""
    from gpu_util import dev
    from music_rule_guidance import rule_maps
    g = load_golden("dps_rule")
    m, vae = _dit(SM, 11), _vae()

    def mean_velocity(roll):
        return roll[:, 0].mean(dim=(1, 2)).reshape(-1, 1)
    rule_maps.FUNC_DICT["mean_velocity"] = mean_velocity
    try:
        res = []
        for target in (100.0, -100.0):
            d = _diffusion("250")
            d.t_end = 0
            _inject(d, g["dpsr250.noise"])
            rule = {"mean_velocity": torch.full((2, 1), target, device="cuda")}
            res.append(_dps_rule_step(d, m, vae, g, "dpsr250", rule_names=("mean_velocity",), rule=rule, step_size=50.0)["sample"])
        d0 = _diffusion("250")
        d0.t_end = 0
        _inject(d0, g["dpsr250.noise"])
        plain = d0.p_sample(_model_fn(m), dev(g["x"]), dev(g["dpsr250.t"]), clip_denoised=False, model_kwargs={"y": dev(g["y"])})["sample"]
    finally:
        del rule_maps.FUNC_DICT["mean_velocity"]
    up, down = res[0] - plain, res[1] - plain
    assert float(up.abs().max()) > 1e-4
    # grad / sqrt(-logp) = -2 e J / |e| = -+ 2 J: targets on either side of the current mean push in opposite directions, equally hard
    assert float((up + down).abs().max()) < 2e-3 * float(up.abs().max())

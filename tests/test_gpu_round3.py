"""-m gpu: round-3 pins against tests/golden/round3.npz (make_golden.py round3, generated from the imported reference):
denoised_fn together with edit_kwargs (process_xstart runs twice in the reference), ModelMeanType.PREVIOUS_X, what the reference does
with SCG / DPS on a learn_sigma=True network (it raises), and SCG with a per-element (learned) noise scale pinned to the reference's own
scg_sample fed the eps half of the network's output."""
import json
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth
from test_gpu_sampler import SM, _dit, _inject, _model_fn, _vae

pytestmark = pytest.mark.gpu
F32 = np.float32


def _diff(rs, **kw):
    from guided_diffusion.script_util import create_diffusion
    args = dict(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing=rs, use_kl=False,
                predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)
    args.update(kw)
    d = create_diffusion(**args)
    d.t_end = 0
    return d


@pytest.mark.parametrize("tag,rs,ddim", [("dfn_edit_ddpm", "", False), ("dfn_edit_ddim", "ddim50", True)])
def test_denoised_fn_is_applied_again_after_the_edit_replacement(tag, rs, ddim, precision):
    """Reference p_mean_variance with edit_kwargs: process_xstart (denoised_fn, clip) on the model's x0 (:294-296), the masked
    replacement, then process_xstart AGAIN on the replaced x0 (:336-342).  With the non-idempotent denoised_fn of the fixture
    (clamp to +-0.5, times 0.9) the ground-truth rows end up clamped and scaled too -- applying it once misses the golden by O(1)."""
    from gpu_util import dev, rel
    g = load_golden("round3")
    m = _dit(SM, 11)
    d = _diff(rs)
    _inject(d, g[f"{tag}.noise"])
    ek = {"gt": dev(g["gt"]), "mask": dev(g["mask"]), "l_start": int(g["l_start"]), "l_end": int(g["l_end"]), "noise_level": 3}

    def dfn(v):
        return v.clamp(-0.5, 0.5) * 0.9
    kw = dict(clip_denoised=bool(int(g[f"{tag}.clip"])), denoised_fn=dfn, model_kwargs={"y": dev(g["y"])}, edit_kwargs=ek)
    x, t = dev(g["x"]), dev(g[f"{tag}.t"])
    out = d.ddim_sample(_model_fn(m), x, t, eta=1.0, **kw) if ddim else d.p_sample(_model_fn(m), x, t, **kw)
    # pred_xstart is clamped to +-0.45 here, so the usual norm-wise 2e-4 is stated on the scale of the UNclamped x0 estimate (~ 3)
    assert np.abs(out["pred_xstart"].cpu().numpy() - g[f"{tag}.pred_xstart"]).max() < 2e-4 * 3
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4
    assert float(out["pred_xstart"].abs().max()) <= 0.45 + 1e-6          # the replaced rows went through denoised_fn as well


@pytest.mark.parametrize("tag,rs,ddim", [("prevx_ddpm", "", False), ("prevx_ddim", "ddim50", True)])
def test_previous_x_mean_type(tag, rs, ddim, precision):
    """ModelMeanType.PREVIOUS_X (reference :331-338): the network's output IS the posterior mean; pred_xstart comes from
    _predict_xstart_from_xprev (:374-384) and is clipped, the mean is not."""
    from gpu_util import dev, rel
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    g = load_golden("round3")
    m = _dit(SM, 11)
    d = SpacedDiffusion(use_timesteps=space_timesteps(1000, rs or [1000]), betas=gd.get_named_beta_schedule("linear", 1000),
                        model_mean_type=gd.ModelMeanType.PREVIOUS_X, model_var_type=gd.ModelVarType.FIXED_LARGE,
                        loss_type=gd.LossType.MSE, rescale_timesteps=False)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])
    kw = dict(clip_denoised=bool(int(g[f"{tag}.clip"])), model_kwargs={"y": dev(g["y"])})
    x, t = dev(g["x"]), dev(g[f"{tag}.t"])
    out = d.ddim_sample(_model_fn(m), x, t, eta=1.0, **kw) if ddim else d.p_sample(_model_fn(m), x, t, **kw)
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4
    if ddim:
        assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 2e-4
    else:
        # x0 = x_prev / coef1 - (coef2 / coef1) x_t with 1 / coef1 = 241 at t = 420: the network's ~1e-5 arithmetic noise becomes
        # ~3e-3 on the few entries the clip leaves inside (-1, 1) -- in the reference's own fp32 evaluation just the same
        inv_c1 = 1.0 / float(d.posterior_mean_coef1[int(g[f"{tag}.t"][0])])
        assert 200 < inv_c1 < 300
        a, b = out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]
        net = 3e-5 if precision == "fp32" else 2e-4             # absolute arithmetic noise of the network output (values ~ 3)
        assert np.abs(a - b).max() < inv_c1 * net and (np.abs(b) == 1).mean() > 0.99 and (a[np.abs(b) == 1] == b[np.abs(b) == 1]).mean() > 0.999


def _learned_model(g):
    from gpu_util import load_module
    from guided_diffusion.dit import DiTRotary
    sd = synth.dit_state_dict(int(g["lsig.seed"]), **dict(SM, out_ch=8))
    for k in ("final_layer.linear.weight", "final_layer.linear.bias"):
        sd[k] = sd[k] * F32(float(g["lsig.final_gain"]))
    return load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=3,
                                 learn_sigma=True), sd)


def test_reference_raises_on_scg_and_dps_with_learned_variances_and_so_does_dps_here():
    """The fixture records what the reference does with a learn_sigma=True network on the SCG and DPS paths: an AssertionError in
    _predict_xstart_from_eps (the 2C-channel output is never split there).  DPS keeps raising here (nothing to reproduce); SCG is
    completed the way p_mean_variance splits the output (next test)."""
    from gpu_util import dev
    from guided_diffusion.condition_functions import composite_nn_zt
    from test_gpu_pins2 import _cls
    g = load_golden("round3")
    ref = json.loads(str(g["reference_raises"]))
    assert ref == {"scg_learned": "AssertionError in _predict_xstart_from_eps", "dps_learned": "AssertionError in _predict_xstart_from_eps"}
    m = _learned_model(g)
    d = _diff("250", learn_sigma=True)
    cond = partial(composite_nn_zt, fns=["nn_z0_mse_dummy"], classifier_scales=[1.], classifiers=[_cls()], rule_names=["note_density"])
    with pytest.raises(NotImplementedError, match="reference"):
        d.p_sample(_model_fn(m), dev(g["x"]), dev(np.full((2,), 100, dtype=np.int64)), clip_denoised=False, cond_fn=cond,
                   guidance_kwargs=SimpleNamespace(schedule=False, method="dps", step_size=1.5, nn=True, vae=False),
                   model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["lsig.target.note_density"])}})


def test_scg_with_learned_per_element_noise_scale(monkeypatch, precision):
    """SCG on a learn_sigma=True network: mean and the TENSOR g_coeff = exp(0.5 log_variance) from the learned-range interpolation
    (reference :299-313, :706-711), candidates mean + g (.) noise per element, the candidates' eps = the first C channels.  Golden: the
    reference's own p_sample / scg_sample driven with exactly that (make_golden.py round3): same winners, sample <= 2e-4; the
    per-element scale and the mean are checked against the reference's p_mean_variance too.  Then the Philox path: unsharded vs
    'rank r of 2' replay, winners rebuilt on the device with the per-element scale, bit-identical."""
    from gpu_util import dev, rel
    from rgm import scg_shard
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    g = load_golden("round3")
    m, vae = _learned_model(g), _vae(2)
    x, t, y = dev(g["x"]), dev(g["lsig.t"]), dev(g["y"])
    tgt = {"pitch_hist": dev(g["lsig.target.pitch_hist"]), "note_density": dev(g["lsig.target.note_density"])}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance")
    scg = {"num_samples": 4, "pitch_hist": 40., "note_density": 1.}
    kw = dict(clip_denoised=False, model_kwargs={"y": y, "rule": tgt}, embed_model=vae, scale_factor=1.2465, guidance_kwargs=guid, scg_kwargs=scg)
    d = _diff("", learn_sigma=True)
    # the step's mean and per-element noise scale
    eps = d._model_eps(x, _model_fn(m)(x, t, y=y), t, None)
    mean, _, gel = d._step("ddpm", x, eps, None, None, t, False, want_g=True)
    assert gel.shape == x.shape and rel(mean.cpu().numpy(), g["lsig.mean"]) < 2e-4
    assert np.abs(np.log(gel.cpu().numpy()) - np.log(g["lsig.g"])).max() < 2e-3          # exp(0.5 logvar): compare in the log domain
    d = _diff("", learn_sigma=True)
    _inject(d, g["lsig.noise"])
    out = d.p_sample(_model_fn(m), x, t, **kw)
    assert d.last_scg["max_ind"].cpu().tolist() == g["lsig.max_ind"].tolist()
    assert rel(out["sample"].cpu().numpy(), g["lsig.sample"]) < 2e-4

    def run():
        dd = _diff("", learn_sigma=True)
        dd.noise = PhiloxNoise(seed=7)
        o = dd.p_sample(_model_fn(m), x, t, **kw)
        return o["sample"], dd.last_scg["total_log_prob"].clone(), dd.last_scg["max_ind"].clone()
    ref_sample, ref_total, ref_idx = run()
    for rank in (0, 1):
        monkeypatch.setattr(scg_shard, "partition", lambda n, r=rank: (r * n // 2, n // 2, True))

        def fake_gather(local, r=rank):
            assert torch.equal(local, ref_total[r * 2:(r + 1) * 2])
            parts = [ref_total[:2], ref_total[2:]]
            parts[r] = local
            return torch.cat(parts, dim=0)
        monkeypatch.setattr(scg_shard, "gather_totals", fake_gather)
        s, total, idx = run()
        assert torch.equal(idx, ref_idx) and torch.equal(s, ref_sample), f"rank {rank}: rebuilt winner differs"


def test_record_statistics_survive_candidate_sharding(monkeypatch):
    """--record under sharding (reference :594-632): each_loss of the winner (per-rule tables ride the same all-gather as the totals) and
    the intermediate piano roll (decoded x0 estimate of the rebuilt winner) are kept when the candidates are split over ranks; the
    'rank r of 2' replay reports what the unsharded step reports."""
    from gpu_util import dev
    from rgm import scg_shard
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    from test_gpu_sampler import _diffusion
    g = load_golden("steps")
    m, vae = _dit(SM, 11), _vae(2)
    tgt = {"pitch_hist": dev(g["scg.target.pitch_hist"]), "note_density": dev(g["scg.target.note_density"])}
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance")
    scg = {"num_samples": 8, "pitch_hist": 40., "note_density": 1.}
    x = dev(g["x"])
    t = torch.full((x.shape[0],), 499, dtype=torch.int64, device="cuda")          # (t + 1) % 100 == 0: a roll is kept

    def run():
        d = _diffusion("")
        d.t_end = 0
        d.noise = PhiloxNoise(seed=99)
        d._reset_records(True, x.shape, x.device)
        d.p_sample(_model_fn(m), x, t, clip_denoised=False, model_kwargs={"y": dev(g["y"]), "rule": tgt}, embed_model=vae,
                   scale_factor=1.2465, guidance_kwargs=guid, scg_kwargs=scg, record=True)
        return d
    ref = run()
    assert set(ref.each_loss) == {"pitch_hist", "note_density"} and len(ref.inter_piano_rolls) == 1 and len(ref.log_probs) == 1
    locals_ = {}
    for rank in (0, 1):                                     # pass 1: what each rank contributes to the all-gather
        monkeypatch.setattr(scg_shard, "partition", lambda n, r=rank: (r * n // 2, n // 2, True))

        def grab(local, r=rank):
            locals_[r] = local.clone()
            return torch.cat([local, local], dim=0)
        monkeypatch.setattr(scg_shard, "gather_totals", grab)
        run()
    B = x.shape[0]
    assert locals_[0].shape == (4, 3 * B)                    # totals + two rules in ONE table
    for rank in (0, 1):
        monkeypatch.setattr(scg_shard, "partition", lambda n, r=rank: (r * n // 2, n // 2, True))

        def gather(local, r=rank):
            assert torch.equal(local, locals_[r])
            return torch.cat([locals_[0], locals_[1]], dim=0)
        monkeypatch.setattr(scg_shard, "gather_totals", gather)
        d = run()
        assert d.log_probs == ref.log_probs and d.loss_std == ref.loss_std and d.loss_range == ref.loss_range
        assert dict(d.each_loss) == dict(ref.each_loss), rank
        a, b = d.inter_piano_rolls[0].numpy(), ref.inter_piano_rolls[0].numpy()
        assert a.shape == b.shape and a.dtype == np.uint8
        assert (a != b).mean() < 1e-3                       # the winner's x0 recomputed in a batch of B instead of n.B rows: boundary flips only

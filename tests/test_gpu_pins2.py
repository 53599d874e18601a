"""-m gpu: the sampler paths round 1 left unpinned, against goldens produced by the reference (tests/golden/make_golden.py
steps2 / next2 / cli2): DDIM + classifier guidance (condition_score), DDIM + SCG, segment-wise SCG (guidance.dc.base) on a
256-row latent and on demo2.yml's circle collage, classifier-free guidance through model_fn / dc_model_fn, DPS under
edit_kwargs, DPS + SCG, grad_nn_zt_xentropy, and a 2-step scripts/sample_rule.py run (uint8 roll + results.csv rows).
Noise is teacher-forced; arrays the fixtures store as seeds are RandomState(seed).randn(shape) like the generator's."""
import importlib.util
import json
import os
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import PKG, load_golden
from rgm import synth
from test_gpu_sampler import SM, _diffusion, _dit, _inject, _model_fn, _vae

pytestmark = pytest.mark.gpu
F32 = np.float32
CLS2 = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
SCHED = dict(schedule=True, t_start=750, t_end=0, interval=1)


def _noise(g, tag, *shape):
    return np.random.RandomState(int(g[f"{tag}.noise_seed"])).randn(*shape).astype(F32)


def _cls():
    from gpu_util import load_module
    from guided_diffusion.dit import DiTRotaryClassifier
    m = DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=16)
    return load_module(m, synth.dit_state_dict(4, **CLS2))


def _cond(cm, fn="grad_nn_zt_mse", scale=10.):
    from guided_diffusion.condition_functions import composite_nn_zt
    return partial(composite_nn_zt, fns=[fn], classifier_scales=[scale], classifiers=[cm], rule_names=["note_density"])


def _targets(g, nd="target.note_density"):
    from gpu_util import dev
    return {"pitch_hist": dev(g["target.pitch_hist"]), "note_density": dev(g[nd])}


def test_ddim_classifier_guidance_is_condition_score(precision):
    """a4: ddim_sample(cond_fn=composite_nn_zt) -- eps -= sqrt(1 - abar) * grad, x0 and the mean re-derived (reference :467-489)."""
    from gpu_util import dev, rel
    g = load_golden("steps2")
    m, cm = _dit(SM, 11), _cls()
    d = _diffusion("ddim50")
    d.t_end = 0
    nz = _noise(g, "dcg", 2, 4, 128, 16)
    kw = dict(clip_denoised=False, eta=1.0)
    _inject(d, nz, nz)
    out = d.ddim_sample(_model_fn(m), dev(g["x"]), dev(g["dcg.t"]), cond_fn=_cond(cm),
                        model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["cg.rule"])}},
                        guidance_kwargs=SimpleNamespace(schedule=False, method="classifier_guidance"), **kw)
    plain = d.ddim_sample(_model_fn(m), dev(g["x"]), dev(g["dcg.t"]), model_kwargs={"y": dev(g["y"])}, **kw)
    assert rel(out["sample"].cpu().numpy(), g["dcg.sample"]) < 2e-4
    assert rel(out["pred_xstart"].cpu().numpy(), g["dcg.pred_xstart"]) < 2e-4
    shift = (out["sample"] - plain["sample"]).cpu().numpy()                   # the guidance term on its own
    assert np.abs(g["dcg.shift"]).max() > 1e-3 and rel(shift, g["dcg.shift"]) < 5e-3
    # the method of the same name is the same arithmetic (what the reference's ddim_sample calls)
    pmv = d.p_mean_variance(_model_fn(m), dev(g["x"]), dev(g["dcg.t"]), clip_denoised=False, model_kwargs={"y": dev(g["y"])})
    cs = d.condition_score(_cond(cm), pmv, dev(g["x"]), dev(g["dcg.t"]),
                           model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["cg.rule"])}})
    assert rel(cs["pred_xstart"].cpu().numpy(), g["dcg.pred_xstart"]) < 2e-4


@pytest.mark.parametrize("tag", ["dscg", "dscgc"])
def test_ddim_scg_step_selects_the_reference_candidates(tag, precision):
    """a4/a8: DDIM + SCG (g_coeff = sigma, the WRAPPED model scores the candidates; reference :933-954), without and with the
    classifier's condition_score applied first."""
    from gpu_util import dev, rel
    g = load_golden("steps2")
    m, vae = _dit(SM, 11), _vae(2)
    use_c = tag == "dscgc"
    d = _diffusion("ddim50")
    d.t_end = 0
    _inject(d, _noise(g, tag, 4, 2, 4, 128, 16))
    guid = SimpleNamespace(method="classifier_guidance" if use_c else "no_guidance", **SCHED)
    out = d.ddim_sample(_model_fn(m), dev(g["x"]), dev(g[f"{tag}.t"]), clip_denoised=False, eta=1.0, cond_fn=_cond(_cls()) if use_c else None,
                        model_kwargs={"y": dev(g["y"]), "rule": _targets(g)}, embed_model=vae, scale_factor=1.2465,
                        guidance_kwargs=guid, scg_kwargs={"num_samples": 4, "pitch_hist": 40., "note_density": 1.})
    assert np.array_equal(d.last_scg["max_ind"].cpu().numpy(), g[f"{tag}.max_ind"])
    assert rel(d.last_scg["total_log_prob"].cpu().numpy(), g[f"{tag}.total_log_prob"]) < 1e-4
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4
    assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 2e-4


def test_segmentwise_scg_picks_the_reference_winner_per_segment(precision):
    """a8, guidance.dc.base = 128 on a 256-row latent reached the reference's way (a 3-window linear DiffCollage behind
    dc_model_fn): two 1024-frame segments, each with its own argmax; note_density targets cut to the segment's windows
    (rule_base = 8), pitch_hist whole (reference :562-592)."""
    import diff_collage as dc
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import dc_model_fn
    g = load_golden("seg")
    m, vae = _dit(SM, 11), _vae(2)

    def eps_fn(x, t, y=None):
        return m(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
    worker = dc.CondIndSimple((4, 16, 128), eps_fn, 3, overlap_size=64)
    assert tuple(worker.shape) == (4, 16, 256)
    mf = partial(dc_model_fn, model=worker.eps_scalar_t_fn, num_classes=3, class_cond=True, cfg=False, w=0.)
    d = _diffusion("")
    d.t_end = 0
    _inject(d, np.random.RandomState(int(g["noise_seed"])).randn(3, 2, 4, 256, 16).astype(F32))
    guid = SimpleNamespace(method="no_guidance", dc=SimpleNamespace(base=128), **SCHED)
    out = d.p_sample(mf, dev(g["x"]), dev(g["t"]), clip_denoised=False, model_kwargs={"y": dev(g["y"]), "rule": _targets(g)},
                     embed_model=vae, scale_factor=1.2465, guidance_kwargs=guid,
                     scg_kwargs={"num_samples": 3, "pitch_hist": 100., "note_density": 1.})
    assert d.last_scg["max_ind"].shape == (2, 2)
    assert np.array_equal(d.last_scg["max_ind"].cpu().numpy(), g["max_ind"])
    assert rel(d.last_scg["total_log_prob"].cpu().numpy(), g["total_log_prob"]) < 1e-4
    assert rel(out["pred_xstart"].cpu().numpy(), g["pred_xstart"]) < 2e-4
    assert rel(out["sample"].cpu().numpy(), g["sample"]) < 2e-4


def _circle_worker(m):
    import diff_collage as dc

    def eps_fn(x, t, y=None):                                                        # scripts/sample_rule.py:120-122
        return m(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
    w = dc.CondIndCircle((4, 16, 128), eps_fn, 2, overlap_size=64)                    # demo2.yml: num_img 1 (+1 for the circle)
    assert tuple(w.shape) == (4, 16, 128)
    return w


def test_demo2_circle_collage_with_segmentwise_scg(precision):
    """The reference's cond_demo/demo2.yml: diff_collage circle (2 windows over a 128-row ring) behind dc_model_fn + SCG with
    dc.base 128 (one segment) and pitch_hist weight 100."""
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import dc_model_fn
    g = load_golden("steps2")
    m, vae = _dit(SM, 11), _vae(2)
    mf = partial(dc_model_fn, model=_circle_worker(m).eps_scalar_t_fn, num_classes=3, class_cond=True, cfg=False, w=0.)
    d = _diffusion("")
    d.t_end = 0
    _inject(d, _noise(g, "circ", 3, 2, 4, 128, 16))
    guid = SimpleNamespace(method="no_guidance", dc=SimpleNamespace(base=128), **SCHED)
    out = d.p_sample(mf, dev(g["x"]), dev(g["circ.t"]), clip_denoised=False, model_kwargs={"y": dev(g["y"]), "rule": _targets(g)},
                     embed_model=vae, scale_factor=1.2465, guidance_kwargs=guid,
                     scg_kwargs={"num_samples": 3, "pitch_hist": 100., "note_density": 1.})
    assert np.array_equal(d.last_scg["max_ind"].cpu().numpy(), g["circ.max_ind"])
    assert rel(d.last_scg["total_log_prob"].cpu().numpy(), g["circ.total_log_prob"]) < 1e-4
    assert rel(out["pred_xstart"].cpu().numpy(), g["circ.pred_xstart"]) < 2e-4
    assert rel(out["sample"].cpu().numpy(), g["circ.sample"]) < 2e-4


def test_cfg_outputs_match_the_reference_model_fn(precision):
    """f4: model_fn(cfg=True, w=4) = (1 + w) m(x, t, y) - w m(x, t, y_null) and dc_model_fn(cfg=True) around the circle worker,
    against the reference's own outputs; class_cond=False runs the null label only."""
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import dc_model_fn, model_fn
    g = load_golden("steps2")
    m = _dit(SM, 11)
    x, t, y = dev(g["x"]), dev(g["cfg.t"]), dev(g["y"])
    assert rel(model_fn(x, t, y, model=m, num_classes=3, class_cond=True, cfg=True, w=4.).cpu().numpy(), g["cfg.eps"]) < 2e-4
    out = dc_model_fn(x, t, y, model=_circle_worker(m).eps_scalar_t_fn, num_classes=3, class_cond=True, cfg=True, w=4.)
    assert rel(out.cpu().numpy(), g["cfg.dc_eps"]) < 2e-4
    assert rel(model_fn(x, t, y, model=m, num_classes=3, class_cond=False, cfg=True, w=4.).cpu().numpy(), g["cfg.uncond_eps"]) < 2e-4


def test_grad_nn_zt_xentropy_matches_autograd_golden(precision):
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import function_map
    g = load_golden("next2")
    grad = function_map["grad_nn_zt_xentropy"](dev(g["x"]), rule=dev(g["xent.rule"]), classifier=_cls())
    assert rel(grad.cpu().numpy(), g["xent.grad"]) < 5e-4


def test_dps_under_edit_kwargs(precision):
    """DPS with replacement conditioning (reference :426-428, :453-455): whole latent editable, the first 32 rows pinned to the
    ground truth by the mask; a partial editable range cannot broadcast in the reference and raises here."""
    from gpu_util import dev, rel
    g = load_golden("next2")
    m, cm = _dit(SM, 11), _cls()
    d = _diffusion("250")
    d.t_end = 0
    ek = {"gt": dev(g["gt"]), "mask": dev(g["mask"]), "l_start": 0, "l_end": 128, "noise_level": 3}
    gk = SimpleNamespace(schedule=False, method="dps", step_size=1.5, nn=True, vae=False)
    kw = dict(clip_denoised=True, edit_kwargs=ek)
    _inject(d, g["dpse.noise"], g["dpse.noise"])
    mk = {"y": dev(g["y"]), "rule": {"note_density": dev(g["rule"])}}
    out = d.p_sample(_model_fn(m), dev(g["x"]), dev(g["dpse.t"]), cond_fn=_cond(cm, "nn_z0_mse_dummy", 1.), model_kwargs=mk,
                     guidance_kwargs=gk, **kw)
    plain = d.p_sample(_model_fn(m), dev(g["x"]), dev(g["dpse.t"]), model_kwargs={"y": dev(g["y"])}, **kw)
    assert rel(out["sample"].cpu().numpy(), g["dpse.sample"]) < 5e-4
    assert rel(out["pred_xstart"].cpu().numpy(), g["dpse.pred_xstart"]) < 5e-4
    assert rel((out["sample"] - plain["sample"]).cpu().numpy(), g["dpse.shift"]) < 5e-3
    with pytest.raises(ValueError):
        d.p_sample(_model_fn(m), dev(g["x"]), dev(g["dpse.t"]), cond_fn=_cond(cm, "nn_z0_mse_dummy", 1.), model_kwargs=mk,
                   guidance_kwargs=gk, clip_denoised=True, edit_kwargs=dict(ek, l_start=32, l_end=96))


def test_dps_combined_with_scg(precision):
    """p_sample with method dps AND scg_kwargs (reference :691-733, cond_table/all/scg_dps_nn_all.yml): the DPS shift is applied on
    every step; inside the schedule the candidates branch from the shifted mean, outside it the step draws once."""
    from gpu_util import dev, rel
    g = load_golden("next2")
    m, cm, vae = _dit(SM, 11), _cls(), _vae(2)
    gs = SimpleNamespace(method="dps", step_size=1.5, nn=True, vae=True, **SCHED)
    mk = {"y": dev(g["y"]), "rule": {"note_density": dev(g["rule"])}}
    for tag, shape in (("dpsscg", (3, 2, 4, 128, 16)), ("dpsscg_off", (2, 4, 128, 16))):
        d = _diffusion("")
        d.t_end = 0
        _inject(d, _noise(g, tag, *shape))
        out = d.p_sample(_model_fn(m), dev(g["x"]), dev(g[f"{tag}.t"]), clip_denoised=False, cond_fn=_cond(cm, "nn_z0_mse_dummy", 1.),
                         model_kwargs=mk, embed_model=vae, scale_factor=1.2465, guidance_kwargs=gs,
                         scg_kwargs={"num_samples": 3, "note_density": 1.})
        if tag == "dpsscg":
            assert np.array_equal(d.last_scg["max_ind"].cpu().numpy(), g["dpsscg.max_ind"])
        assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 5e-4
        assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 5e-4
    # a DPS cond_fn under ddim_sample is rejected (the reference's condition_score cannot broadcast its (B,) output either)
    d = _diffusion("ddim50")
    with pytest.raises(NotImplementedError):
        d.ddim_sample(_model_fn(m), dev(g["x"]), dev(np.array([5, 5])), cond_fn=_cond(cm, "nn_z0_mse_dummy", 1.), model_kwargs=mk,
                      guidance_kwargs=SimpleNamespace(schedule=False, method="dps", step_size=1., nn=True, vae=False))


def test_step_rejects_a_log_prob_where_a_gradient_is_expected():
    """a cond_fn returning (B,) log-probabilities must not reach the step kernel as `grad` (out-of-bounds read)."""
    from gpu_util import dev
    d = _diffusion("")
    x = dev(np.zeros((2, 4, 128, 16), dtype=F32))
    with pytest.raises(AssertionError):
        d._step("ddpm", x, x, torch.zeros(2, device="cuda"), None, dev(np.array([3, 3])), False)


def test_two_step_sample_rule_cli_reproduces_the_reference_roll_and_losses(tmp_path, monkeypatch, precision):
    """a13: scripts/sample_rule.py on a 2-step stochastic-DDIM chain with SCG (the YAML, the initial noise and the candidate noise
    of the reference run are in the fixture): the uint8 roll and the results.csv rows of the reference."""
    g = load_golden("cli2")
    monkeypatch.chdir(tmp_path)
    cfg = tmp_path / "configs" / "cond_demo" / "two_step.yml"
    cfg.parent.mkdir(parents=True)
    cfg.write_text(str(g["config_yaml"]))
    spec = importlib.util.spec_from_file_location("sample_rule_cli2", os.path.join(PKG, "scripts", "sample_rule.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    q = [np.random.RandomState(int(g["xT_seed"])).randn(2, 4, 128, 16).astype(F32),
         np.random.RandomState(int(g["scg_noise_seed"])).randn(4, 2, 4, 128, 16).astype(F32)]

    def noise_fn(shape, device):
        z = q.pop(0)
        assert tuple(z.shape) == tuple(shape), (z.shape, shape)
        return torch.from_numpy(z).to(device)
    cli.NOISE_FN = noise_fn
    cli.KEEP_FLOAT_ROLLS = []
    res = cli.main(["--config_path", str(cfg), "--batch_size", "2", "--num_samples", "2", "--model", "DiTRotary_B_8", "--image_size", "128",
                    "16", "--in_channels", "4", "--scale_factor", "1.2465", "--class_cond", "True", "--num_classes", "3", "--class_label", "1",
                    "--synthetic_weights", "True", "--progress", "False", "--gemm_precision", precision])
    assert not q, "the CLI drew less noise than the reference run"
    out_dir = os.path.join("loggings", cli.output_dir_for(str(cfg), 1))
    u8 = np.stack([np.load(os.path.join(out_dir, f"sample_{i}_y_1.npy")) for i in range(2)])        # (B,3,128,T)
    assert u8.shape == g["u8"].shape and u8.dtype == np.uint8
    bad = u8 != g["u8"]
    # like the library-path tests: EVERY differing entry must sit on a quantisation boundary of this run's own float roll (the CLI keeps
    # it for the test: KEEP_FLOAT_ROLLS), and their number stays bounded
    from gpu_util import u8_flip_report
    roll = np.concatenate(cli.KEEP_FLOAT_ROLLS, axis=0)                       # (B,3,128,T) float
    n_bad, unexplained, far = u8_flip_report(u8.transpose(0, 2, 3, 1), g["u8"].transpose(0, 2, 3, 1), roll)
    assert n_bad == int(bad.sum()) and unexplained == 0, (n_bad, unexplained, far)
    # the CLI's final decode is the exact-fp32 one (midi_util.FINAL_DECODE_EXACT): what is left in the bf16x3 modes comes from the two noisy
    # latents themselves (measured 164 of 786 432; with the decode in the loop's arithmetic: 270)
    assert bad.mean() < (1.5e-4 if precision == "fp32" else 3e-4), bad.sum()
    ref = json.loads(str(g["results_json"]))
    assert list(res.columns) == list(g["columns"])
    import pandas as pd
    df = pd.read_csv(os.path.join(out_dir, "results.csv"))
    assert list(df.columns) == list(g["columns"])
    for col in ref:
        a = np.array([np.asarray(v, dtype=np.float64) for v in res[col]])
        b = np.array([np.asarray(v, dtype=np.float64) for v in ref[col]])
        tol = 1e-6 if "target_rule" in col else 2e-3
        assert np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-30), (col, a, b)
    print(f"[cli2 {precision}] uint8 mismatches {bad.sum()} / {bad.size}; losses {res.filter(like='.loss').values.tolist()}")


@pytest.mark.parametrize("tag,rs,ddim,px", [("dfn_ddpm", "", False, False), ("dfn_ddim", "ddim50", True, False),
                                           ("x0_ddpm", "250", False, True), ("x0_ddim", "ddim50", True, True)])
def test_denoised_fn_and_x0_predicting_models(tag, rs, ddim, px, precision):
    """denoised_fn (applied to the x0 estimate before the clip, reference :281-286) and predict_xstart=True (ModelMeanType.START_X,
    :323-333) through the fused step, against the reference's outputs."""
    from gpu_util import dev, rel
    from guided_diffusion.script_util import create_diffusion
    g = load_golden("hooks")
    m = _dit(SM, 11)
    d = create_diffusion(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing=rs, use_kl=False,
                         predict_xstart=px, rescale_timesteps=False, rescale_learned_sigmas=False)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])
    kw = dict(clip_denoised=True, denoised_fn=None if px else (lambda v: v.clamp(-0.5, 0.5) * 0.9), model_kwargs={"y": dev(g["y"])})
    x, t = dev(g["x"]), dev(g[f"{tag}.t"])
    out = d.ddim_sample(_model_fn(m), x, t, eta=1.0, **kw) if ddim else d.p_sample(_model_fn(m), x, t, **kw)
    assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 2e-4
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4


def test_dps_with_classifier_free_guidance_in_the_eps_network(precision):
    """DPS through model_fn(cfg=True, w=4) (the reference differentiates both conditional and null-label passes with autograd):
    one 2B-row saved-activation forward, one backward with the cotangents (1+w) g | -w g."""
    from gpu_util import dev, rel
    from guided_diffusion.condition_functions import model_fn
    g = load_golden("cfgdps")
    m, cm = _dit(SM, 11), _cls()
    mf = partial(model_fn, model=m, num_classes=3, class_cond=True, cfg=True, w=4.)
    d = _diffusion("250")
    d.t_end = 0
    _inject(d, g["noise"], g["noise"])
    gk = SimpleNamespace(schedule=False, method="dps", step_size=1.5, nn=True, vae=False)
    kw = dict(clip_denoised=False, model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["rule"])}})
    out = d.p_sample(mf, dev(g["x"]), dev(g["t"]), cond_fn=_cond(cm, "nn_z0_mse_dummy", 1.), guidance_kwargs=gk, **kw)
    plain = d.p_sample(mf, dev(g["x"]), dev(g["t"]), **kw)
    assert rel(out["pred_xstart"].cpu().numpy(), g["pred_xstart"]) < 5e-4
    assert rel(out["sample"].cpu().numpy(), g["sample"]) < 5e-4
    assert rel((out["sample"] - plain["sample"]).cpu().numpy(), g["shift"]) < 5e-3


@pytest.mark.parametrize("tag,rs,ddim,guided", [("ddpm", "", False, False), ("cg250", "250", False, True), ("ddim", "ddim50", True, False)])
def test_learned_variance_checkpoints(tag, rs, ddim, guided, precision):
    """learn_sigma=True (ModelVarType.LEARNED_RANGE, reference :299-313): the 2C-channel network output is split, the per-element
    log-variance is interpolated inside the fused DDPM step (rgm_ddpm_step_learned), classifier guidance scales its gradient by
    that variance, DDIM ignores it."""
    from gpu_util import dev, load_module, rel
    from guided_diffusion.dit import DiTRotary
    from guided_diffusion.script_util import create_diffusion
    g = load_golden("learned")
    arch = dict(SM, out_ch=8)
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=3,
                              learn_sigma=True), synth.dit_state_dict(int(g["seed"]), **arch))
    d = create_diffusion(learn_sigma=True, diffusion_steps=1000, noise_schedule="linear", timestep_respacing=rs, use_kl=False,
                         predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)
    d.t_end = 0
    _inject(d, g[f"{tag}.noise"])
    kw = dict(clip_denoised=False, model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["rule"])}})
    if guided:
        kw.update(cond_fn=_cond(_cls()), guidance_kwargs=SimpleNamespace(schedule=False, method="classifier_guidance"))
    x, t = dev(g["x"]), dev(g[f"{tag}.t"])
    out = d.ddim_sample(_model_fn(m), x, t, eta=1.0, **kw) if ddim else d.p_sample(_model_fn(m), x, t, **kw)
    assert rel(out["pred_xstart"].cpu().numpy(), g[f"{tag}.pred_xstart"]) < 2e-4
    assert rel(out["sample"].cpu().numpy(), g[f"{tag}.sample"]) < 2e-4

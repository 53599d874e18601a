"""CPU: the numpy oracle replayed against the golden vectors captured from the reference.

This is what pins oracle/ (SURVEY 8c): every fixture under tests/golden/ was produced by
tests/golden/make_golden.py importing /root/reference in the build container.  Tolerances are
fp32 re-association noise between numpy/OpenBLAS and torch/oneDNN (measured 2e-7 .. 4e-5).
"""
import numpy as np
import pytest
from conftest import load_golden, rel_err

from rgm import synth
from oracle import diffusion_np as odf, dit_np as odit, vae_np as ovae, rules_np as orl, collage_np as ocl

F32 = np.float32
XL2 = dict(depth=2, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
SM = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, out_ch=4, num_classes=3)
CLS = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
CLS2 = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
CHD = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=8, chord=True)


def test_schedule_tables_bit_exact():
    g = load_golden("schedule")
    for tag, rs in (("full", ""), ("ddim50", "ddim50"), ("r250", "250")):
        S = odf.Schedule(1000, "linear", rs)
        for k in ("timestep_map", "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_mean_coef1",
                  "posterior_mean_coef2", "model_variance"):
            assert np.array_equal(np.asarray(getattr(S, k)), g[f"{tag}.{k}"]), (tag, k)
    assert odf.Schedule(1000, "linear", "ddim50").timestep_map == list(range(0, 1000, 20))
    assert odf.Schedule(1000, "linear", "250").timestep_map[-3:] == [991, 995, 999]


def test_dit_forward_xl_width_depth2():
    g = load_golden("dit_xl_d2")
    sd = synth.dit_state_dict(int(g["seed"]), **XL2)
    for H in (128, 64):
        out = odit.dit_forward(sd, g[f"x{H}"], g[f"t{H}"], g[f"y{H}"], depth=2, heads=16)
        assert rel_err(out, g[f"out{H}"]) < 1e-4


@pytest.mark.parametrize("tag,arch", [("s8", CLS), ("s8d2", CLS2)])
def test_classifier_logits_and_guidance_gradient(tag, arch):
    g = load_golden("classifier")
    sd = synth.dit_state_dict(int(g[f"{tag}.seed"]), **arch)
    grad, logits = odit.grad_nn_zt_mse(sd, g[f"{tag}.x"], g[f"{tag}.t"], g[f"{tag}.rule"], 10.,
                                       depth=arch["depth"], heads=arch["heads"])
    assert rel_err(logits, g[f"{tag}.logits"]) < 1e-4
    assert rel_err(grad, g[f"{tag}.grad"]) < 2e-4


def test_chord_classifier_heads_and_gradient():
    g = load_golden("classifier")
    sd = synth.dit_state_dict(int(g["chord.seed"]), **CHD)
    grad, (key, ch) = odit.grad_nn_zt_chord(sd, g["chord.x"], g["chord.t"], g["chord.rule"], 10., depth=2, heads=6)
    assert key.shape == (2, 25) and ch.shape == (2, 8, 8)
    assert rel_err(key, g["chord.key"]) < 1e-4
    assert rel_err(ch, g["chord.logits"]) < 1e-4
    assert rel_err(grad, g["chord.grad"]) < 2e-4


def test_vae_decoder_and_uint8_roll():
    g = load_golden("vae_decoder")
    sd = synth.vae_state_dict(int(g["seed"]))
    out = ovae.decode(sd, g["z"])
    assert rel_err(out, g["out"]) < 2e-5
    dec = odf.decode_latent(g["lat"], lambda z: ovae.decode(sd, z), 1.2465)
    u8 = ovae.quantise_roll(dec)
    assert u8.shape == g["u8"].shape == (1, 128, 256, 3) and u8.dtype == np.uint8
    # fp32 re-association moves a handful of values across a truncation / threshold boundary
    bad = (u8 != g["u8"])
    assert bad.mean() < 2e-4
    assert np.abs(u8.astype(int) - g["u8"].astype(int))[bad].max() <= 3


def _sparse_roll(rng, n, T):
    r = -1 + 0.08 * rng.rand(n, 3, 128, T).astype(F32)
    for b in range(n):
        for _ in range(60 * T // 1024 + 5):
            p = rng.randint(0, 128)
            s = rng.randint(0, T - 8)
            L = rng.randint(4, 120)
            r[b, 0, p, s:s + L] = rng.uniform(-0.5, 1.0)
            r[b, 1, p, s] = 1.0
    return r.astype(F32)


def test_rules_bit_exact_counts_and_histogram():
    g = load_golden("rules")
    roll = _sparse_roll(np.random.RandomState(400), 3, 1024)
    for name in ("note_density", "note_density_hr_1", "note_density_hr_2", "note_density_class", "note_density_pixel"):
        r = roll.copy()
        out = orl.FUNC_DICT[name](r)
        assert np.array_equal(out, g[name]), name                    # integer counts: bit-exact
        assert float(r.astype(np.float64).sum()) == float(g[name + ".roll_after_sum"])   # in-place side effects
    r = roll.copy()
    assert rel_err(orl.pitch_hist(r), g["pitch_hist"]) < 1e-6
    r = roll.copy()
    orl.note_density(r)
    assert rel_err(orl.pitch_hist(r), g["pitch_hist_after_nd"]) < 1e-6   # rule order matters (in-place threshold)
    assert np.array_equal(orl.mse_loss_mean(g["note_density"], g["mse_target"]), g["mse_loss"])
    assert orl.pitch_hist(roll[:1].copy()).shape == g["pitch_hist_b1"].shape == (12,)
    assert orl.note_density(roll[:1].copy()).shape == g["note_density_b1"].shape == (16,)


def test_pitch_hist_gradient_matches_reference_autograd():
    """The written-out gradient of rule_x0_mse_dummy(pitch_hist) against what the reference's autograd returned (dps_rule golden)."""
    g = load_golden("dps_rule")
    r = (np.random.RandomState(int(g["ph.rseed"])).rand(2, 3, 128, 256).astype(np.float32) * 2 - 1) * 0.8
    logp, grad = orl.pitch_hist_logp_grad(r, g["ph.target"])
    assert rel_err(logp, g["ph.logp"]) < 1e-5
    assert np.abs(grad - grad[:, :, :, :1]).max() == 0
    assert rel_err(grad[:, :, :, 0], g["ph.grad_rows"]) < 1e-5
    lp2, g2 = orl.pitch_hist_logp_grad(r, g["ph.target"], scale=2.5)
    assert rel_err(lp2, 2.5 * logp) < 1e-6 and rel_err(g2, 2.5 * grad) < 1e-6


def test_chord_quantisation_matches_reference():
    """The integer roll get_chords hands to the (music21) analyser, and the side effects on the caller's roll."""
    from conftest import chord_test_roll
    g = load_golden("chord_quantise")
    roll = chord_test_roll(int(g["seed"]))
    q = orl.chord_quantise(roll)
    assert np.array_equal(q, g["q"].astype(np.intc))
    assert float(roll.astype(np.float64).sum()) == float(g["after_sum"]) and int((roll == -1).sum()) == int(g["after_minus1"])
    assert np.array_equal(roll[:, 0, 60], g["after_ch0_row60"])


def _np_model(sd, arch):
    def f(x, t, y=None, rule=None):
        return odit.dit_forward(sd, x, t, y, depth=arch["depth"], heads=arch["heads"])
    return f


def test_teacher_forced_steps():
    g = load_golden("steps")
    sd = synth.dit_state_dict(11, **SM)
    model = _np_model(sd, SM)
    x, y = g["x"], g["y"]
    for tag, rs, ddim in (("ddpm", "", False), ("ddim", "ddim50", True), ("ddpm250", "250", False)):
        S = odf.Schedule(1000, "linear", rs)
        if ddim:
            o = odf.ddim_sample(S, model, x, g[f"{tag}.t"], g[f"{tag}.noise"], eta=1.0, model_kwargs={"y": y})
        else:
            o = odf.p_sample(S, model, x, g[f"{tag}.t"], g[f"{tag}.noise"], model_kwargs={"y": y})
        assert rel_err(o["sample"], g[f"{tag}.sample"]) < 1e-4, tag
        assert rel_err(o["pred_xstart"], g[f"{tag}.pred_xstart"]) < 1e-4, tag


def test_classifier_guided_step():
    g = load_golden("steps")
    sd = synth.dit_state_dict(11, **SM)
    csd = synth.dit_state_dict(4, **CLS2)
    S = odf.Schedule(1000, "linear", "250")

    def cond(xx, tt, y=None, rule=None):
        return odit.grad_nn_zt_mse(csd, xx, tt, rule["note_density"], 10., depth=2, heads=6)[0]
    o = odf.p_sample(S, _np_model(sd, SM), g["x"], g["cg.t"], g["cg.noise"], cond_fn=cond,
                     model_kwargs={"y": g["y"], "rule": {"note_density": g["cg.rule"]}},
                     guidance={"schedule": False}, return_aux=True)
    assert rel_err(o["sample"], g["cg.sample"]) < 1e-4
    assert np.abs(o["aux"]["grad"]).max() > 1e-3                     # guidance actually moved the mean


def test_scg_branch_and_select_step():
    g = load_golden("steps")
    sd = synth.dit_state_dict(11, **SM)
    vsd = synth.vae_state_dict(2)
    S = odf.Schedule(1000, "linear", "")
    tgt = {"pitch_hist": g["scg.target.pitch_hist"], "note_density": g["scg.target.note_density"]}
    o = odf.p_sample(S, _np_model(sd, SM), g["x"], g["scg.t"], g["scg.noise"],
                     model_kwargs={"y": g["y"], "rule": tgt},
                     guidance=dict(schedule=True, t_start=750, t_end=0, interval=1),
                     scg_kwargs={"num_samples": 4, "pitch_hist": 40., "note_density": 1.},
                     decode_fn=lambda z: ovae.decode(vsd, z), scale_factor=1.2465,
                     func_dict=orl.FUNC_DICT, loss_dict=orl.LOSS_DICT, return_aux=True)
    assert np.array_equal(o["aux"]["max_ind"], g["scg.max_ind"])
    assert rel_err(o["sample"], g["scg.sample"]) < 1e-5
    assert rel_err(o["aux"]["total_log_prob"], g["scg.total_log_prob"]) < 1e-5


def test_diff_collage_split_merge_and_eps():
    g = load_golden("collage")
    w = g["w"]
    xs, ov = ocl.split_wimg(w, 7)
    assert ov == 64 and xs.shape == (14, 4, 16, 128)
    assert np.array_equal(xs[3], w[0, :, :, 192:320])              # window 3 of sample 0
    assert rel_err(ocl.merge_wimg(xs, 64, 7, True), g["merge_avg"]) < 1e-7
    sd = synth.dit_state_dict(11, **SM)

    def eps(x, t, y=None):
        return odit.dit_forward(sd, np.ascontiguousarray(x.transpose(0, 1, 3, 2)), t, y, depth=2, heads=6).transpose(0, 1, 3, 2)
    assert rel_err(ocl.condind_eps(w, g["t"], eps, 7, 64, y=g["y"]), g["eps_linear"]) < 1e-4
    assert rel_err(ocl.condind_eps(w, g["t"], eps, 8, 64, y=g["y"], circle=True), g["eps_circle"]) < 1e-4


def test_end_to_end_ddim50_small_model():
    g = load_golden("e2e_ddim50_sm")
    seed = int(g["seed"])
    sd = synth.dit_state_dict(seed, final_std=0.3 / SM["hidden"] ** 0.5, **SM)
    S = odf.Schedule(1000, "linear", "ddim50")
    rng = np.random.RandomState(700 + seed)
    xT = rng.randn(2, 4, 128, 16).astype(F32)
    nz = [rng.randn(2, 4, 128, 16).astype(F32) for _ in range(50)]
    lat = odf.sample_loop(S, _np_model(sd, SM), xT, nz, ddim=True, eta=1.0, model_kwargs={"y": g["y"]})
    assert rel_err(lat, g["latent"]) < 1e-3                          # north-star latent tolerance
    vsd = synth.vae_state_dict(2)
    u8 = ovae.quantise_roll(odf.decode_latent(lat, lambda z: ovae.decode(vsd, z), 1.2465))
    assert (u8 != g["u8"]).mean() < 1e-3


def test_vae_encoder_and_encode_latent():
    """Editing path: Encoder + quant_conv and the tiling of _encode, against the reference's outputs."""
    g = load_golden("edit")
    vsd = synth.vae_state_dict(int(g["seed"]), encoder=True)
    assert rel_err(ovae.encode_moments(vsd, g["tiles"]), g["moments"]) < 2e-5
    assert rel_err(ovae.encode_latent(vsd, g["roll"], 1.2465), g["latent"]) < 2e-5


def test_replacement_conditioned_steps():
    """edit_kwargs in p_sample / ddim_sample (reference p_mean_variance :293-298, condition_mean :408-414)."""
    g = load_golden("edit")
    sd = synth.dit_state_dict(11, **SM)
    model = _np_model(sd, SM)
    edit = {"gt": g["gt"], "mask": g["mask"], "l_start": int(g["l_start"]), "l_end": int(g["l_end"])}
    for tag, rs, ddim, clip in (("ddpm", "", False, True), ("ddim", "ddim50", True, False)):
        S = odf.Schedule(1000, "linear", rs)
        kw = dict(clip_denoised=clip, model_kwargs={"y": g["y"]}, edit=edit)
        o = odf.ddim_sample(S, model, g["x"], g[f"{tag}.t"], g[f"{tag}.noise"], eta=1.0, **kw) if ddim \
            else odf.p_sample(S, model, g["x"], g[f"{tag}.t"], g[f"{tag}.noise"], **kw)
        assert rel_err(o["sample"], g[f"{tag}.sample"]) < 1e-4, tag
        assert rel_err(o["pred_xstart"], g[f"{tag}.pred_xstart"]) < 1e-3, tag   # c2 ~ 6 amplifies the eps noise here
    csd = synth.dit_state_dict(4, **CLS2)
    S = odf.Schedule(1000, "linear", "250")

    def cond(xx, tt, y=None, rule=None):
        return odit.grad_nn_zt_mse(csd, xx, tt, rule["note_density"], 10., depth=2, heads=6)[0]
    full = dict(edit, mask=np.zeros_like(g["mask"]), l_start=0, l_end=128)
    o = odf.p_sample(S, model, g["x"], g["cg.t"], g["cg.noise"], cond_fn=cond,
                     model_kwargs={"y": g["y"], "rule": {"note_density": g["cg.rule"]}}, guidance={"schedule": False}, edit=full)
    assert rel_err(o["sample"], g["cg.sample"]) < 1e-4


def test_ddim_condition_score_and_cfg_against_round2_goldens():
    """steps2.npz: DDIM + classifier guidance (condition_score, reference :467-489) and classifier-free guidance through
    model_fn / dc_model_fn around the one-window circle collage (condition_functions.py:17-42) -- the oracle vs the reference."""
    g = load_golden("steps2")
    sd = synth.dit_state_dict(11, **SM)
    csd = synth.dit_state_dict(4, **CLS2)

    def omf(x, t, y=None, rule=None):
        return odit.dit_forward(sd, x, t, y, depth=2, heads=6)

    def ocond(xx, tt, y=None, rule=None):
        return odit.grad_nn_zt_mse(csd, xx, tt, rule["note_density"], 10., depth=2, heads=6)[0]
    S = odf.Schedule(1000, "linear", "ddim50")
    nz = np.random.RandomState(int(g["dcg.noise_seed"])).randn(2, 4, 128, 16).astype(F32)
    o = odf.ddim_sample(S, omf, g["x"], g["dcg.t"], nz, eta=1.0, cond_fn=ocond, model_kwargs={"y": g["y"], "rule": {"note_density": g["cg.rule"]}},
                        guidance={"schedule": False})
    assert rel_err(o["sample"], g["dcg.sample"]) < 1e-4 and rel_err(o["pred_xstart"], g["dcg.pred_xstart"]) < 1e-4
    u = odf.ddim_sample(S, omf, g["x"], g["dcg.t"], nz, eta=1.0, model_kwargs={"y": g["y"]})
    assert rel_err(o["sample"] - u["sample"], g["dcg.shift"]) < 2e-3
    net = lambda a, b, c: odit.dit_forward(sd, a, b, c, depth=2, heads=6)           # noqa: E731
    assert rel_err(odf.model_fn(net, g["x"], g["cfg.t"], g["y"], cfg=True, w=4.), g["cfg.eps"]) < 1e-4
    assert rel_err(odf.model_fn(net, g["x"], g["cfg.t"], g["y"], class_cond=False), g["cfg.uncond_eps"]) < 1e-4

    def oeps(xx, tt, y=None):
        return odit.dit_forward(sd, np.ascontiguousarray(xx.transpose(0, 1, 3, 2)), tt, y, depth=2, heads=6).transpose(0, 1, 3, 2)
    circ = lambda a, b, c: ocl.condind_eps(a, b, oeps, 2, 64, y=c, circle=True)     # noqa: E731
    assert rel_err(odf.model_fn(circ, g["x"], g["cfg.t"], g["y"], cfg=True, w=4., transpose=True), g["cfg.dc_eps"]) < 1e-4


def test_segmentwise_selection_index_arithmetic():
    """oracle _scg_segments (reference :562-592) on a synthetic decoded roll: per-segment argmax with the note_density target cut to
    the segment's windows -- checked against a direct per-segment evaluation, and against steps2.npz's recorded winners' shape."""
    rng = np.random.RandomState(3)
    n, B = 3, 2
    x0 = (-1 + 0.1 * rng.rand(n * B, 1, 128, 2048)).astype(F32)
    x0[:, 0, 40:80, ::7] = 0.5
    cand = rng.randn(n * B, 4, 256, 16).astype(F32)
    tgt = {"note_density": (rng.rand(B, 32) * 5).astype(F32)}
    sample, inds, totals = odf._scg_segments(cand, x0, B, n, {"rule": tgt}, {"note_density": 1.}, orl.FUNC_DICT, orl.LOSS_DICT, 128)
    assert sample.shape == (B, 4, 256, 16) and inds.shape == (2, B) and totals.shape == (n, 2, B)
    for i in range(2):
        gen = orl.note_density(np.ascontiguousarray(x0[:, :, :, i * 1024:(i + 1) * 1024]).copy())
        t_i = np.concatenate((tgt["note_density"][:, :16][:, i * 8:(i + 1) * 8], tgt["note_density"][:, 16:][:, i * 8:(i + 1) * 8]), axis=-1)
        lp = -orl.mse_loss_mean(gen, np.tile(t_i, (n, 1))).reshape(n, B)
        assert np.allclose(lp, totals[:, i]) and np.array_equal(lp.argmax(0), inds[i])
        for b in range(B):
            assert np.array_equal(sample[b, :, i * 128:(i + 1) * 128], cand.reshape(n, B, 4, 256, 16)[inds[i, b], b, :, i * 128:(i + 1) * 128])
    g = load_golden("seg")
    assert g["max_ind"].shape == (2, 2) and g["total_log_prob"].shape == (3, 2, 2)


def test_torch_cpu_restatement_for_the_cpu_baseline_leg():
    """oracle/dit_torch.py (what bench.py's cpu_baseline times): DiTRotary forward at XL width and a teacher-forced DDIM step
    against the reference's goldens -- the same fixtures that pin the numpy oracle."""
    import torch
    from oracle import dit_torch as odt
    g = load_golden("dit_xl_d2")
    sd = odt.to_torch(synth.dit_state_dict(int(g["seed"]), **XL2))
    for H in (128, 64):
        out = odt.dit_forward(sd, torch.from_numpy(g[f"x{H}"]), torch.from_numpy(g[f"t{H}"]), torch.from_numpy(g[f"y{H}"]), depth=2, heads=16)
        assert rel_err(out.numpy(), g[f"out{H}"]) < 1e-4
    s = load_golden("steps")
    sm = odt.to_torch(synth.dit_state_dict(11, **SM))
    S = odf.Schedule(1000, "linear", "ddim50")
    y = torch.from_numpy(s["y"])
    smp, x0 = odt.ddim_step(S, lambda x, t: odt.dit_forward(sm, x, t, y, depth=2, heads=6), torch.from_numpy(s["x"]),
                            torch.from_numpy(s["ddim.t"]), torch.from_numpy(s["ddim.noise"]), eta=1.0)
    assert rel_err(smp.numpy(), s["ddim.sample"]) < 1e-4 and rel_err(x0.numpy(), s["ddim.pred_xstart"]) < 1e-4


def test_every_committed_fixture_carries_the_seeds_the_generator_script_pins():
    """Round-2 review: steps2.npz could not be regenerated from HEAD because a seed counter in make_golden.py had drifted after the
    fixture was written.  The script now pins every stored seed by name (FIXTURE_SEEDS, enforced when a fixture is saved); this test
    reads that table with ast (no reference needed) and checks every committed .npz against it -- in both directions."""
    import ast
    import glob
    import os
    from conftest import GOLDEN
    src = open(os.path.join(GOLDEN, "make_golden.py")).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "FIXTURE_SEEDS")
    table = ast.literal_eval(node.value)
    seen = {}
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        g = np.load(path)
        stored = {k: int(g[k]) for k in g.files if k.endswith("seed") and g[k].ndim == 0}
        if stored:
            seen[os.path.basename(path)[:-4]] = stored
    assert seen == table
    # and the arrays the tests rebuild from those seeds are what the fixture's own outputs were computed from: spot-check one
    g = load_golden("steps2")
    nz = np.random.RandomState(int(g["circ.noise_seed"])).randn(3, 2, 4, 128, 16).astype(np.float32)
    assert nz.shape[1:] == g["circ.sample"].shape and np.isfinite(g["circ.sample"]).all()

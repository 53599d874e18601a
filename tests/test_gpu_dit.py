"""-m gpu: DiTRotary / DiTRotaryClassifier forward on the MI355X against the reference's goldens.

The product modules (guided_diffusion.dit) are driven exactly like the reference's: build from
DiT_models-style constructors, load_state_dict, .to('cuda'), call model(x, t, y).  Tolerance: the
north star's 1e-3 relative on latents; measured fp32 re-association noise is ~1e-5.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _eps_model(arch):
    from guided_diffusion.dit import DiTRotary
    return DiTRotary(input_size=[128, 16], patch_size=arch["patch"], in_channels=arch["in_ch"], hidden_size=arch["hidden"],
                     depth=arch["depth"], num_heads=arch["heads"], num_classes=arch.get("num_classes", 0), learn_sigma=False)


@pytest.mark.parametrize("tag,depth", [("xl_d2", 2), ("xl_d28", 28)])
def test_dit_forward_matches_reference_golden(tag, depth, precision):
    from gpu_util import dev, rel, load_module
    g = load_golden(f"dit_{tag}")
    arch = dict(depth=depth, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = load_module(_eps_model(arch), synth.dit_state_dict(int(g["seed"]), device="cuda", **arch))
    for H in (128, 64):
        out = m(dev(g[f"x{H}"]), dev(g[f"t{H}"]), dev(g[f"y{H}"]))
        assert out.shape == (2, 4, H, 16)
        assert rel(out.cpu().numpy(), g[f"out{H}"]) < TOL, (tag, H)
    # batch invariance + unconditional call path (y=None)
    x, t = dev(g["x128"]), dev(g["t128"])
    a = m(x.repeat(3, 1, 1, 1), t.repeat(3), dev(g["y128"]).repeat(3))
    assert torch.equal(a[:2], a[2:4]) and torch.equal(a[:2], a[4:])        # same tiles, same arithmetic: bitwise
    assert m(x, t).shape == (2, 4, 128, 16)


def test_fresh_module_outputs_zero_like_the_reference():
    """adaLN-zero init (reference dit.py:597-606): a freshly constructed DiTRotary returns exactly 0."""
    from guided_diffusion.dit import DiT_models
    m = DiT_models["DiTRotary_B_8"](input_size=[128, 16], in_channels=4, num_classes=3, learn_sigma=False).to("cuda").eval()
    out = m(torch.randn(2, 4, 128, 16, device="cuda"), torch.tensor([5, 900], device="cuda"), torch.tensor([0, 1], device="cuda"))
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("tag,depth", [("s8", 12), ("s8d2", 2)])
def test_classifier_logits(tag, depth, precision):
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotaryClassifier
    g = load_golden("classifier")
    arch = dict(depth=depth, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    m = DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=depth, num_heads=6, num_classes=16)
    m = load_module(m, synth.dit_state_dict(int(g[f"{tag}.seed"]), **arch))
    out = m(dev(g[f"{tag}.x"]), dev(g[f"{tag}.t"]))
    assert rel(out.cpu().numpy(), g[f"{tag}.logits"]) < TOL


def test_chord_classifier_heads(precision):
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotaryClassifier
    g = load_golden("classifier")
    arch = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=8, chord=True)
    m = DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=8, chord=True)
    m = load_module(m, synth.dit_state_dict(int(g["chord.seed"]), **arch))
    key, ch = m(dev(g["chord.x"]), dev(g["chord.t"]))
    assert rel(key.cpu().numpy(), g["chord.key"]) < TOL
    assert rel(ch.cpu().numpy(), g["chord.logits"]) < TOL


def test_attention_backward_kernel_vs_oracle(precision):
    """d(qkv) of the rotary attention core against the hand-written numpy backward (pinned to autograd goldens): the fp32-MFMA kernels in fp32
    mode, the bf16x3 ones (round 6: every operand split hi + lo as it is fetched, v_mfma_f32_32x32x16_bf16) in the bf16x3 modes."""
    from gpu_util import dev, rel
    from rgm import native as R
    from rgm.synth import rotary_freqs
    from oracle import dit_np as odit
    rng = np.random.RandomState(5)
    for N, T, heads, hd in ((2, 257, 6, 64), (1, 129, 6, 64), (2, 256, 12, 64), (2, 256, 16, 72), (1, 128, 16, 72), (1, 200, 4, 72)):
        D = heads * hd
        qkv = rng.randn(N * T, 3 * D).astype(np.float32)
        d_o = rng.randn(N * T, D).astype(np.float32)
        cos, sin = odit.rotary_tables(rotary_freqs(hd // 2), T)
        r = qkv.reshape(N, T, 3, heads, hd)
        q, k, v = (np.ascontiguousarray(r[:, :, i].transpose(0, 2, 1, 3)) for i in range(3))
        qr, kr = odit.apply_rotary(q, cos, sin), odit.apply_rotary(k, cos, sin)
        s = (qr.astype(np.float64) @ kr.astype(np.float64).transpose(0, 1, 3, 2)) * hd ** -0.5
        p = np.exp(s - s.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        do = d_o.reshape(N, T, heads, hd).transpose(0, 2, 1, 3).astype(np.float64)
        dv = p.transpose(0, 1, 3, 2) @ do
        dp = do @ v.astype(np.float64).transpose(0, 1, 3, 2)
        ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * hd ** -0.5
        dq = odit.apply_rotary((ds @ kr).astype(np.float32), cos, sin, inverse=True)
        dk = odit.apply_rotary((ds.transpose(0, 1, 3, 2) @ qr).astype(np.float32), cos, sin, inverse=True)
        ref = np.stack((dq, dk, dv.astype(np.float32)), axis=0).transpose(1, 3, 0, 2, 4).reshape(N * T, 3 * D)
        qd, gd, cd, sd_ = dev(qkv), dev(d_o), dev(cos), dev(sin)
        od = torch.empty(N * T, D, device="cuda")
        lse = torch.empty(N * heads * T, device="cuda")
        # forward through the product path to get O and lse exactly as the backward will see them
        import ctypes as C
        lib = C.CDLL(R.LIB_PATH)
        ws = torch.empty(1, device="cuda")
        R.check(R.lib.rgm_rotary_attention(R.ptr(qd), R.ptr(od), R.ptr(cd), R.ptr(sd_), N, T, heads, hd, hd // 4, R.current_stream()))
        lse_ref = (np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)).astype(np.float32)    # (N, heads, T)
        lse.copy_(dev(lse_ref.reshape(-1)))
        out = torch.full((N * T, 3 * D), float("nan"), device="cuda")
        R.check(R.lib.rgm_rotary_attention_bwd(R.ptr(qd), R.ptr(od), R.ptr(gd), R.ptr(lse), R.ptr(out), R.ptr(cd), R.ptr(sd_),
                                               N, T, heads, hd, hd // 4, R.current_stream()))
        torch.cuda.synchronize()
        assert rel(out.cpu().numpy(), ref) < (1e-5 if precision == "fp32" else 4e-5), (N, T, heads, hd)


@pytest.mark.parametrize("tag,depth", [("s8d2", 2), ("s8", 12)])
def test_classifier_guidance_gradient_matches_autograd_golden(tag, depth, precision):
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotaryClassifier
    from guided_diffusion.condition_functions import grad_nn_zt_mse
    g = load_golden("classifier")
    arch = dict(depth=depth, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    m = DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=depth, num_heads=6, num_classes=16)
    m = load_module(m, synth.dit_state_dict(int(g[f"{tag}.seed"]), **arch))
    x, t, rule = dev(g[f"{tag}.x"]), dev(g[f"{tag}.t"]), dev(g[f"{tag}.rule"])
    logits, grad = m.value_and_grad(x, t, rule, "mse", 10.0)
    assert rel(logits.cpu().numpy(), g[f"{tag}.logits"]) < TOL
    assert rel(grad.cpu().numpy(), g[f"{tag}.grad"]) < 5e-4
    assert torch.equal(grad_nn_zt_mse(x, t, rule=rule, classifier_scale=10., classifier=m), grad)
    # plain forward still agrees with the saved-activation forward
    # the plain forward and the saved-activation forward run the same kernels except, in presplit mode, the GELU (fused
    # exp2/rcp form in the GEMM epilogue vs libm tanh in the stand-alone activation pass that keeps the pre-activation)
    assert rel(m(x, t).cpu().numpy(), logits.cpu().numpy()) < (1e-5 if precision == "bf16x3_presplit" else 1e-6)


def test_chord_classifier_guidance_gradient(precision):
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotaryClassifier
    from guided_diffusion.condition_functions import grad_nn_zt_chord
    g = load_golden("classifier")
    arch = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=8, chord=True)
    m = DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=8, chord=True)
    m = load_module(m, synth.dit_state_dict(int(g["chord.seed"]), **arch))
    grad = grad_nn_zt_chord(dev(g["chord.x"]), dev(g["chord.t"]), rule=dev(g["chord.rule"]), classifier_scale=10., classifier=m)
    assert rel(grad.cpu().numpy(), g["chord.grad"]) < 5e-4


def test_classifier_guided_p_sample_step_matches_reference(precision):
    """BASELINE config 3 shape in miniature: p_sample on the '250' chain with composite_nn_zt guidance."""
    from functools import partial
    from types import SimpleNamespace
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotary, DiTRotaryClassifier
    from guided_diffusion.condition_functions import model_fn, composite_nn_zt
    from guided_diffusion.script_util import create_diffusion
    g = load_golden("steps")
    sm = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6,
                              num_classes=3, learn_sigma=False), synth.dit_state_dict(11, **sm))
    carch = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    cm = load_module(DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6,
                                         num_classes=16), synth.dit_state_dict(4, **carch))
    d = create_diffusion(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="250", use_kl=False,
                         predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)
    d.t_end = 0
    nz = torch.from_numpy(g["cg.noise"])
    d.noise_fn = lambda shape, device: nz.to(device)
    cond = partial(composite_nn_zt, fns=["grad_nn_zt_mse"], classifier_scales=[10.], classifiers=[cm], rule_names=["note_density"])
    out = d.p_sample(partial(model_fn, model=m, num_classes=3, class_cond=True, cfg=False, w=0.), dev(g["x"]), dev(g["cg.t"]),
                     clip_denoised=False, cond_fn=cond, model_kwargs={"y": dev(g["y"]), "rule": {"note_density": dev(g["cg.rule"])}},
                     guidance_kwargs=SimpleNamespace(schedule=False, method="classifier_guidance"))
    assert rel(out["sample"].cpu().numpy(), g["cg.sample"]) < 2e-4


def test_cfg_model_fn_is_one_batched_forward_with_the_same_result(precision):
    """Classifier-free guidance (condition_functions.py:22-23): the 2B-row batched forward equals the two B-row passes."""
    from functools import partial
    from gpu_util import dev, load_module, rel
    from guided_diffusion.condition_functions import dc_model_fn, model_fn
    from guided_diffusion.dit import DiTRotary
    arch = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6,
                              num_classes=3, learn_sigma=False), synth.dit_state_dict(11, **arch))
    calls = []

    def counted(x, t, y):
        calls.append(x.shape[0])
        return m(x, t, y)
    rng = np.random.RandomState(5)
    x, t, y = dev(rng.randn(3, 4, 128, 16).astype(np.float32)), dev(np.array([10, 500, 999])), dev(np.array([0, 2, 1]))
    null = torch.full((3,), 3, dtype=torch.int64, device="cuda")
    ref = (1 + 4.0) * m(x, t, y) - 4.0 * m(x, t, null)
    out = model_fn(x, t, y, model=counted, num_classes=3, class_cond=True, cfg=True, w=4.0)
    assert calls == [6]
    assert rel(out.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    out_dc = dc_model_fn(x.permute(0, 1, 3, 2), t, y, model=counted, num_classes=3, class_cond=True, cfg=True, w=4.0)
    assert rel(out_dc.permute(0, 1, 3, 2).cpu().numpy(), ref.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("tag,arch,seed", [("sm", dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, out_ch=4, num_classes=3), 11),
                                           ("xl2", dict(depth=2, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3), 1)])
def test_eps_network_input_gradient_matches_autograd_golden(tag, arch, seed, precision):
    """DPS: (d eps / d x)^T g through the saved-activation forward + dgrad chain vs the reference's autograd,
    at the classifier width (head_dim 64) and at XL width (head_dim 72)."""
    from gpu_util import dev, load_module, rel
    from guided_diffusion.dit import DiTRotary
    g = load_golden("dps")
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=arch["hidden"], depth=arch["depth"],
                              num_heads=arch["heads"], num_classes=3, learn_sigma=False), synth.dit_state_dict(seed, **arch))
    x, t, y, gg = dev(g[f"{tag}.x"]), dev(g[f"{tag}.t"]), dev(g[f"{tag}.y"]), dev(g[f"{tag}.g"])
    eps, grad = m.vjp(x, t, y, gg)
    assert rel(eps.cpu().numpy(), g[f"{tag}.eps"]) < 2e-4
    assert rel(grad.cpu().numpy(), g[f"{tag}.grad"]) < 5e-4
    assert rel(eps.cpu().numpy(), m(x, t, y).cpu().numpy()) < (1e-6 if precision == "fp32" else 3e-5)
    # the two phases may be separate calls (DPS forms g from eps in between); the gradient is linear in g
    m.vjp_forward(x, t, y)
    g2 = m.vjp_backward(2.0 * gg)
    assert rel(g2.cpu().numpy(), 2.0 * grad.cpu().numpy()) < 1e-6

"""eps-network adapters and classifier-gradient cond_fns -- reference API
(guided_diffusion/condition_functions.py:17-42, :58-85, :149-174).

model_fn / dc_model_fn are thin: class-conditional call, null label (= num_classes) when unconditional,
classifier-free guidance as one 2B-row forward.  The grad_nn_zt_* guidance functions call the classifier's
fused value-and-input-gradient kernel chain (no autograd graph; weights are frozen at sampling time).
DPS variants: nn_z0_* (value + input gradient from the same fused chain) and rule_x0_* on the decoded roll (value + roll
gradient; the sampler pulls it back through the VAE decoder and the eps-network).
"""
import torch as th
import torch.nn as nn

from rgm import native as _rgm


def _null_labels(x, num_classes):
    return th.full((x.shape[0],), num_classes, dtype=th.int64, device=x.device)


def _cfg_eps(model, x, t, y, num_classes, w):
    """Classifier-free guidance (reference :22-23) as ONE forward over the conditional and the null-label copies of the
    batch (2B rows fill the GPU better than two B-row passes; per-row results are identical) + one combine."""
    B = x.shape[0]
    e = model(th.cat([x, x], dim=0), th.cat([t, t], dim=0), th.cat([y.to(th.int64), _null_labels(x, num_classes)], dim=0))
    return th.add(e[:B], e[:B] - e[B:], alpha=w)            # (1 + w) * e_cond - w * e_null


def model_fn(x, t, y=None, rule=None, model=nn.Identity(), num_classes=3, class_cond=True, cfg=False, w=0.):
    """`rule` is a dummy argument (model_kwargs carries it for the cond_fn / SCG)."""
    if not class_cond:
        return model(x, t, _null_labels(x, num_classes))
    if cfg:
        return _cfg_eps(model, x, t, y, num_classes, w)
    return model(x, t, y)


def dc_model_fn(x, t, y=None, rule=None, model=nn.Identity(), num_classes=3, class_cond=True, cfg=False, w=0.):
    """DiffCollage eps functions work on (4, pitch, time); the sampler's latents are (4, time, pitch)."""
    xt = x.permute(0, 1, 3, 2)
    if not class_cond:
        return model(xt, t, _null_labels(xt, num_classes)).permute(0, 1, 3, 2)
    if cfg:
        return _cfg_eps(model, xt, t, y, num_classes, w).permute(0, 1, 3, 2)
    return model(xt, t, y).permute(0, 1, 3, 2)


def _value_and_grad(classifier, x, t, target, loss_kind, scale):
    if not hasattr(classifier, "value_and_grad"):
        raise NotImplementedError("classifier guidance needs a native DiTRotaryClassifier (value_and_grad)")
    return classifier.value_and_grad(x, t, target, loss_kind, scale)[1]


def grad_nn_zt_xentropy(x, y=None, rule=None, classifier=nn.Identity()):
    """grad_x of log softmax(classifier(x, t = 0))[rule] (reference :46-56; its signature has no `t` / `classifier_scale`, so --
    as in the reference -- it cannot be reached through composite_nn_zt; kept for callers that bind it directly)."""
    assert rule is not None
    return _value_and_grad(classifier, x, _zeros_t(x), rule, "xent", 1.0)


def grad_nn_zt_mse(x, t, y=None, rule=None, classifier_scale=10., classifier=nn.Identity()):
    """grad_x of -sum((classifier(x,t) - rule)^2), times classifier_scale."""
    assert rule is not None
    return _value_and_grad(classifier, x, t, rule, "mse", classifier_scale)


def grad_nn_zt_chord(x, t, y=None, rule=None, classifier_scale=10., classifier=nn.Identity(), both=False):
    """grad_x of -sum CE(chord_logits, rule), times classifier_scale."""
    assert rule is not None
    if both:
        raise NotImplementedError("both=True (key + chord) is unused by the shipped configs")
    return _value_and_grad(classifier, x, t, rule, "chord_ce", classifier_scale)


# ---- DPS log-probabilities on the x0 estimate (reference :88-138).  Each has a `_vag` twin returning (log_probs, d sum(log_probs)/dx)
# from the classifier's fused value-and-input-gradient chain: the sampler's DPS branch needs both (condition_mean :415-465).
def _zeros_t(x):
    return th.zeros(x.shape[0], dtype=th.int64, device=x.device)


def _mse_logp(logits, rule):
    return -((logits - rule.to(logits.dtype)) ** 2).sum(dim=-1)


def nn_z0_mse_dummy(x, t, y=None, rule=None, classifier_scale=0.1, classifier=nn.Identity()):
    """-MSE(classifier(x0, 0), rule) * classifier_scale; t is a dummy (the classifier sees a clean latent, t = 0)."""
    assert rule is not None
    return _mse_logp(classifier(x, _zeros_t(x)), rule) * classifier_scale


def nn_z0_mse(x, rule=None, classifier=nn.Identity()):
    return _mse_logp(classifier(x, _zeros_t(x)), rule)


def _chord_logp(chord_logits, rule, B):
    lp = -th.nn.functional.cross_entropy(chord_logits.reshape(-1, chord_logits.shape[-1]), rule.reshape(-1).long(), reduction="none")
    return lp.reshape(B, -1).mean(dim=-1)


def nn_z0_chord_dummy(x, t, y=None, rule=None, classifier_scale=0.1, classifier=nn.Identity(), both=False):
    if both:
        raise NotImplementedError("both=True (key + chord) is unused by the shipped configs")
    _, chord_logits = classifier(x, _zeros_t(x))
    return _chord_logp(chord_logits, rule, x.shape[0]) * classifier_scale


def _nn_z0_mse_dummy_vag(x, rule, classifier_scale, classifier):
    logits, grad = classifier.value_and_grad(x, _zeros_t(x), rule, "mse", classifier_scale)
    return _mse_logp(logits, rule) * classifier_scale, grad


def _nn_z0_chord_dummy_vag(x, rule, classifier_scale, classifier):
    nw = rule.reshape(x.shape[0], -1).shape[1]                         # mean over the chord windows
    logits, grad = classifier.value_and_grad(x, _zeros_t(x), rule, "chord_ce", classifier_scale / nw)
    return _chord_logp(logits, rule, x.shape[0]) * classifier_scale, grad


# ---- DPS through a rule on the DECODED roll (reference :122-138; guidance.nn False, guidance.vae True).  x is the decoded
# piano roll (N,3,128,T); the `_vag` twin also returns d log p / d roll, which the sampler pulls back through the VAE
# decoder (rgm_vae_decode_latent_vjp) and the eps-network (rgm_dit_vjp) -- the reference leaves that to autograd.
def rule_x0_mse_dummy(x, t, y=None, rule=None, rule_name='pitch_hist'):
    from music_rule_guidance.rule_maps import FUNC_DICT
    return _mse_logp(FUNC_DICT[rule_name](x), rule.to(x.device))


def rule_x0_mse(x, rule=None, rule_name='pitch_hist', soft=False):
    if soft:
        raise NotImplementedError("soft rules: the reference's rule programs take no `soft` argument either")
    from music_rule_guidance.rule_maps import FUNC_DICT
    return _mse_logp(FUNC_DICT[rule_name](x), rule.to(x.device))


def _rule_x0_vag(roll, rule, name, scale):
    """(scale * log p (N,), d (scale * sum log p) / d roll or None when it is identically zero) for one rule program."""
    import functools
    from music_rule_guidance import music_rules, rule_maps
    fn = rule_maps.FUNC_DICT[name]
    base = fn.func if isinstance(fn, functools.partial) else fn
    N, Cc, _, T = roll.shape
    if fn is music_rules.total_pitch_class_histogram:
        _rgm.require_cuda(roll)
        assert roll.is_contiguous() and roll.dtype == th.float32
        tgt = rule.to(roll.device, th.float32).reshape(N, 12).contiguous()
        logp = th.empty((N,), dtype=th.float32, device=roll.device)
        d_roll = th.empty_like(roll)
        scratch = th.empty((N, 140), dtype=th.float32, device=roll.device)
        with th.cuda.device(roll.device):
            _rgm.check(_rgm.lib.rgm_rule_pitch_hist_vag(_rgm.ptr(roll), _rgm.ptr(tgt), float(scale), None, _rgm.ptr(logp),
                                                        _rgm.ptr(d_roll), _rgm.ptr(scratch), N, Cc, T, _rgm.current_stream()))
        return logp, d_roll
    if base in (music_rules.note_density, music_rules.note_density_class, music_rules.get_chords):
        # counting behind hard thresholds: the reference's autograd returns an all-zero gradient for these too
        # (every element of the roll is overwritten by a constant before it is summed, music_rules.py:66-70)
        return _mse_logp(fn(roll).reshape(N, -1).float(), rule.to(roll.device).reshape(N, -1)) * scale, None
    with th.enable_grad():                                   # a user-registered torch rule: autograd on the roll only
        r = roll.detach().requires_grad_(True)
        lp = _mse_logp(fn(r).reshape(N, -1), rule.to(roll.device).reshape(N, -1)) * scale
        g = th.autograd.grad(lp.sum(), r)[0]
    return lp.detach(), g


def composite_rule_value_and_grad(roll, t, y=None, rule=None, fns=None, classifier_scales=None, rule_names=None):
    """(log_probs (N,), d sum(log_probs) / d roll) of composite_rule on a decoded roll."""
    lp, grad = 0, None
    for fn, scale, name in zip(fns, classifier_scales, rule_names):
        if fn not in ("rule_x0_mse_dummy", "rule_x0_mse"):
            raise NotImplementedError(f"DPS rule guidance with cond_fn '{fn}'")
        a, b = _rule_x0_vag(roll, rule[name], name, scale)
        lp = lp + a
        if b is not None:
            grad = b if grad is None else grad + b
    return lp, grad


function_map = {
    "grad_nn_zt_xentropy": grad_nn_zt_xentropy,
    "grad_nn_zt_mse": grad_nn_zt_mse,
    "grad_nn_zt_chord": grad_nn_zt_chord,
    "nn_z0_chord_dummy": nn_z0_chord_dummy, "nn_z0_mse_dummy": nn_z0_mse_dummy, "nn_z0_mse": nn_z0_mse,
    "rule_x0_mse_dummy": rule_x0_mse_dummy, "rule_x0_mse": rule_x0_mse,
}
_vag_map = {"nn_z0_mse_dummy": _nn_z0_mse_dummy_vag, "nn_z0_chord_dummy": _nn_z0_chord_dummy_vag}


def composite_nn_zt_value_and_grad(x, t, y=None, rule=None, fns=None, classifier_scales=None, classifiers=None, rule_names=None):
    """(log_probs (B,), d sum(log_probs) / dx) of composite_nn_zt for the DPS cond_fns -- what autograd provides in the reference."""
    lp, grad = 0, 0
    for fn, scale, cls, name in zip(fns, classifier_scales, classifiers, rule_names):
        if fn not in _vag_map:
            raise NotImplementedError(f"DPS guidance with cond_fn '{fn}'")
        a, b = _vag_map[fn](x, rule[name], scale, cls)
        lp, grad = lp + a, grad + b
    return lp, grad


_SIDE_STREAMS = {}     # device -> side streams for classifiers 1.. of a composite cond_fn (composite_nn_zt)
CONCURRENT_CLASSIFIERS = __import__("os").environ.get("RGM_CLS_STREAMS", "1") != "0"


def composite_nn_zt(x, t, y=None, rule=None, fns=None, classifier_scales=None, classifiers=None, rule_names=None):
    """sum of the classifiers' guidance gradients (reference condition_functions.py:161-167).  The classifiers are independent chains
    of small launches (DiTRotary-S/8-cls at the sampler's batch: a few dozen workgroups per kernel on a 256-CU chip), each with its own
    native handle and workspace: classifier i > 0 runs on a side stream forked from and joined to the caller's stream, so the chains
    interleave on the device; the sum is formed on the caller's stream in the reference's order (RGM_CLS_STREAMS=0: one after the other)."""
    items = list(zip(fns, classifier_scales, classifiers, rule_names))
    if not (CONCURRENT_CLASSIFIERS and len(items) > 1 and th.is_tensor(x) and x.is_cuda):
        out = 0
        for fn, scale, cls, name in items:
            out = out + function_map[fn](x, t, y=y, rule=rule[name], classifier_scale=scale, classifier=cls)
        return out
    main = th.cuda.current_stream(x.device)
    pool = _SIDE_STREAMS.setdefault(x.device, [])
    while len(pool) < len(items) - 1:
        pool.append(th.cuda.Stream(device=x.device))
    parts = []
    for i, (fn, scale, cls, name) in enumerate(items):
        if i == 0:
            continue
        side = pool[i - 1]
        side.wait_stream(main)
        with th.cuda.stream(side):
            g = function_map[fn](x, t, y=y, rule=rule[name], classifier_scale=scale, classifier=cls)
        parts.append((i, g, side))
    fn, scale, cls, name = items[0]
    out = function_map[fn](x, t, y=y, rule=rule[name], classifier_scale=scale, classifier=cls)
    for i, g, side in parts:
        main.wait_stream(side)
        if th.is_tensor(g):
            g.record_stream(main)
        out = out + g
    return out


def composite_rule(x, t, y=None, rule=None, fns=None, classifier_scales=None, rule_names=None):
    out = 0
    for fn, scale, name in zip(fns, classifier_scales, rule_names):
        out = out + function_map[fn](x, t, y=y, rule=rule[name], rule_name=name) * scale
    return out

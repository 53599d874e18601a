"""eps-network adapters and classifier-gradient cond_fns -- reference API
(guided_diffusion/condition_functions.py:17-42, :58-85, :149-174).

model_fn / dc_model_fn are thin: class-conditional call, null label (= num_classes) when unconditional,
classifier-free guidance as one 2B-row forward.  The grad_nn_zt_* guidance functions call the classifier's
fused value-and-input-gradient kernel chain (no autograd graph; weights are frozen at sampling time).
DPS variants nn_z0_* are provided (value + input gradient from the same fused chain); rule_x0_* (through the VAE decoder)
are a 'next' row.
"""
import torch as th
import torch.nn as nn


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


def _dps_rule(*a, **k):
    raise NotImplementedError("DPS through rule(decode(x0)) (rule_x0_*) needs the VAE decoder's backward: 'next' row SURVEY 8f.1")


function_map = {
    "grad_nn_zt_mse": grad_nn_zt_mse,
    "grad_nn_zt_chord": grad_nn_zt_chord,
    "nn_z0_chord_dummy": nn_z0_chord_dummy, "nn_z0_mse_dummy": nn_z0_mse_dummy, "nn_z0_mse": nn_z0_mse,
    "rule_x0_mse_dummy": _dps_rule, "rule_x0_mse": _dps_rule,
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


def composite_nn_zt(x, t, y=None, rule=None, fns=None, classifier_scales=None, classifiers=None, rule_names=None):
    out = 0
    for fn, scale, cls, name in zip(fns, classifier_scales, classifiers, rule_names):
        out = out + function_map[fn](x, t, y=y, rule=rule[name], classifier_scale=scale, classifier=cls)
    return out


def composite_rule(x, t, y=None, rule=None, fns=None, classifier_scales=None, rule_names=None):
    out = 0
    for fn, scale, name in zip(fns, classifier_scales, rule_names):
        out = out + function_map[fn](x, t, y=y, rule=rule[name], rule_name=name) * scale
    return out

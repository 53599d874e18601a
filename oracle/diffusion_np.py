"""Scheduler oracle: beta schedules, timestep respacing, one DDPM / DDIM / SCG step (numpy).

Test infrastructure (see oracle/__init__.py).  Restates, on CPU in numpy:
  guided_diffusion/gaussian_diffusion.py:31-62   get_named_beta_schedule
  guided_diffusion/gaussian_diffusion.py:138-189 GaussianDiffusion.__init__ tables (float64)
  guided_diffusion/respace.py:7-60,72-86,116-128 space_timesteps / SpacedDiffusion / t remap
  guided_diffusion/gaussian_diffusion.py:252-364 p_mean_variance (EPSILON, FIXED_LARGE)
  guided_diffusion/gaussian_diffusion.py:387-407 condition_mean (classifier guidance branch)
  guided_diffusion/gaussian_diffusion.py:467-489 condition_score
  guided_diffusion/gaussian_diffusion.py:491-554 scg_sample (argmax branch)
  guided_diffusion/gaussian_diffusion.py:562-592 scg_sample (dc.base > 0: one winner per time segment)
  guided_diffusion/condition_functions.py:17-42  model_fn / dc_model_fn (null label, classifier-free guidance)
  guided_diffusion/gaussian_diffusion.py:635-735 p_sample
  guided_diffusion/gaussian_diffusion.py:881-976 ddim_sample
  guided_diffusion/gaussian_diffusion.py:1347-1358 _decode, :1398-1400 guide_schedule

Noise is always INJECTED by the caller in the reference's draw order (SURVEY 7 "RNG parity").
Arithmetic follows the reference: tables in float64, cast to float32 at lookup
(_extract_into_tensor :1331-1344), elementwise work in float32.
"""
import math
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- schedules
def named_beta_schedule(name, T):
    """gaussian_diffusion.py:31-62."""
    if name == "linear":
        s = 1000 / T
        return np.linspace(s * 0.0001, s * 0.02, T, dtype=np.float64)
    if name == "cosine":
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - f((i + 1) / T) / f(i / T), 0.999) for i in range(T)])
    if name == "stable-diffusion":
        s = 1000 / T
        return np.linspace(s * math.sqrt(0.00085), s * math.sqrt(0.012), T, dtype=np.float64) ** 2
    raise NotImplementedError(name)


def space_timesteps(T, section_counts):
    """respace.py:7-60 ("ddimN" fixed stride, or comma-separated per-section counts)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, T):
                if len(range(0, T, stride)) == want:
                    return set(range(0, T, stride))
            raise ValueError("no integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(T, len(section_counts))
    start, out = 0, []
    for i, cnt in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError("section too small")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            out.append(start + round(cur))
            cur += stride
        start += size
    return set(out)


class Schedule:
    """Tables of a (re-spaced) Gaussian diffusion; EPSILON mean, FIXED_LARGE variance."""

    def __init__(self, steps=1000, noise_schedule="linear", timestep_respacing=""):
        base = named_beta_schedule(noise_schedule, steps)
        use = space_timesteps(steps, timestep_respacing if timestep_respacing else [steps])
        ac = np.cumprod(1.0 - base)
        last, betas, self.timestep_map = 1.0, [], []
        for i, a in enumerate(ac):                       # respace.py:77-84
            if i in use:
                betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = len(betas)
        al = 1.0 - betas
        self.alphas_cumprod = np.cumprod(al)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1 - self.alphas_cumprod_prev) / (1 - self.alphas_cumprod)
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1 - self.alphas_cumprod_prev) * np.sqrt(al) / (1 - self.alphas_cumprod)
        # FIXED_LARGE (:316-329)
        self.model_variance = np.append(self.posterior_variance[1], betas[1:])
        self.model_log_variance = np.log(self.model_variance)

    def ex(self, arr, t, ndim=4):
        """_extract_into_tensor: float64 table -> float32 per-sample column."""
        return arr[np.asarray(t)].astype(F32).reshape((-1,) + (1,) * (ndim - 1))

    def map_t(self, t):
        return np.asarray(self.timestep_map, dtype=np.int64)[np.asarray(t)]


def guide_schedule(t0, t_start=750, t_end=0, interval=1):
    """:1398-1400."""
    return bool(t_start > t0 >= t_end and (t0 + 1) % interval == 0)


# ----------------------------------------------------------------------------- step math
def xstart_from_eps(S, x, t, eps):
    return S.ex(S.sqrt_recip_alphas_cumprod, t) * x - S.ex(S.sqrt_recipm1_alphas_cumprod, t) * eps


def eps_from_xstart(S, x, t, x0):
    return (S.ex(S.sqrt_recip_alphas_cumprod, t) * x - x0) / S.ex(S.sqrt_recipm1_alphas_cumprod, t)


def posterior_mean(S, x0, x, t):
    return S.ex(S.posterior_mean_coef1, t) * x0 + S.ex(S.posterior_mean_coef2, t) * x


def p_mean_variance(S, model, x, t, clip_denoised, model_kwargs, edit=None):
    """:252-357; `model(x, t_mapped, **kw)` is any callable returning eps (float32).

    edit = dict(gt, mask, l_start, l_end) is the replacement-based conditioning of scripts/edit.py (:293-298)."""
    eps = model(x, S.map_t(t), **model_kwargs).astype(F32)
    if edit is not None:
        x0e = xstart_from_eps(S, x, t, eps)
        if clip_denoised:
            x0e = np.clip(x0e, -1, 1)
        x0e = edit["mask"] * edit["gt"] + (1 - edit["mask"]) * x0e
        eps = eps_from_xstart(S, x, t, x0e).astype(F32)
    x0 = xstart_from_eps(S, x, t, eps)
    if clip_denoised:
        x0 = np.clip(x0, -1, 1)
    return {
        "mean": posterior_mean(S, x0, x, t),
        "variance": np.broadcast_to(S.ex(S.model_variance, t), x.shape),
        "log_variance": np.broadcast_to(S.ex(S.model_log_variance, t), x.shape),
        "pred_xstart": x0,
    }


def decode_latent(z, decode_fn, scale_factor):
    """_decode :1347-1358: (N,4,H,16) latent -> (N,3,128,8H) roll through 16x16 squares."""
    H, W = z.shape[-2:]
    z = (z / F32(scale_factor)).transpose(0, 1, 3, 2)
    n_seg = H // W
    tiles = np.concatenate(np.split(z, n_seg, axis=-1), axis=0)     # segment-major, batch-minor
    dec = decode_fn(np.ascontiguousarray(tiles))
    return np.concatenate(np.split(dec, n_seg, axis=0), axis=-1)


def model_fn(model, x, t, y=None, num_classes=3, class_cond=True, cfg=False, w=0., transpose=False):
    """condition_functions.py:17-27 (transpose=False) / :30-42 dc_model_fn (transpose=True: the DiffCollage eps function
    works on (4, pitch, time), the sampler's latent is (4, time, pitch)).  `model(x, t, y)` is any eps callable."""
    if transpose:
        x = np.ascontiguousarray(x.transpose(0, 1, 3, 2))
    y_null = np.full((x.shape[0],), num_classes, dtype=np.int64)
    if class_cond:
        if cfg:
            out = F32(1 + w) * model(x, t, y).astype(F32) - F32(w) * model(x, t, y_null).astype(F32)
        else:
            out = model(x, t, y)
    else:
        out = model(x, t, y_null)
    out = out.astype(F32)
    return np.ascontiguousarray(out.transpose(0, 1, 3, 2)) if transpose else out


def _scg_segments(cand, x0, B, n, model_kwargs, scg_kwargs, func_dict, loss_dict, base):
    """:562-592.  One argmax per time segment of `base` latent rows (8*base roll frames); note_density / chord targets are
    cut to the segment's windows (rule_base = base // 16 windows per segment), pitch_hist targets are used whole."""
    total_len = x0.shape[-1]
    seg = base * 8
    rule_base = base // 16
    cv = cand.reshape((n, B) + cand.shape[1:])
    pieces, inds, totals = [], [], []
    for i, s0 in enumerate(range(0, total_len, seg)):
        s1 = min(s0 + seg, total_len)
        cur = np.ascontiguousarray(x0[:, :, :, s0:s1])
        total = np.zeros(n * B, dtype=F32)
        for name, target in model_kwargs["rule"].items():
            gen = func_dict[name](cur)
            if name == "note_density":
                half = target.shape[-1] // 2
                sl = slice(i * rule_base, min((i + 1) * rule_base, half))
                target = np.concatenate((target[:, :half][:, sl], target[:, half:][:, sl]), axis=-1)
            elif "chord" in name:
                target = target[:, i * rule_base: min((i + 1) * rule_base, target.shape[-1])]
            lp = -loss_dict[name](gen, np.tile(target, (n, 1)))
            total = (total + lp * F32(scg_kwargs.get(name, 1.0))).astype(F32)
        total = total.reshape(n, B)
        mi = total.argmax(axis=0)
        pieces.append(cv[mi, np.arange(B)][:, :, s0 // 8: s1 // 8])
        inds.append(mi)
        totals.append(total)
    return np.concatenate(pieces, axis=-2), np.stack(inds), np.stack(totals, axis=1)


def scg_sample(S, model, t, mean_pred, g_coeff, decode_fn, scale_factor, model_kwargs, scg_kwargs,
               noise, func_dict, loss_dict, wrap_t=True, return_aux=False, edit=None, dc_base=0):
    """:491-554 (dc.base<=0 branch) and :562-592 (dc_base > 0).  `noise` has shape (n,B,C,H,W) -- the randn_like draw of :512.

    wrap_t=False reproduces the p_sample quirk (model passed unwrapped).
    """
    n = scg_kwargs["num_samples"]
    B = mean_pred.shape[0]
    cand = (np.broadcast_to(mean_pred[None], (n,) + mean_pred.shape) + g_coeff * noise).astype(F32)
    cand = cand.reshape((n * B,) + mean_pred.shape[1:])
    t_rep = np.tile(np.asarray(t), n)
    y_rep = np.tile(np.asarray(model_kwargs["y"]), n)
    eps = model(cand, S.map_t(t_rep) if wrap_t else t_rep, y=y_rep).astype(F32)
    x0 = xstart_from_eps(S, cand, t_rep, eps)
    if edit is not None:                                             # :520-522 only the editable rows are decoded
        x0 = np.ascontiguousarray(x0[:, :, edit["l_start"]:edit["l_end"], :])
    if decode_fn is not None:
        x0 = decode_latent(x0, decode_fn, scale_factor)
    if dc_base > 0:
        sample, inds, totals = _scg_segments(cand, x0, B, n, model_kwargs, scg_kwargs, func_dict, loss_dict, dc_base)
        if return_aux:
            return sample, {"total_log_prob": totals, "max_ind": inds, "pred_xstart_dec": x0}
        return sample
    total = np.zeros(n * B, dtype=F32)
    each = {}
    for name, target in model_kwargs["rule"].items():
        gen = func_dict[name](x0)
        tgt = np.tile(target, (n, 1))
        lp = -loss_dict[name](gen, tgt)
        each[name] = lp
        total = (total + lp * F32(scg_kwargs.get(name, 1.0))).astype(F32)
    total = total.reshape(n, B)
    max_ind = total.argmax(axis=0)                                   # first max wins
    sample = cand.reshape((n, B) + mean_pred.shape[1:])[max_ind, np.arange(B)]
    if return_aux:
        return sample, {"total_log_prob": total, "max_ind": max_ind, "pred_xstart_dec": x0, "each": each}
    return sample


def p_sample(S, model, x, t, noise, clip_denoised=False, cond_fn=None, model_kwargs=None,
             guidance=None, scg_kwargs=None, decode_fn=None, scale_factor=1.0, t_end=0,
             func_dict=None, loss_dict=None, return_aux=False, edit=None, dc_base=0):
    """:635-735.  `guidance` = dict(schedule,t_start,t_end,interval) or None.

    noise: (B,...) for the plain / unguided-SCG draw, (n,B,...) for the SCG draw, None when
    the reference draws nothing (SCG and t[0]==t_end).
    """
    model_kwargs = model_kwargs or {}
    t = np.asarray(t)
    if guidance is not None:
        use_g = guide_schedule(int(t[0]), guidance["t_start"], guidance["t_end"], guidance["interval"]) \
            if guidance["schedule"] else True
    else:
        use_g = False
    out = p_mean_variance(S, model, x, t, clip_denoised, model_kwargs, edit=edit)
    aux = {"mean_unguided": out["mean"]}
    if cond_fn is not None and (use_g or scg_kwargs is not None):
        if edit is None:
            grad = cond_fn(x, S.map_t(t), **model_kwargs).astype(F32)   # :404 (cond_fn is wrapped)
        else:                                                        # :408-414 gradient on the editable rows only
            ls, le = edit["l_start"], edit["l_end"]
            grad = np.zeros_like(x)
            grad[:, :, ls:le, :] = cond_fn(np.ascontiguousarray(x[:, :, ls:le, :]), S.map_t(t), **model_kwargs).astype(F32)
        out["mean"] = (out["mean"] + out["variance"] * grad).astype(F32)
        aux["grad"] = grad
    g = np.exp(F32(0.5) * out["log_variance"]).astype(F32)
    if scg_kwargs is None:
        mask = (t > t_end).astype(F32).reshape((-1,) + (1,) * (x.ndim - 1))
        sample = out["mean"] + mask * g * noise
    elif int(t[0]) > t_end:
        if use_g:
            sample = scg_sample(S, model, t, out["mean"], g, decode_fn, scale_factor, model_kwargs,
                                scg_kwargs, noise, func_dict, loss_dict, wrap_t=False,
                                return_aux=return_aux, edit=edit, dc_base=dc_base)
            if return_aux:
                sample, a2 = sample
                aux.update(a2)
        else:
            sample = out["mean"] + g * noise
    else:
        sample = out["mean"]
    res = {"sample": sample.astype(F32), "pred_xstart": out["pred_xstart"], "mean": out["mean"]}
    if return_aux:
        res["aux"] = aux
    return res


def ddim_sample(S, model, x, t, noise, eta=0.0, clip_denoised=False, cond_fn=None, model_kwargs=None,
                guidance=None, scg_kwargs=None, decode_fn=None, scale_factor=1.0, t_end=0,
                func_dict=None, loss_dict=None, return_aux=False, edit=None, dc_base=0):
    """:881-976."""
    model_kwargs = model_kwargs or {}
    t = np.asarray(t)
    if guidance is not None:
        use_g = guide_schedule(int(t[0]), guidance["t_start"], guidance["t_end"], guidance["interval"]) \
            if guidance["schedule"] else True
    else:
        use_g = False
    out = p_mean_variance(S, model, x, t, clip_denoised, model_kwargs, edit=edit)
    if cond_fn is not None and use_g:                                # condition_score :467-489
        ab = S.ex(S.alphas_cumprod, t)
        eps = eps_from_xstart(S, x, t, out["pred_xstart"])
        eps = eps - np.sqrt(1 - ab) * cond_fn(x, S.map_t(t), **model_kwargs).astype(F32)
        out["pred_xstart"] = xstart_from_eps(S, x, t, eps)
        out["mean"] = posterior_mean(S, out["pred_xstart"], x, t)
    eps = eps_from_xstart(S, x, t, out["pred_xstart"])
    ab = S.ex(S.alphas_cumprod, t)
    abp = S.ex(S.alphas_cumprod_prev, t)
    sigma = F32(eta) * np.sqrt((1 - abp) / (1 - ab)) * np.sqrt(1 - ab / abp)
    mean_pred = out["pred_xstart"] * np.sqrt(abp) + np.sqrt(1 - abp - sigma ** 2) * eps
    aux = {"mean_pred": mean_pred, "sigma": sigma}
    if scg_kwargs is None:
        mask = (t != t_end).astype(F32).reshape((-1,) + (1,) * (x.ndim - 1))
        sample = mean_pred + mask * sigma * noise
    elif int(t[0]) > t_end:
        if use_g:
            sample = scg_sample(S, model, t, mean_pred, sigma, decode_fn, scale_factor, model_kwargs,
                                scg_kwargs, noise, func_dict, loss_dict, wrap_t=True,
                                return_aux=return_aux, edit=edit, dc_base=dc_base)
            if return_aux:
                sample, a2 = sample
                aux.update(a2)
        else:
            sample = mean_pred + sigma * noise
    else:
        sample = mean_pred
    res = {"sample": sample.astype(F32), "pred_xstart": out["pred_xstart"]}
    if return_aux:
        res["aux"] = aux
    return res


def sample_loop(S, model, x_T, noises, ddim=False, **kw):
    """p_sample_loop_progressive :809-879 / ddim_sample_loop_progressive :1073-1143.

    `noises[i]` is the injected draw for loop position i (index T-1-i); returns final sample.
    """
    t_end = kw.get("t_end", 0)
    idx = list(range(S.num_timesteps))[::-1]
    if t_end:
        idx = idx[:-t_end]
    x = x_T
    step = ddim_sample if ddim else p_sample
    for k, i in enumerate(idx):
        t = np.full((x.shape[0],), i, dtype=np.int64)
        x = step(S, model, x, t, noises[k], **kw)["sample"]
    return x

"""DiTRotary / DiTRotaryClassifier oracle (numpy fp32): forward, and input-gradient backward.

Test infrastructure (see oracle/__init__.py).  Restates, with weights given as a dict keyed by
the reference's state_dict names:
  guided_diffusion/dit.py:25-26    modulate
  guided_diffusion/dit.py:47-70    TimestepEmbedder (cos||sin, max_period 1e4)
  guided_diffusion/dit.py:219-227  FlattenPatchify1D
  guided_diffusion/dit.py:263-288  RotaryAttention (SDPA branch, scale head_dim**-0.5)
  guided_diffusion/dit.py:332-336  DiTBlockRotary
  guided_diffusion/dit.py:372-376  FinalLayerPatch1D
  guided_diffusion/dit.py:608-634  DiTRotary.unpatchify / forward
  guided_diffusion/dit.py:803-831  DiTRotaryClassifier.forward (plain and chord heads)
  guided_diffusion/condition_functions.py:58-85  grad_nn_zt_mse / grad_nn_zt_chord
PARITY UNPINNED (third-party, restated from the pinned versions' documented behaviour):
  rotary-embedding-torch==0.3.2 rotate_queries_or_keys -> `rotary_tables`/`apply_rotary`
  timm==0.9.2 Mlp (fc1 -> GELU(tanh) -> fc2)           -> `mlp`
The reference obtains input gradients with autograd; here the backward is written out by hand
(weights frozen: dgrad only) and pinned against autograd-generated goldens.
"""
import math
import numpy as np

F32 = np.float32


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def silu(x):
    with np.errstate(over="ignore"):          # exp(+large) -> inf -> x/inf = -0: the correct limit
        return x / (1 + np.exp(-x))


def gelu_tanh(x):
    c = F32(math.sqrt(2.0 / math.pi))
    return F32(0.5) * x * (1 + np.tanh(c * (x + F32(0.044715) * x * x * x)))


def gelu_tanh_grad(x):
    c = F32(math.sqrt(2.0 / math.pi))
    u = c * (x + F32(0.044715) * x * x * x)
    th = np.tanh(u)
    du = c * (1 + F32(3 * 0.044715) * x * x)
    return F32(0.5) * (1 + th) + F32(0.5) * x * (1 - th * th) * du


def layernorm(x, eps, w=None, b=None):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    y = (x - mu) / np.sqrt(var + F32(eps))
    if w is not None:
        y = y * w + b
    return y.astype(F32)


def layernorm_bwd(dy, x, eps, w=None):
    """d/dx of layernorm (affine weight w optional)."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1 / np.sqrt(var + F32(eps))
    xh = (x - mu) * rstd
    g = dy if w is None else dy * w
    return ((g - g.mean(-1, keepdims=True) - xh * (g * xh).mean(-1, keepdims=True)) * rstd).astype(F32)


def timestep_embedding(t, dim=256, max_period=10000):
    """dit.py:47-65 (float32 throughout)."""
    half = dim // 2
    freqs = np.exp(F32(-math.log(max_period)) * np.arange(half, dtype=F32) / F32(half)).astype(F32)
    args = np.asarray(t).astype(F32)[:, None] * freqs[None]
    return np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(F32)


def patchify(x, patch):
    """dit.py:219-224: (N,C,H,W) -> (N, H*W/patch, patch*C); token = h*(W/patch)+w//patch,
    feature = (w%patch)*C + c."""
    n, c, h, w = x.shape
    return x.transpose(0, 2, 3, 1).reshape(n, h * w // patch, patch * c)


def unpatchify(tok, W, out_ch):
    """dit.py:608-616."""
    n = tok.shape[0]
    return tok.reshape(n, -1, W, out_ch).transpose(0, 3, 1, 2)


def rotary_tables(freqs, T):
    """rotary-embedding-torch 0.3.2: angle[pos, 2j] = angle[pos, 2j+1] = pos * freqs[j]."""
    ang = np.arange(T, dtype=F32)[:, None] * np.asarray(freqs, dtype=F32)[None]
    return np.cos(ang).astype(F32), np.sin(ang).astype(F32)        # (T, rot/2)


def apply_rotary(x, cos, sin, inverse=False):
    """x: (..., T, hd); rotate interleaved pairs of the first 2*cos.shape[1] channels."""
    r = cos.shape[1] * 2
    xr = x[..., :r].reshape(x.shape[:-1] + (r // 2, 2))
    a, b = xr[..., 0], xr[..., 1]
    s = -sin if inverse else sin
    out = np.stack((a * cos - b * s, b * cos + a * s), axis=-1).reshape(x.shape[:-1] + (r,))
    return np.concatenate((out, x[..., r:]), axis=-1).astype(F32)


def mlp(x, sd, pre):
    h = linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    return linear(gelu_tanh(h), sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def attention(m, sd, pre, heads, cos, sin, cache=None):
    n, T, D = m.shape
    hd = D // heads
    qkv = linear(m, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(n, T, 3, heads, hd)
    q, k, v = (qkv[:, :, i].transpose(0, 2, 1, 3) for i in range(3))       # (n,heads,T,hd)
    q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
    s = (q @ k.transpose(0, 1, 3, 2)) * F32(hd ** -0.5)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = (p / p.sum(-1, keepdims=True)).astype(F32)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(n, T, D)
    if cache is not None:
        cache.update(q=q, k=k, v=v, p=p, o=o)
    return linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def block(x, c_silu, sd, i, heads, cos, sin, cache=None):
    """dit.py:332-336."""
    pre = f"blocks.{i}."
    mod = linear(c_silu, sd[pre + "adaLN_modulation.1.weight"], sd[pre + "adaLN_modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = (a[:, None, :] for a in np.split(mod, 6, axis=1))
    m1 = layernorm(x, 1e-6) * (1 + sc1) + sh1
    ca = {} if cache is not None else None
    a = attention(m1, sd, pre + "attn.", heads, cos, sin, ca)
    x1 = (x + g1 * a).astype(F32)
    m2 = layernorm(x1, 1e-6) * (1 + sc2) + sh2
    pre_act = linear(m2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    f = linear(gelu_tanh(pre_act), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    x2 = (x1 + g2 * f).astype(F32)
    if cache is not None:
        cache.update(x=x, x1=x1, sc1=sc1, g1=g1, sc2=sc2, g2=g2, pre_act=pre_act, **ca)
    return x2


def block_bwd(dx2, cache, sd, i, heads, cos, sin):
    """Input gradient of `block` (weights and conditioning are constants)."""
    pre = f"blocks.{i}."
    c = cache
    n, T, D = dx2.shape
    hd = D // heads
    du = (c["g2"] * dx2) @ sd[pre + "mlp.fc2.weight"]
    dm2 = (du * gelu_tanh_grad(c["pre_act"])) @ sd[pre + "mlp.fc1.weight"]
    dx1 = dx2 + layernorm_bwd(dm2 * (1 + c["sc2"]), c["x1"], 1e-6)
    do = ((c["g1"] * dx1) @ sd[pre + "attn.proj.weight"]).reshape(n, T, heads, hd).transpose(0, 2, 1, 3)
    p, q, k, v = c["p"], c["q"], c["k"], c["v"]
    dv = p.transpose(0, 1, 3, 2) @ do
    dp = do @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * F32(hd ** -0.5)
    dq = apply_rotary(ds @ k, cos, sin, inverse=True)
    dk = apply_rotary(ds.transpose(0, 1, 3, 2) @ q, cos, sin, inverse=True)
    dqkv = np.stack((dq, dk, dv), axis=0).transpose(1, 3, 0, 2, 4).reshape(n, T, 3 * D)
    dm1 = dqkv @ sd[pre + "attn.qkv.weight"]
    return (dx1 + layernorm_bwd(dm1 * (1 + c["sc1"]), c["x"], 1e-6)).astype(F32)


def _embed(x, t, sd, patch):
    tok = patchify(x.astype(F32), patch)
    h = silu(linear(tok, sd["x_embedder.MLP.0.weight"], sd["x_embedder.MLP.0.bias"]))
    h = linear(h, sd["x_embedder.MLP.2.weight"], sd["x_embedder.MLP.2.bias"])
    te = timestep_embedding(t)
    c = silu(linear(te, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"]))
    c = linear(c, sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    return h.astype(F32), c.astype(F32)


def dit_forward(sd, x, t, y=None, *, depth, heads, patch=8, out_ch=4, return_tokens=False):
    """DiTRotary.forward dit.py:618-634.  x (N,C,H,16) f32, t (N,) int, y (N,) int or None."""
    h, c = _embed(x, t, sd, patch)
    if y is not None and "y_embedder.embedding_table.weight" in sd:
        c = c + sd["y_embedder.embedding_table.weight"][np.asarray(y)]
    cs = silu(c).astype(F32)
    cos, sin = rotary_tables(sd["rotary_emb.freqs"], h.shape[1])
    for i in range(depth):
        h = block(h, cs, sd, i, heads, cos, sin)
    mod = linear(cs, sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
    sh, sc = (a[:, None, :] for a in np.split(mod, 2, axis=1))
    tok = linear(layernorm(h, 1e-6) * (1 + sc) + sh, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    if return_tokens:
        return tok.astype(F32)
    return np.ascontiguousarray(unpatchify(tok, x.shape[-1], out_ch)).astype(F32)


def _head(z, sd, norm, head):
    z = layernorm(z, 1e-5, sd[norm + ".weight"], sd[norm + ".bias"])
    z1 = linear(z, sd[head + ".0.weight"], sd[head + ".0.bias"])
    return linear(silu(z1), sd[head + ".2.weight"], sd[head + ".2.bias"]), z1


def _head_bwd(dlogits, zin, z1, sd, norm, head):
    dz1 = (dlogits @ sd[head + ".2.weight"])
    sg = 1 / (1 + np.exp(-z1))
    dz1 = dz1 * (sg * (1 + z1 * (1 - sg)))
    dzn = dz1 @ sd[head + ".0.weight"]
    return layernorm_bwd(dzn, zin, 1e-5, sd[norm + ".weight"])


def classifier_forward(sd, x, t, *, depth, heads, patch=8, chord=False, caches=None):
    """DiTRotaryClassifier.forward dit.py:803-831."""
    h, c = _embed(x, t, sd, patch)
    n = h.shape[0]
    h = np.concatenate((np.broadcast_to(sd["cls_token"], (n, 1, h.shape[2])), h), axis=1).astype(F32)
    cs = silu(c).astype(F32)
    cos, sin = rotary_tables(sd["rotary_emb.freqs"], h.shape[1])
    for i in range(depth):
        ca = {} if caches is not None else None
        h = block(h, cs, sd, i, heads, cos, sin, ca)
        if caches is not None:
            caches.append(ca)
    if caches is not None:
        caches.append({"h": h, "cos": cos, "sin": sin})
    if not chord:
        logits, z1 = _head(h[:, 0], sd, "norm", "classifier_head")
        if caches is not None:
            caches[-1]["z1"] = z1
        return logits.astype(F32)
    n_token = x.shape[2] // x.shape[3]
    key, z1k = _head(h[:, 0], sd, "norm_key", "classifier_head_key")
    pooled = h[:, 1:].reshape(n, n_token, -1, h.shape[2]).mean(axis=-2).astype(F32)
    ch, z1c = _head(pooled, sd, "norm", "classifier_head")
    if caches is not None:
        caches[-1].update(z1k=z1k, z1c=z1c, pooled=pooled)
    return key.astype(F32), ch.astype(F32)


def _backbone_bwd(dh, caches, sd, x_shape, depth, heads, patch):
    last = caches[-1]
    for i in reversed(range(depth)):
        dh = block_bwd(dh, caches[i], sd, i, heads, last["cos"], last["sin"])
    dtok = dh[:, 1:]                                          # drop the cls token
    # x_embedder backward: Linear(256,D) <- SiLU <- Linear(32,256) <- patchify
    n, c, H, W = x_shape
    tok = last["tok_in"]
    z = linear(tok, sd["x_embedder.MLP.0.weight"], sd["x_embedder.MLP.0.bias"])
    dz = dtok @ sd["x_embedder.MLP.2.weight"]
    sg = 1 / (1 + np.exp(-z))
    dz = dz * (sg * (1 + z * (1 - sg)))
    dtin = dz @ sd["x_embedder.MLP.0.weight"]                 # (n, T, patch*c)
    return np.ascontiguousarray(dtin.reshape(n, H, W, c).transpose(0, 3, 1, 2)).astype(F32)


def grad_nn_zt_mse(sd, x, t, rule, scale, *, depth, heads, patch=8):
    """condition_functions.py:58-64: d/dx of -sum((cls(x,t)-rule)^2), times classifier_scale.
    Returns (grad, logits)."""
    caches = []
    logits = classifier_forward(sd, x, t, depth=depth, heads=heads, patch=patch, caches=caches)
    caches[-1]["tok_in"] = patchify(x.astype(F32), patch)
    dlogits = (-2 * (logits - rule.astype(F32))).astype(F32)
    h = caches[-1]["h"]
    dh = np.zeros_like(h)
    dh[:, 0] = _head_bwd(dlogits, h[:, 0], caches[-1]["z1"], sd, "norm", "classifier_head")
    g = _backbone_bwd(dh, caches, sd, x.shape, depth, heads, patch)
    return (g * F32(scale)).astype(F32), logits


def grad_nn_zt_chord(sd, x, t, rule, scale, *, depth, heads, patch=8):
    """condition_functions.py:67-85 (both=False): d/dx of -sum CE(chord_logits, rule) * scale."""
    caches = []
    key, ch = classifier_forward(sd, x, t, depth=depth, heads=heads, patch=patch, chord=True, caches=caches)
    caches[-1]["tok_in"] = patchify(x.astype(F32), patch)
    n, n_token, K = ch.shape
    z = ch - ch.max(-1, keepdims=True)
    p = np.exp(z)
    p = p / p.sum(-1, keepdims=True)
    onehot = np.eye(K, dtype=F32)[np.asarray(rule).reshape(n, n_token)]
    dch = (onehot - p).astype(F32)                             # d(-CE)/dlogits
    h = caches[-1]["h"]
    dpool = _head_bwd(dch, caches[-1]["pooled"], caches[-1]["z1c"], sd, "norm", "classifier_head")
    per = (h.shape[1] - 1) // n_token
    dh = np.zeros_like(h)
    dh[:, 1:] = np.repeat(dpool / F32(per), per, axis=1)
    g = _backbone_bwd(dh, caches, sd, x.shape, depth, heads, patch)
    return (g * F32(scale)).astype(F32), (key, ch)

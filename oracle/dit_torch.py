"""DiTRotary forward + one DDIM step in torch on the HOST CPU (fp32) -- the `cpu_baseline` leg of bench.py.

Test infrastructure (see oracle/__init__.py): a checker / reported baseline, never the product path.  The reference is a
torch program (ATen ops on CPU when no GPU is given), so the side-by-side CPU number of BASELINE config[0] ("C1":
unconditional DDIM-50, batch 2, DiTRotary_XL_8) is timed with the same operator set the reference would execute --
F.linear, F.layer_norm, F.scaled_dot_product_attention, F.gelu(tanh) -- restated here because the reference itself cannot
travel to the GPU box.  Function by function it follows:
  guided_diffusion/dit.py:47-70    TimestepEmbedder            guided_diffusion/dit.py:219-227  FlattenPatchify1D
  guided_diffusion/dit.py:263-288  RotaryAttention (SDPA)      guided_diffusion/dit.py:332-336  DiTBlockRotary
  guided_diffusion/dit.py:372-376  FinalLayerPatch1D           guided_diffusion/dit.py:608-634  unpatchify / forward
  guided_diffusion/gaussian_diffusion.py:881-976 ddim_sample (eta given), :252-364 p_mean_variance (EPSILON, fixed variance)
rotary-embedding-torch 0.3.2 / timm 0.9.2 Mlp restated as in dit_np.py (parity unpinned slice, see oracle/__init__.py).
Pinned by tests/test_oracle_golden.py against the same reference-generated fixtures as the numpy oracle.
"""
import math

import torch
import torch.nn.functional as F


def to_torch(sd):
    import numpy as np
    return {k: (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))).float().cpu() for k, v in sd.items()}


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _rotary(x, freqs):
    """rotate_queries_or_keys(seq_dim=-2): interleaved pairs of the first 2*len(freqs) channels of every head."""
    T = x.shape[-2]
    ang = (torch.arange(T, dtype=torch.float32)[:, None] * freqs[None]).repeat_interleave(2, dim=-1)
    r = ang.shape[-1]
    xm, xr = x[..., :r], x[..., r:]
    pairs = xm.reshape(*xm.shape[:-1], -1, 2)
    rot = torch.stack((-pairs[..., 1], pairs[..., 0]), dim=-1).reshape(xm.shape)
    return torch.cat((xm * ang.cos() + rot * ang.sin(), xr), dim=-1)


def _modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


@torch.no_grad()
def dit_forward(sd, x, t, y=None, *, depth, heads, patch=8, out_ch=4):
    """x (N,C,H,W) f32, t (N,) int64, y (N,) int64 or None -> eps (N,out_ch,H,W)."""
    n, c, hh, w = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(n, hh * w // patch, patch * c)
    h = F.linear(F.silu(F.linear(tok, sd["x_embedder.MLP.0.weight"], sd["x_embedder.MLP.0.bias"])),
                 sd["x_embedder.MLP.2.weight"], sd["x_embedder.MLP.2.bias"])
    cc = F.linear(F.silu(F.linear(timestep_embedding(t), sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])),
                  sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    if y is not None and "y_embedder.embedding_table.weight" in sd:
        cc = cc + sd["y_embedder.embedding_table.weight"][y]
    cs = F.silu(cc)
    D = h.shape[-1]
    hd = D // heads
    freqs = sd["rotary_emb.freqs"]
    for i in range(depth):
        p = f"blocks.{i}."
        sh1, sc1, g1, sh2, sc2, g2 = F.linear(cs, sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"]).chunk(6, dim=1)
        m1 = _modulate(F.layer_norm(h, (D,), eps=1e-6), sh1, sc1)
        qkv = F.linear(m1, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(n, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = _rotary(qkv[0], freqs), _rotary(qkv[1], freqs), qkv[2]
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n, -1, D)
        h = h + g1.unsqueeze(1) * F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        m2 = _modulate(F.layer_norm(h, (D,), eps=1e-6), sh2, sc2)
        f = F.linear(F.gelu(F.linear(m2, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]), approximate="tanh"),
                     sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        h = h + g2.unsqueeze(1) * f
    sh, sc = F.linear(cs, sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"]).chunk(2, dim=1)
    out = F.linear(_modulate(F.layer_norm(h, (D,), eps=1e-6), sh, sc), sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    return out.reshape(n, -1, w, out_ch).permute(0, 3, 1, 2).contiguous()


@torch.no_grad()
def ddim_step(S, model, x, t, noise, eta=1.0):
    """One ddim_sample call (no guidance, clip_denoised False) with the float64 tables of oracle.diffusion_np.Schedule `S`."""
    def ex(arr):
        return torch.from_numpy(arr)[t].float().view(-1, 1, 1, 1)
    tmap = torch.tensor(S.timestep_map, dtype=torch.int64)[t]
    eps = model(x, tmap)
    c1, c2 = ex(S.sqrt_recip_alphas_cumprod), ex(S.sqrt_recipm1_alphas_cumprod)
    x0 = c1 * x - c2 * eps
    e2 = (c1 * x - x0) / c2
    ab, abp = ex(S.alphas_cumprod), ex(S.alphas_cumprod_prev)
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * e2
    mask = (t != 0).float().view(-1, 1, 1, 1)
    return mean + mask * sigma * noise, x0

"""Deterministic synthetic ("random-init") weights for benchmarks, smoke runs and parity tests.

There is no network for checkpoints, so every measured / tested configuration uses weights of
the reference architecture drawn from a counter-based generator: value i of tensor `key` is a
pure function of (seed, key, i) -- splitmix64 finaliser -> uniform in [-a, a) -- so numpy on any
host (build container, GPU box) reproduces them bit-for-bit without shipping 2.7 GB fixtures.

The reference's own initialiser zeroes every adaLN projection and the final linear
(guided_diffusion/dit.py:597-606), which makes a fresh DiTRotary output exactly 0 and would
leave every kernel after adaLN untested (SURVEY headline 4); the scales below are chosen so
that shift/scale/gate and the output are O(0.1..1) instead.

State-dict key layout follows the reference: guided_diffusion/dit.py:565-576,
:770-778 and taming/modules/diffusionmodules/model.py:436-504 (Lightning ckpt prefixes
`decoder.` / `post_quant_conv.`, taming/models/klvae_pedal.py:50-59).
"""
import zlib
import numpy as np

def _base(seed, key):
    return ((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B9)) & 0xFFFFFFFF)


def uniform(seed, key, shape, std):
    """float32 tensor, i.i.d. uniform with standard deviation `std` (a = std*sqrt(3)).

    value[i] = lowbias32(i + base(seed, key)) -> top 24 bits -> [-a, a)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint32) + np.uint32(_base(seed, key))
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    u = (x >> np.uint32(8)).astype(np.float32)
    a = np.float32(std * 3 ** 0.5)
    return ((u * np.float32(2.0 ** -23) - np.float32(1.0)) * a).reshape(shape)


def uniform_torch(seed, key, shape, std, device):
    """Same values as `uniform`, generated on `device` with torch integer ops (plumbing only:
    lets the GPU box materialise the 674 M synthetic DiT-XL weights in about a second)."""
    import torch
    n = int(np.prod(shape)) if len(shape) else 1
    M = 0xFFFFFFFF
    x = (torch.arange(n, dtype=torch.int64, device=device) + _base(seed, key)) & M
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & M
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & M
    x = x ^ (x >> 16)
    u = (x >> 8).to(torch.float32)
    a = float(np.float32(std * 3 ** 0.5))
    return ((u * float(2.0 ** -23) - 1.0) * a).reshape(shape)


# ------------------------------------------------------------------ architecture specs
def dit_param_shapes(*, depth, hidden, heads, patch=8, in_ch=4, out_ch=4, num_classes=0,
                     class_dropout=True, classifier=False, cls_classes=0, chord=False):
    """Ordered (key, shape) list == the reference module's state_dict() key order."""
    D = hidden
    rot = int(D // heads * 0.5)
    s = [("x_embedder.MLP.0.weight", (256, in_ch * patch)), ("x_embedder.MLP.0.bias", (256,)),
         ("x_embedder.MLP.2.weight", (D, 256)), ("x_embedder.MLP.2.bias", (D,)),
         ("t_embedder.mlp.0.weight", (D, 256)), ("t_embedder.mlp.0.bias", (D,)),
         ("t_embedder.mlp.2.weight", (D, D)), ("t_embedder.mlp.2.bias", (D,))]
    if classifier:
        s = [("cls_token", (1, 1, D))] + s
    elif num_classes:
        s.append(("y_embedder.embedding_table.weight", (num_classes + int(class_dropout), D)))
    s.append(("rotary_emb.freqs", (rot // 2,)))
    for i in range(depth):
        p = f"blocks.{i}."
        s += [(p + "attn.rotary_emb.freqs", (rot // 2,)),
              (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
              (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
              (p + "mlp.fc1.weight", (4 * D, D)), (p + "mlp.fc1.bias", (4 * D,)),
              (p + "mlp.fc2.weight", (D, 4 * D)), (p + "mlp.fc2.bias", (D,)),
              (p + "adaLN_modulation.1.weight", (6 * D, D)), (p + "adaLN_modulation.1.bias", (6 * D,))]
    if classifier:
        s += [("norm.weight", (D,)), ("norm.bias", (D,)),
              ("classifier_head.0.weight", (D // 4, D)), ("classifier_head.0.bias", (D // 4,)),
              ("classifier_head.2.weight", (cls_classes, D // 4)), ("classifier_head.2.bias", (cls_classes,))]
        if chord:
            s += [("norm_key.weight", (D,)), ("norm_key.bias", (D,)),
                  ("classifier_head_key.0.weight", (D // 4, D)), ("classifier_head_key.0.bias", (D // 4,)),
                  ("classifier_head_key.2.weight", (25, D // 4)), ("classifier_head_key.2.bias", (25,))]
    else:
        s += [("final_layer.linear.weight", (patch * out_ch, D)), ("final_layer.linear.bias", (patch * out_ch,)),
              ("final_layer.adaLN_modulation.1.weight", (2 * D, D)), ("final_layer.adaLN_modulation.1.bias", (2 * D,))]
    return s


def rotary_freqs(rot_dim, theta=10000.0):
    """rotary-embedding-torch 0.3.2 'lang' freqs: 1/theta^(2j/dim), float32."""
    j = np.arange(0, rot_dim, 2, dtype=np.float32)
    return (np.float32(1.0) / (np.float32(theta) ** (j / np.float32(rot_dim)))).astype(np.float32)


def _gen(seed, key, shape, std, device):
    return uniform(seed, key, shape, std) if device is None else uniform_torch(seed, key, shape, std, device)


def _const(arr, device):
    if device is None:
        return arr
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device)


def dit_state_dict(seed, *, final_std=None, device=None, **arch):
    """Synthetic weights for DiTRotary / DiTRotaryClassifier.

    device=None -> dict of numpy float32 arrays; device="cuda" -> the SAME values as torch tensors
    generated on the device (bit-identical; see uniform_torch)."""
    sd = {}
    D, heads = arch["hidden"], arch["heads"]
    for key, shape in dit_param_shapes(**arch):
        if key.endswith("freqs"):
            sd[key] = _const(rotary_freqs(int(D // heads * 0.5)), device)
        elif key == "cls_token":
            sd[key] = _gen(seed, key, shape, 0.5, device)
        elif key.endswith("embedding_table.weight"):
            sd[key] = _gen(seed, key, shape, 0.3, device)
        elif key.startswith("norm") and key.endswith(".weight"):
            sd[key] = _gen(seed, key, shape, 0.1, device) + 1.0
        elif key.endswith(".bias"):
            sd[key] = _gen(seed, key, shape, 0.05, device)
        else:                                             # Linear weight (out, in)
            fan_out, fan_in = shape
            std = (2.0 / (fan_in + fan_out)) ** 0.5       # xavier
            if "adaLN_modulation" in key:
                std = 0.6 / fan_in ** 0.5
            elif key.startswith("t_embedder.mlp"):
                std = 2.0 / fan_in ** 0.5
            elif key.startswith("final_layer.linear"):
                std = final_std if final_std is not None else 1.0 / fan_in ** 0.5
            sd[key] = _gen(seed, key, shape, std, device)
    return sd


def vae_decoder_param_shapes(ch=128, ch_mult=(1, 2, 2, 4), num_res_blocks=2, z_ch=4, out_ch=3, embed_dim=4):
    """(key, shape) list of the checkpoint slice AutoencoderKL.decode reads, in module order."""
    s = [("post_quant_conv.weight", (z_ch, embed_dim, 1, 1)), ("post_quant_conv.bias", (z_ch,))]
    d = "decoder."
    bi = ch * ch_mult[-1]
    s += [(d + "conv_in.weight", (bi, z_ch, 3, 3)), (d + "conv_in.bias", (bi,))]

    def res(p, cin, cout):
        r = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)),
             (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,)),
             (p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)),
             (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))]
        if cin != cout:
            r += [(p + "nin_shortcut.weight", (cout, cin, 1, 1)), (p + "nin_shortcut.bias", (cout,))]
        return r

    s += res(d + "mid.block_1.", bi, bi)
    a = d + "mid.attn_1."
    s += [(a + "norm.weight", (bi,)), (a + "norm.bias", (bi,))]
    for nm in ("q", "k", "v", "proj_out"):
        s += [(a + nm + ".weight", (bi, bi, 1, 1)), (a + nm + ".bias", (bi,))]
    s += res(d + "mid.block_2.", bi, bi)
    for lvl in reversed(range(len(ch_mult))):
        bo = ch * ch_mult[lvl]
        for ib in range(num_res_blocks + 1):
            s += res(f"{d}up.{lvl}.block.{ib}.", bi, bo)
            bi = bo
        if lvl != 0:
            s += [(f"{d}up.{lvl}.upsample.conv.weight", (bi, bi, 3, 3)), (f"{d}up.{lvl}.upsample.conv.bias", (bi,))]
    s += [(d + "norm_out.weight", (bi,)), (d + "norm_out.bias", (bi,)),
          (d + "conv_out.weight", (out_ch, bi, 3, 3)), (d + "conv_out.bias", (out_ch,))]
    return s


def vae_encoder_param_shapes(ch=128, ch_mult=(1, 2, 2, 4), num_res_blocks=2, z_ch=4, in_ch=3, embed_dim=4):
    """(key, shape) list of the checkpoint slice AutoencoderKL.encode reads (taming Encoder, model.py:342-402, + quant_conv)."""
    e = "encoder."
    s = [(e + "conv_in.weight", (ch, in_ch, 3, 3)), (e + "conv_in.bias", (ch,))]

    def res(p, cin, cout):
        r = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)),
             (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,)),
             (p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)),
             (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))]
        if cin != cout:
            r += [(p + "nin_shortcut.weight", (cout, cin, 1, 1)), (p + "nin_shortcut.bias", (cout,))]
        return r

    bi = ch
    for lvl in range(len(ch_mult)):
        bo = ch * ch_mult[lvl]
        for ib in range(num_res_blocks):
            s += res(f"{e}down.{lvl}.block.{ib}.", bi, bo)
            bi = bo
        if lvl != len(ch_mult) - 1:
            s += [(f"{e}down.{lvl}.downsample.conv.weight", (bi, bi, 3, 3)), (f"{e}down.{lvl}.downsample.conv.bias", (bi,))]
    s += res(e + "mid.block_1.", bi, bi)
    a = e + "mid.attn_1."
    s += [(a + "norm.weight", (bi,)), (a + "norm.bias", (bi,))]
    for nm in ("q", "k", "v", "proj_out"):
        s += [(a + nm + ".weight", (bi, bi, 1, 1)), (a + nm + ".bias", (bi,))]
    s += res(e + "mid.block_2.", bi, bi)
    s += [(e + "norm_out.weight", (bi,)), (e + "norm_out.bias", (bi,)),
          (e + "conv_out.weight", (2 * z_ch, bi, 3, 3)), (e + "conv_out.bias", (2 * z_ch,)),
          ("quant_conv.weight", (2 * embed_dim, 2 * z_ch, 1, 1)), ("quant_conv.bias", (2 * embed_dim,))]
    return s


def vae_state_dict(seed, device=None, encoder=False, **arch):
    sd = {}
    shapes = vae_decoder_param_shapes(**arch) + (vae_encoder_param_shapes() if encoder else [])
    for key, shape in shapes:
        if "norm" in key and key.endswith(".weight"):
            sd[key] = _gen(seed, key, shape, 0.1, device) + 1.0
        elif key.endswith(".bias"):
            sd[key] = _gen(seed, key, shape, 0.05, device)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.6 if "conv_out" not in key else 1.0      # keep activations O(1) through swish
            sd[key] = _gen(seed, key, shape, gain / fan_in ** 0.5, device)
    return sd


DIT_XL_8 = dict(depth=28, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4)
DIT_B_8 = dict(depth=12, hidden=768, heads=12, patch=8, in_ch=4, out_ch=4)
CLS_S_8 = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True)

"""taming KL-VAE decoder oracle (numpy fp32).

Test infrastructure (see oracle/__init__.py).  Restates:
  taming/models/klvae_pedal.py:80-85                    AutoencoderKL.decode (post_quant_conv -> decoder)
  taming/modules/diffusionmodules/model.py:29-35         swish, GroupNorm(32, eps=1e-6, affine)
  taming/modules/diffusionmodules/model.py:49-53         Upsample (nearest x2 + conv3x3)
  taming/modules/diffusionmodules/model.py:117-137       ResnetBlock.forward (temb=None, dropout 0)
  taming/modules/diffusionmodules/model.py:168-192       AttnBlock.forward
  taming/modules/diffusionmodules/model.py:506-537       Decoder.forward
  guided_diffusion/midi_util.py:42-64                    decode_sample_for_midi (uint8 quantisation)
  taming/modules/diffusionmodules/model.py:56-75         Downsample (zero pad (0,1,0,1) + conv3x3 stride 2)
  taming/modules/diffusionmodules/model.py:404-433       Encoder.forward
  taming/models/klvae_pedal.py:61-68                     AutoencoderKL.encode_save (encoder -> quant_conv)
  guided_diffusion/gaussian_diffusion.py:1382-1395       _encode (tile the roll, keep the posterior mean, scale)
Weights: dict keyed like the Lightning checkpoint's ["state_dict"] ("decoder.*", "post_quant_conv.*").
Config (taming-transformers/configs/pr/kl/f8-all-onset.yaml): ch=128, ch_mult=(1,2,2,4),
num_res_blocks=2, attn_resolutions=[], z_channels=4, out_ch=3, resolution=128.
"""
import numpy as np

F32 = np.float32


def swish(x):
    return (x / (1 + np.exp(-x))).astype(F32)


def groupnorm(x, w, b, groups=32, eps=1e-6):
    m, c, h, wd = x.shape
    g = x.reshape(m, groups, -1)
    mu = g.mean(-1, keepdims=True)
    var = ((g - mu) ** 2).mean(-1, keepdims=True)
    y = ((g - mu) / np.sqrt(var + F32(eps))).reshape(m, c, h, wd)
    return (y * w[None, :, None, None] + b[None, :, None, None]).astype(F32)


def conv2d(x, w, b, pad):
    """NCHW conv, stride 1, kernel k in {1,3}; one matmul per tap, fp32 accumulate."""
    m, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    xh = np.ascontiguousarray(x.transpose(0, 2, 3, 1))
    if pad:
        xh = np.pad(xh, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    out = np.zeros((m * h * wd, cout), dtype=F32)
    for ky in range(k):
        for kx in range(k):
            tap = np.ascontiguousarray(xh[:, ky:ky + h, kx:kx + wd, :]).reshape(-1, cin)
            out += tap @ np.ascontiguousarray(w[:, :, ky, kx].T)          # BLAS sgemm
    out += b
    return np.ascontiguousarray(out.reshape(m, h, wd, cout).transpose(0, 3, 1, 2))


def resnet_block(x, sd, p):
    h = conv2d(swish(groupnorm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])), sd[p + "conv1.weight"], sd[p + "conv1.bias"], 1)
    h = conv2d(swish(groupnorm(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])), sd[p + "conv2.weight"], sd[p + "conv2.bias"], 1)
    if p + "nin_shortcut.weight" in sd:
        x = conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"], 0)
    return (x + h).astype(F32)


def attn_block(x, sd, p):
    m, c, h, w = x.shape
    hn = groupnorm(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
    q = conv2d(hn, sd[p + "q.weight"], sd[p + "q.bias"], 0).reshape(m, c, h * w)
    k = conv2d(hn, sd[p + "k.weight"], sd[p + "k.bias"], 0).reshape(m, c, h * w)
    v = conv2d(hn, sd[p + "v.weight"], sd[p + "v.bias"], 0).reshape(m, c, h * w)
    s = (q.transpose(0, 2, 1) @ k) * F32(int(c) ** -0.5)            # (m, hw_q, hw_k)
    s = s - s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr = (pr / pr.sum(-1, keepdims=True)).astype(F32)
    o = (v @ pr.transpose(0, 2, 1)).reshape(m, c, h, w)             # o[c, i] = sum_j v[c, j] p[i, j]
    return (x + conv2d(o, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"], 0)).astype(F32)


def upsample_nearest2(x):
    return x.repeat(2, axis=2).repeat(2, axis=3)


def decode(sd, z, ch_mult=(1, 2, 2, 4), num_res_blocks=2):
    """AutoencoderKL.decode: z (M,4,16,16) -> (M,3,128,128)."""
    z = conv2d(z.astype(F32), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"], 0)
    d = "decoder."
    h = conv2d(z, sd[d + "conv_in.weight"], sd[d + "conv_in.bias"], 1)
    h = resnet_block(h, sd, d + "mid.block_1.")
    h = attn_block(h, sd, d + "mid.attn_1.")
    h = resnet_block(h, sd, d + "mid.block_2.")
    for lvl in reversed(range(len(ch_mult))):
        for ib in range(num_res_blocks + 1):
            h = resnet_block(h, sd, f"{d}up.{lvl}.block.{ib}.")
        if lvl != 0:
            h = conv2d(upsample_nearest2(h), sd[f"{d}up.{lvl}.upsample.conv.weight"], sd[f"{d}up.{lvl}.upsample.conv.bias"], 1)
    h = swish(groupnorm(h, sd[d + "norm_out.weight"], sd[d + "norm_out.bias"]))
    return conv2d(h, sd[d + "conv_out.weight"], sd[d + "conv_out.bias"], 1)


def downsample_conv(x, w, b):
    """Downsample.forward with_conv: pad right/bottom by one zero, 3x3 conv with stride 2, no further padding."""
    m, cin, h, wd = x.shape
    cout = w.shape[0]
    xh = np.pad(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), ((0, 0), (0, 1), (0, 1), (0, 0)))
    ho, wo = h // 2, wd // 2
    out = np.zeros((m * ho * wo, cout), dtype=F32)
    for ky in range(3):
        for kx in range(3):
            tap = np.ascontiguousarray(xh[:, ky:ky + 2 * ho:2, kx:kx + 2 * wo:2, :]).reshape(-1, cin)
            out += tap @ np.ascontiguousarray(w[:, :, ky, kx].T)
    out += b
    return np.ascontiguousarray(out.reshape(m, ho, wo, cout).transpose(0, 3, 1, 2))


def encode_moments(sd, x, ch_mult=(1, 2, 2, 4), num_res_blocks=2):
    """AutoencoderKL.encode_save(x, range_fix=False): x (M,3,128,128) -> moments (M,8,16,16) = mean | logvar."""
    e = "encoder."
    h = conv2d(x.astype(F32), sd[e + "conv_in.weight"], sd[e + "conv_in.bias"], 1)
    for lvl in range(len(ch_mult)):
        for ib in range(num_res_blocks):
            h = resnet_block(h, sd, f"{e}down.{lvl}.block.{ib}.")
        if lvl != len(ch_mult) - 1:
            h = downsample_conv(h, sd[f"{e}down.{lvl}.downsample.conv.weight"], sd[f"{e}down.{lvl}.downsample.conv.bias"])
    h = resnet_block(h, sd, e + "mid.block_1.")
    h = attn_block(h, sd, e + "mid.attn_1.")
    h = resnet_block(h, sd, e + "mid.block_2.")
    h = swish(groupnorm(h, sd[e + "norm_out.weight"], sd[e + "norm_out.bias"]))
    h = conv2d(h, sd[e + "conv_out.weight"], sd[e + "conv_out.bias"], 1)
    return conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"], 0)


def encode_latent(sd, roll, scale_factor=1.0):
    """_encode: roll (B,3,128,128k) -> latent (B,4,16k,16) * scale_factor."""
    b, _, h, w = roll.shape
    k = w // h
    micro = np.concatenate(np.split(roll, k, axis=-1), axis=0)
    z = encode_moments(sd, micro)[:, :4]
    z = np.concatenate(np.split(z, k, axis=0), axis=-1)
    return (z.transpose(0, 1, 3, 2) * F32(scale_factor)).astype(F32)


def quantise_roll(roll, threshold=-0.95):
    """midi_util.py:59-63: threshold background, scale to [0,127], truncate to uint8,
    (B,3,128,T) -> (B,128,T,3)."""
    r = roll.astype(F32).copy()
    r[r <= F32(threshold)] = -1.0
    q = np.clip((r + 1) * F32(63.5), 0, 127).astype(np.uint8)
    return np.ascontiguousarray(q.transpose(0, 2, 3, 1))

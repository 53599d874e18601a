"""AutoencoderKL (decode side) -- reference API (taming/models/klvae_pedal.py:17-85), native MI355X decoder.

Owns the parameters of `post_quant_conv` and `decoder.*` under the SAME state_dict keys as the reference's
Lightning checkpoint (weights under ckpt["state_dict"]; prefixes decoder. / post_quant_conv. / encoder. /
quant_conv. / loss.), loads them with strict=False like init_from_ckpt, and decodes through librgm_hip.so:
    decode(z)            (M,4,16,16) -> (M,3,128,128)                       [AutoencoderKL.decode]
    decode_latent(x, s)  (N,4,H,16)/s -> (N,3,128,8H)  fused _decode        [gaussian_diffusion._decode]
    decode_latent_u8(..) same, straight to the uint8 (N,128,8H,3) roll      [midi_util.decode_sample_for_midi]
The encoder / training parts of the reference class are out of scope of the sampling path.
"""
import ctypes as C

import torch
import torch.nn as nn

from rgm import native as _rgm
from rgm.synth import vae_decoder_param_shapes, vae_encoder_param_shapes


def _attach(root, dotted, param):
    *path, leaf = dotted.split(".")
    mod = root
    for name in path:
        nxt = mod._modules.get(name)
        if nxt is None:
            nxt = nn.Module()
            mod.add_module(name, nxt)
        mod = nxt
    mod.register_parameter(leaf, param)


class DiagonalGaussianDistribution:
    """Posterior of AutoencoderKL.encode (taming/modules/distributions/distributions.py:24-62): host-side view of the moments."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape, device=self.parameters.device)

    def mode(self):
        return self.mean


# globals a tensor checkpoint legitimately needs (the same set torch's weights_only unpickler admits, plus data-only numpy
# reconstruction for the scalars Lightning keeps in callback state); everything else -- the rest of builtins included --
# resolves to an inert stub, so a REDUCE op in a crafted file can call nothing but these constructors
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "bytearray"), ("builtins", "complex"), ("builtins", "slice"),
    ("_codecs", "encode"),
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
}
_SAFE_TORCH_ATTRS = {"float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool", "complex64",
                     "FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
                     "CharStorage", "ByteStorage", "BoolStorage"}


def _numpy_scalar(dtype, data):
    """data-only stand-in for numpy.core.multiarray.scalar (the real one unpickles `data` for object dtypes)"""
    import numpy as np
    dtype = np.dtype(dtype)
    if dtype.hasobject or not isinstance(data, (bytes, bytearray)):
        return None
    return np.frombuffer(data, dtype=dtype, count=1)[0]


def read_lightning_state_dict(path):
    """["state_dict"] of a torch / pytorch-lightning checkpoint without importing what else it pickles: tensors and plain
    containers load normally (weights_only first); if the file references classes this stack does not have (Lightning
    callbacks, omegaconf nodes), it is re-read with an unpickler that resolves ONLY an explicit allow-list of constructors
    (_SAFE_GLOBALS: tensor / storage rebuilders, torch dtypes, OrderedDict, plain builtin containers and scalars, data-only
    numpy reconstruction) and replaces every other global -- builtins such as eval / exec / getattr / __import__ and all other
    torch / numpy callables included -- by an inert stub class: a crafted checkpoint cannot run code through a pickle REDUCE."""
    import pickle
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        class _Stub:
            def __init__(self, *a, **k):
                pass

            def __setstate__(self, state):
                pass

            def __call__(self, *a, **k):
                return _Stub()

        class _Unpickler(pickle.Unpickler):
            def find_class(self, module, name):
                if (module, name) in _SAFE_GLOBALS or (module == "torch" and name in _SAFE_TORCH_ATTRS):
                    return super().find_class(module, name)
                if (module, name) in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
                    return _numpy_scalar
                return type(name, (_Stub,), {"__module__": module})

        class _Pickle:
            Unpickler = _Unpickler
            load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
            __name__ = "pickle"
        obj = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_Pickle)
    sd = obj["state_dict"] if isinstance(obj, dict) and "state_dict" in obj else obj
    if not isinstance(sd, dict) or not all(torch.is_tensor(v) for v in sd.values()):
        raise ValueError(f"{path}: no tensor state_dict found")
    return dict(sd)


DECODE_CHUNK_SAMPLES = int(__import__("os").environ.get("RGM_VAE_CHUNK_SAMPLES", "0"))


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig=None, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        dd = dict(ch=128, ch_mult=(1, 2, 2, 4), num_res_blocks=2, z_channels=4, out_ch=3, attn_resolutions=[])
        dd.update({k: v for k, v in dict(ddconfig or {}).items() if k in dd})
        if (dd["ch"], tuple(dd["ch_mult"]), dd["num_res_blocks"], dd["z_channels"], dd["out_ch"], list(dd["attn_resolutions"])) \
                != (128, (1, 2, 2, 4), 2, 4, 3, []):
            raise NotImplementedError("the native decoder implements the kl/f8-all-onset config (ch 128, mult 1-2-2-4)")
        self.embed_dim = embed_dim
        for key, shape in vae_decoder_param_shapes() + vae_encoder_param_shapes():
            p = nn.Parameter(torch.empty(shape))
            with torch.no_grad():
                if key.endswith("weight") and len(shape) == 4:
                    nn.init.kaiming_uniform_(p, a=5 ** 0.5)
                elif "norm" in key and key.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
            _attach(self, key, p)
        self._handle, self._dirty, self._ws = None, True, None
        self.register_load_state_dict_post_hook(lambda m, _: setattr(m, "_dirty", True))
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=list(ignore_keys))

    def init_from_ckpt(self, path, ignore_keys=list()):
        """Restore a Lightning checkpoint's ["state_dict"] (reference klvae_pedal.py:50-59).  The reference's VAE checkpoint
        is a pytorch-lightning 1.0.8 file: besides the tensors it pickles callback classes / omegaconf containers that this
        stack does not ship, so it is read with an unpickler that stubs every non-torch global (read_lightning_state_dict);
        every decoder / post_quant_conv (and, when present, encoder / quant_conv) parameter must be found -- a key mismatch
        raises instead of leaving layers at their random initialisation."""
        sd = read_lightning_state_dict(path)
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        own = self.state_dict()
        have_encoder = any(k.startswith("encoder.") for k in sd)
        need = [k for k in own if not (k.startswith(("encoder.", "quant_conv.")) and not have_encoder)]
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"{path}: {len(missing)} AutoencoderKL parameters are missing from the checkpoint, e.g. {missing[:4]}")
        bad = [k for k in need if tuple(sd[k].shape) != tuple(own[k].shape)]
        if bad:
            raise ValueError(f"{path}: shape mismatch for {bad[:4]} (kl/f8-all-onset expects {tuple(own[bad[0]].shape)})")
        self.load_state_dict({k: sd[k] for k in need}, strict=False)          # loss.* (discriminator) keys are not ours
        print(f"Restored from {path}")

    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def _ensure_native(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _rgm.RgmError("AutoencoderKL.decode runs only on a HIP device; there is no CPU path in the product")
        if self._handle is None:
            h = C.c_void_p()
            with torch.cuda.device(dev):
                _rgm.check(_rgm.lib.rgm_vae_create(C.byref(h)))
            self._handle, self._dirty = h, True
        if self._dirty:
            torch.cuda.synchronize(dev)
            with torch.cuda.device(dev):
                for key, p in self.state_dict().items():
                    t = p.detach().to(torch.float32).contiguous()
                    shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                    _rgm.check(_rgm.lib.rgm_vae_set_param(self._handle, key.encode(), _rgm.ptr(t), shape, t.dim()))
            self._dirty = False
        return dev

    def _workspace(self, M, dev):
        need = _rgm.lib.rgm_vae_workspace_bytes(self._handle, M)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws, need

    @torch.no_grad()
    def decode(self, z, octave=False):
        if octave:
            raise NotImplementedError("octave re-indexing is unused by the sampling path")
        _rgm.require_cuda(z)
        dev = self._ensure_native()
        z = z.detach().to(torch.float32).contiguous()
        M = z.shape[0]
        assert tuple(z.shape[1:]) == (4, 16, 16), "the decoder consumes 16x16 latent squares"
        out = torch.empty((M, 3, 128, 128), dtype=torch.float32, device=dev)
        ws, need = self._workspace(M, dev)
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_vae_decode(self._handle, _rgm.ptr(z), _rgm.ptr(out), M, _rgm.ptr(ws), need, _rgm.current_stream()))
        return out

    @torch.no_grad()
    def decode_latent(self, latent, scale_factor=1., want_u8=False, threshold=-0.95, want_float=True):
        """(N,4,H,16) latent -> (N,3,128,8H) roll [and/or (N,128,8H,3) uint8], squares gathered in-kernel."""
        _rgm.require_cuda(latent)
        dev = self._ensure_native()
        latent = latent.detach().to(torch.float32).contiguous()
        N, Cc, H, W = latent.shape
        assert Cc == 4 and W == 16 and H % 16 == 0, f"latent must be (N,4,16k,16), got {tuple(latent.shape)}"
        roll = torch.empty((N, 3, 128, 8 * H), dtype=torch.float32, device=dev) if want_float else None
        u8 = torch.empty((N, 128, 8 * H, 3), dtype=torch.uint8, device=dev) if want_u8 else None
        # samples per native call (0 = all at once): a chunk whose largest activation fits the 256 MB Infinity Cache keeps the
        # GroupNorm -> conv -> GroupNorm traffic of the decoder out of HBM (experiment: RGM_VAE_CHUNK_SAMPLES)
        chunk = DECODE_CHUNK_SAMPLES if 0 < DECODE_CHUNK_SAMPLES < N else N
        ws, need = self._workspace(chunk * (H // 16), dev)
        with torch.cuda.device(dev):
            for n0 in range(0, N, chunk):
                nn_ = min(chunk, N - n0)
                _rgm.check(_rgm.lib.rgm_vae_decode_latent(self._handle, _rgm.ptr(latent[n0:n0 + nn_]), 1.0 / float(scale_factor),
                                                          _rgm.ptr(roll[n0:n0 + nn_]) if roll is not None else None,
                                                          _rgm.ptr(u8[n0:n0 + nn_]) if u8 is not None else None, float(threshold), nn_, H,
                                                          _rgm.ptr(ws), need, _rgm.current_stream()))
        if want_u8 and want_float:
            return roll, u8
        return u8 if want_u8 else roll

    @torch.no_grad()
    def decode_latent_save(self, latent, scale_factor=1.):
        """decode_latent keeping the activations decode_latent_vjp needs (rgm_vae_decode_latent_save): what the reference gets
        from autograd when dps_rule differentiates rule(_decode(x0_hat)) (gaussian_diffusion.py:425-433)."""
        _rgm.require_cuda(latent)
        dev = self._ensure_native()
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_vae_enable_grad(self._handle))            # no-op once built; set_param invalidates
        latent = latent.detach().to(torch.float32).contiguous()
        N, Cc, H, W = latent.shape
        assert Cc == 4 and W == 16 and H % 16 == 0, f"latent must be (N,4,16k,16), got {tuple(latent.shape)}"
        need = _rgm.lib.rgm_vae_grad_workspace_bytes(self._handle, N * (H // 16))
        if getattr(self, "_gws", None) is None or self._gws.numel() < need or self._gws.device != dev:
            self._gws = None
            self._gws = torch.empty(need, dtype=torch.uint8, device=dev)
        roll = torch.empty((N, 3, 128, 8 * H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_vae_decode_latent_save(self._handle, _rgm.ptr(latent), 1.0 / float(scale_factor), _rgm.ptr(roll),
                                                           N, H, _rgm.ptr(self._gws), need, _rgm.current_stream()))
        self._saved = (N, H, 1.0 / float(scale_factor), need)
        return roll

    @torch.no_grad()
    def decode_latent_vjp(self, d_roll):
        """d_latent (N,4,H,16) = (d roll / d latent)^T d_roll for the last decode_latent_save."""
        if getattr(self, "_saved", None) is None:
            raise _rgm.RgmError("decode_latent_vjp: no decode_latent_save to differentiate")
        N, H, inv_scale, need = self._saved
        _rgm.require_cuda(d_roll)
        dev = d_roll.device
        d_roll = d_roll.detach().to(torch.float32).contiguous()
        assert tuple(d_roll.shape) == (N, 3, 128, 8 * H), f"d_roll must be {(N, 3, 128, 8 * H)}, got {tuple(d_roll.shape)}"
        out = torch.empty((N, 4, H, 16), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_vae_decode_latent_vjp(self._handle, _rgm.ptr(d_roll), inv_scale, _rgm.ptr(out), N, H,
                                                          _rgm.ptr(self._gws), need, _rgm.current_stream()))
        return out

    @torch.no_grad()
    def encode_save(self, x, range_fix=False):
        """x (M,3,128,128) piano-roll tiles in [-1,1] -> moments (M,8,16,16): mean | logvar (klvae_pedal.py:61-68)."""
        _rgm.require_cuda(x)
        dev = self._ensure_native()
        x = x.detach().to(torch.float32).contiguous()
        M = x.shape[0]
        assert tuple(x.shape[1:]) == (3, 128, 128), f"the encoder consumes (M,3,128,128) tiles, got {tuple(x.shape)}"
        out = torch.empty((M, 8, 16, 16), dtype=torch.float32, device=dev)
        ws, need = self._workspace(M, dev)
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_vae_encode(self._handle, _rgm.ptr(x), _rgm.ptr(out), M, _rgm.ptr(ws), need, _rgm.current_stream()))
        if range_fix:
            mean, logvar = torch.chunk(out, 2, dim=1)
            out = torch.concat((torch.sigmoid(mean) * 2 - 1, logvar), dim=1)
        return out

    def encode(self, x, range_fix=False):
        return DiagonalGaussianDistribution(self.encode_save(x, range_fix=range_fix))

    def __del__(self):
        try:
            if self._handle is not None:
                _rgm.lib.rgm_vae_destroy(self._handle)
        except Exception:
            pass

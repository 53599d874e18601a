"""DiTRotary eps-network and DiTRotaryClassifier -- reference API, native MI355X forward.

Mirrors the public surface of the reference's guided_diffusion/dit.py that the sampling path
uses (dit.py:538-634 DiTRotary, :735-831 DiTRotaryClassifier, :969 DiT_models): same constructor
arguments, same state_dict keys (x_embedder.MLP.{0,2}.*, t_embedder.mlp.{0,2}.*,
y_embedder.embedding_table.weight, rotary_emb.freqs, blocks.N.attn.{qkv,proj}.*,
blocks.N.attn.rotary_emb.freqs, blocks.N.mlp.{fc1,fc2}.*, blocks.N.adaLN_modulation.1.*,
final_layer.{linear,adaLN_modulation.1}.*, cls_token, norm*, classifier_head*), same
`model(x, t, y=None)` call.  The modules only OWN parameters (torch = device memory); the forward
is one call into librgm_hip.so (rgm_dit_forward / rgm_dit_classify).  No eager fallback.

Legacy non-rotary variants of the reference (DiT, DiT_classifier, PatchEmbed, ...) are out of
scope and are not registered.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from rgm import native as _rgm
from rgm.native import DitCfg
from rgm.synth import dit_param_shapes


# ---- conditioning ahead of the step (round 6).  The adaLN modulation of a sample depends on (t, y) only, and a sampling loop visits timesteps it
# knows in advance; its 0.9 GB of weights (XL) are the one HBM-bound pass of a forward.  A loop that tells the model which timestep it is at and
# which come next (`cond_hint`, set by gaussian_diffusion around its eps-network calls) gets the modulation rows of the next steps -- for every
# label of the table -- from ONE pass (rgm_dit_cond_rows, up to COND_AHEAD rows) and the forwards gather their rows (rgm_dit_forward_cond).
# Same rows, bit for bit, as the per-forward pass (tests/test_gpu_round6.py); RGM_COND_AHEAD=0 turns it off.
COND_AHEAD = int(os.environ.get("RGM_COND_AHEAD", "32"))
_HINT = None


class cond_hint:
    """with cond_hint((t_now, [t_now, t_next, ...])): model(x, t, y) -- t is th.full(t_now) by the caller's construction (original timesteps)."""

    def __init__(self, hint):
        self.hint = hint

    def __enter__(self):
        global _HINT
        self.prev, _HINT = _HINT, self.hint

    def __exit__(self, *exc):
        global _HINT
        _HINT = self.prev


def _attach(root, dotted, param):
    """Register `param` under a dotted state_dict key, creating bare container modules on the way."""
    *path, leaf = dotted.split(".")
    mod = root
    for name in path:
        nxt = mod._modules.get(name)
        if nxt is None:
            nxt = nn.Module()
            mod.add_module(name, nxt)
        mod = nxt
    mod.register_parameter(leaf, param)


class _NativeDiT(nn.Module):
    """Parameter container + native handle shared by the eps-network and the classifiers."""

    def __init__(self, *, input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio,
                 kind, out_channels=0, n_embed=0, n_out=0, chord=False):
        super().__init__()
        if isinstance(input_size, int):
            input_size = [input_size, input_size]
        assert float(mlp_ratio) == 4.0, "the native blocks implement mlp_ratio=4 (all registered configs)"
        self.input_size = list(input_size)
        self.patch_size, self.in_channels, self.num_heads = patch_size, in_channels, num_heads
        self.hidden_size, self.depth = hidden_size, depth
        self._kind, self._n_embed, self._n_out, self._out_ch = kind, n_embed, n_out, out_channels
        self._arch = dict(depth=depth, hidden=hidden_size, heads=num_heads, patch=patch_size, in_ch=in_channels,
                          out_ch=out_channels, num_classes=n_embed, class_dropout=False,
                          classifier=kind != 0, cls_classes=n_out, chord=chord)
        shared = {}
        for key, shape in dit_param_shapes(**self._arch):
            if key.endswith("rotary_emb.freqs"):          # ONE Parameter, visible under every alias
                if "freqs" not in shared:
                    shared["freqs"] = nn.Parameter(torch.empty(shape), requires_grad=False)
                _attach(self, key, shared["freqs"])
            else:
                _attach(self, key, nn.Parameter(torch.empty(shape)))
        self._handle = None
        self._dirty = True
        self._version = 0     # bumped whenever the parameters are uploaded again (conditioning rows computed ahead are then stale)
        self._ws = None
        self._gws = None      # workspace of the input-gradient calls (saved activations)
        self.register_load_state_dict_post_hook(lambda m, _: setattr(m, "_dirty", True))
        self.reset_parameters()

    # ---- initialisation with the reference's distributions (dit.py:578-606, :780-801)
    def reset_parameters(self):
        D, heads = self.hidden_size, self.num_heads
        rot = int(D // heads * 0.5)
        with torch.no_grad():
            for key, p in self.named_parameters(remove_duplicate=True):
                if key.endswith("freqs"):
                    p.copy_(1.0 / (10000.0 ** (torch.arange(0, rot, 2).float() / rot)))
                elif key == "cls_token":
                    nn.init.normal_(p, std=1e-6)
                elif key.endswith("embedding_table.weight") or key.startswith("t_embedder") and key.endswith("weight"):
                    nn.init.normal_(p, std=0.02)
                elif "adaLN_modulation" in key or key.startswith("final_layer"):
                    nn.init.zeros_(p)                      # adaLN-zero: a fresh eps-network outputs 0
                elif key.startswith("norm"):
                    (nn.init.ones_ if key.endswith("weight") else nn.init.zeros_)(p)
                elif p.dim() == 2:
                    nn.init.xavier_uniform_(p)
                else:
                    nn.init.zeros_(p)
        self._dirty = True

    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    # ---- native side
    def _device(self):
        return next(self.parameters()).device

    def _ensure_native(self, n_tokens):
        dev = self._device()
        if dev.type != "cuda":
            raise _rgm.RgmError("DiTRotary runs only on a HIP device (model.to('cuda')); there is no CPU path "
                                "in the product -- the CPU restatement lives in oracle/ for tests only")
        if self._handle is not None and n_tokens > self._max_tokens:
            _rgm.lib.rgm_dit_destroy(self._handle)
            self._handle = None
        if self._handle is None:
            self._max_tokens = max(n_tokens, 2 * self.input_size[0] * self.input_size[1] // self.patch_size // 2 + 1, 257)
            self._max_tokens = min(self._max_tokens, 288)
            cfg = DitCfg(depth=self.depth, hidden=self.hidden_size, heads=self.num_heads, patch=self.patch_size,
                         in_ch=self.in_channels, out_ch=self._out_ch, width=self.input_size[1],
                         n_embed=self._n_embed, kind=self._kind, n_out=self._n_out, max_tokens=self._max_tokens)
            h = C.c_void_p()
            with torch.cuda.device(dev):
                _rgm.check(_rgm.lib.rgm_dit_create(C.byref(cfg), C.byref(h)))
            self._handle, self._dirty = h, True
        if self._dirty:
            torch.cuda.synchronize(dev)
            with torch.cuda.device(dev):
                for key, p in self.state_dict().items():
                    t = p.detach().to(torch.float32).contiguous()
                    shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                    _rgm.check(_rgm.lib.rgm_dit_set_param(self._handle, key.encode(), _rgm.ptr(t), shape, t.dim()))
            self._dirty = False
            self._version += 1

    def _workspace(self, N, H):
        need = _rgm.lib.rgm_dit_workspace_bytes(self._handle, N, H)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self._device():
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device())
        return self._ws, need

    @staticmethod
    def _as_index(t, dtype):
        if t.is_floating_point():
            if not bool((t == t.round()).all()):
                raise NotImplementedError("fractional timesteps (rescale_timesteps with T != 1000) are not supported")
            t = t.round()
        return t.to(dtype).contiguous()

    def __del__(self):
        try:
            if self._handle is not None:
                _rgm.lib.rgm_dit_destroy(self._handle)
        except Exception:
            pass


class DiTRotary(_NativeDiT):
    """Diffusion eps-network with rotary attention (reference dit.py:538-634)."""

    def __init__(self, input_size=32, patch_size=8, in_channels=3, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, num_classes=9, learn_sigma=True):
        self.learn_sigma = learn_sigma
        self.num_classes = num_classes
        out_ch = in_channels * 2 if learn_sigma else in_channels
        n_embed = (num_classes + int(class_dropout_prob > 0)) if num_classes else 0
        super().__init__(input_size=input_size, patch_size=patch_size, in_channels=in_channels,
                         hidden_size=hidden_size, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                         kind=0, out_channels=out_ch, n_embed=n_embed)
        self.out_channels = out_ch

    def forward(self, x, t, y=None):
        """x (N,C,H,W) f32, t (N,) diffusion timesteps, y (N,) class labels or None -> eps (N,out,H,W)."""
        _rgm.require_cuda(x, t, y)
        N, _, H, W = x.shape
        assert W == self.input_size[1], "pitch axis of the latent is fixed by input_size[1]"
        self._ensure_native(H * W // self.patch_size)
        x = x.detach().to(torch.float32).contiguous()
        yy = self._as_index(y, torch.int32) if (self.num_classes and y is not None) else None
        out = torch.empty((N, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        if _HINT is not None and COND_AHEAD > 0:
            rows, idx = self._rows_ahead(_HINT, yy, N, H, x.device)
            if rows is not None:
                ws, need = self._workspace(N, H)
                with torch.cuda.device(x.device):
                    _rgm.check(_rgm.lib.rgm_dit_forward_cond(self._handle, _rgm.ptr(x), _rgm.ptr(rows), _rgm.ptr(idx), rows.shape[0],
                                                             _rgm.ptr(out), N, H, _rgm.ptr(ws), need, _rgm.current_stream()))
                return out
        t = self._as_index(t, torch.int64)
        ws, need = self._workspace(N, H)
        with torch.cuda.device(x.device):
            _rgm.check(_rgm.lib.rgm_dit_forward(self._handle, _rgm.ptr(x), _rgm.ptr(t), _rgm.ptr(yy), _rgm.ptr(out),
                                                N, H, _rgm.ptr(ws), need, _rgm.current_stream()))
        return out

    def cond_rows(self, ts, ys, H):
        """adaLN modulation rows of the (t, y) pairs (python ints; ys None: no label table) -> (len(ts), (6 depth + 2) hidden) float32."""
        dev = self._device()
        self._ensure_native(H * self.input_size[1] // self.patch_size)
        U = len(ts)
        t = torch.tensor(ts, dtype=torch.int64, device=dev)
        y = torch.tensor(ys, dtype=torch.int32, device=dev) if ys is not None else None
        rows = torch.empty((U, (6 * self.depth + 2) * self.hidden_size), dtype=torch.float32, device=dev)
        ws, need = self._workspace(U, H)
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_dit_cond_rows(self._handle, _rgm.ptr(t), _rgm.ptr(y), U, H, _rgm.ptr(rows), _rgm.ptr(ws), need,
                                                  _rgm.current_stream()))
        return rows

    def _rows_ahead(self, hint, yy, N, H, dev):
        """(rows table, per-sample row index) for a forward at timestep hint[0]; the table holds every label of the next timesteps."""
        t_now, upcoming = hint
        R = self._n_embed if yy is not None else 1              # rows per timestep: one per entry of the label table
        if R > COND_AHEAD:
            return None, None
        key = (self._version, _rgm.lib.rgm_get_gemm_precision(), R, str(dev))
        cache = getattr(self, "_ahead", None)
        if cache is None or cache["key"] != key or t_now not in cache["slot"]:
            steps = [int(v) for v in upcoming[:max(1, COND_AHEAD // R)]]
            if not steps or steps[0] != int(t_now):
                return None, None
            ts = [v for v in steps for _ in range(R)]
            ys = [c for _ in steps for c in range(R)] if yy is not None else None
            cache = {"key": key, "rows": self.cond_rows(ts, ys, H), "slot": {v: k for k, v in enumerate(steps)}, "idx": {}}
            self._ahead = cache
            self.ahead_passes = getattr(self, "ahead_passes", 0) + 1      # (bench.py reports how many weight passes its timed steps paid)
        base = cache["slot"][int(t_now)] * R
        if yy is not None:
            idx = yy + base
        else:
            ik = (base, N)
            if ik not in cache["idx"]:
                cache["idx"][ik] = torch.full((N,), base, dtype=torch.int32, device=dev)
            idx = cache["idx"][ik]
        return cache["rows"], idx

    def _grad_ws(self, N, H, dev):
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_dit_enable_grad(self._handle))
        need = _rgm.lib.rgm_dit_grad_workspace_bytes(self._handle, N, H)
        if self._gws is None or self._gws.numel() < need or self._gws.device != dev:
            self._gws = None
            self._gws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._gws, need

    def vjp_forward(self, x, t, y=None):
        """eps = forward(x, t, y), keeping the activations the input-gradient needs (DPS guidance, reference
        condition_mean :415-465).  The first call keeps W^T copies of the Linear weights on the device."""
        _rgm.require_cuda(x, t, y)
        N, _, H, W = x.shape
        assert W == self.input_size[1]
        self._ensure_native(H * W // self.patch_size)
        x = x.detach().to(torch.float32).contiguous()
        t = self._as_index(t, torch.int64)
        yy = self._as_index(y, torch.int32) if (self.num_classes and y is not None) else None
        eps = torch.empty((N, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        ws, need = self._grad_ws(N, H, x.device)
        with torch.cuda.device(x.device):
            _rgm.check(_rgm.lib.rgm_dit_vjp(self._handle, _rgm.ptr(x), _rgm.ptr(t), _rgm.ptr(yy), None, _rgm.ptr(eps), None,
                                            N, H, _rgm.ptr(ws), need, _rgm.current_stream()))
        self._vjp_shape = (N, H)
        return eps

    def vjp_backward(self, g_eps):
        """grad_x = (d eps / d x)^T g_eps for the forward of the last vjp_forward call."""
        _rgm.require_cuda(g_eps)
        N, H = self._vjp_shape
        g = g_eps.detach().to(torch.float32).contiguous()
        assert g.shape == (N, self.out_channels, H, self.input_size[1])
        grad = torch.empty((N, self.in_channels, H, self.input_size[1]), dtype=torch.float32, device=g.device)
        ws, need = self._grad_ws(N, H, g.device)
        with torch.cuda.device(g.device):
            _rgm.check(_rgm.lib.rgm_dit_vjp(self._handle, None, None, None, _rgm.ptr(g), None, _rgm.ptr(grad), N, H, _rgm.ptr(ws),
                                            need, _rgm.current_stream()))
        return grad

    def vjp(self, x, t, y, g_eps):
        """(eps, (d eps / d x)^T g_eps) -- what th.autograd.grad((model(x, t, y) * g_eps).sum(), x) returns in the reference."""
        eps = self.vjp_forward(x, t, y)
        return eps, self.vjp_backward(g_eps)


class DiTRotaryClassifier(_NativeDiT):
    """Guidance classifier on noisy latents (reference dit.py:735-831); chord=True adds the key head."""

    def __init__(self, input_size=32, patch_size=8, in_channels=3, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, num_classes=9, chord=False):
        self.chord = chord
        self.num_classes = num_classes
        super().__init__(input_size=input_size, patch_size=patch_size, in_channels=in_channels,
                         hidden_size=hidden_size, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                         kind=2 if chord else 1, n_out=num_classes, chord=chord)

    def forward(self, x, t, y=None):
        _rgm.require_cuda(x, t)
        N, _, H, W = x.shape
        self._ensure_native(H * W // self.patch_size + 1)
        x = x.detach().to(torch.float32).contiguous()
        t = self._as_index(t, torch.int64)
        ws, need = self._workspace(N, H)
        if self.chord:
            n_token = H // W
            key = torch.empty((N, 25), dtype=torch.float32, device=x.device)
            out = torch.empty((N, n_token, self.num_classes), dtype=torch.float32, device=x.device)
        else:
            key = None
            out = torch.empty((N, self.num_classes), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _rgm.check(_rgm.lib.rgm_dit_classify(self._handle, _rgm.ptr(x), _rgm.ptr(t), _rgm.ptr(out), _rgm.ptr(key),
                                                 N, H, _rgm.ptr(ws), need, _rgm.current_stream()))
        return (key, out) if self.chord else out

    def value_and_grad(self, x, t, target, loss_kind, scale):
        """(logits, grad_x) with grad_x = scale * d(sum log p)/dx, in one native call (no autograd graph).

        loss_kind "mse": log p = -sum (logits - target)^2, target (N, num_classes) float
        loss_kind "chord_ce": log p = -sum CE(chord_logits, target), target (N, H/W) integer   [chord=True]
        loss_kind "xent": log p = log softmax(logits)[target], target (N,) integer"""
        _rgm.require_cuda(x, t, target)
        N, _, H, W = x.shape
        self._ensure_native(H * W // self.patch_size + 1)
        x = x.detach().to(torch.float32).contiguous()
        t = self._as_index(t, torch.int64)
        if loss_kind == "mse":
            assert not self.chord
            tgt, kind = target.to(torch.float32).contiguous(), 0
            logits = torch.empty((N, self.num_classes), dtype=torch.float32, device=x.device)
        elif loss_kind == "xent":                       # log softmax(logits)[target] on a plain classifier (grad_nn_zt_xentropy)
            assert not self.chord
            tgt, kind = target.reshape(N).to(torch.int64).contiguous(), 1
            logits = torch.empty((N, self.num_classes), dtype=torch.float32, device=x.device)
        elif loss_kind == "chord_ce":
            assert self.chord
            tgt, kind = target.reshape(N, -1).to(torch.int64).contiguous(), 1
            logits = torch.empty((N, H // W, self.num_classes), dtype=torch.float32, device=x.device)
        else:
            raise ValueError(loss_kind)
        grad = torch.empty_like(x)
        need = _rgm.lib.rgm_dit_grad_workspace_bytes(self._handle, N, H)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _rgm.check(_rgm.lib.rgm_dit_cls_value_and_grad(self._handle, _rgm.ptr(x), _rgm.ptr(t), _rgm.ptr(tgt), kind, float(scale),
                                                           _rgm.ptr(logits), _rgm.ptr(grad), N, H, _rgm.ptr(self._ws), need,
                                                           _rgm.current_stream()))
        return logits, grad


def _eps(depth, hidden, heads, patch):
    return lambda **kw: DiTRotary(depth=depth, hidden_size=hidden, patch_size=patch, num_heads=heads, **kw)


def _cls(depth, hidden, heads, patch, chord=False):
    return lambda **kw: DiTRotaryClassifier(depth=depth, hidden_size=hidden, patch_size=patch, num_heads=heads,
                                            chord=chord, **kw)


# Registry names of the reference (dit.py:969-983) for the rotary family.
DiT_models = {
    "DiTRotary_XL_8": _eps(28, 1152, 16, 8),
    "DiTRotary_XL_16": _eps(28, 1152, 16, 16),
    "DiTRotary_B_8": _eps(12, 768, 12, 8),
    "DiTRotary_B_16": _eps(12, 768, 12, 16),
    "DiTRotary-XS/8-cls": _cls(4, 384, 6, 8),
    "DiTRotary-S/8-cls": _cls(12, 384, 6, 8),
    "DiTRotary-S/8-chord-cls": _cls(12, 384, 6, 8, chord=True),
    "DiTRotary-B/8-cls": _cls(12, 768, 12, 8),
}

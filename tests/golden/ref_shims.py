"""Shims that let the *reference* (/root/reference, pure Python) import in this container.

Test infrastructure only -- used by tests/golden/make_golden.py, which runs ONLY in the
build container (the reference never travels to the GPU box).  Nothing here is product code.

Two kinds of stand-ins:
  * inert stubs for packages that are import-time only on the sampling path
    (mpi4py, blobfile, wandb, mido, music21, PIL) -- rank 0 / size 1, no behaviour;
  * numeric restatements of the two un-vendored packages whose arithmetic IS on the path,
    written from the documented behaviour of the pinned versions (environment.yml:242,263):
      - rotary-embedding-torch==0.3.2  RotaryEmbedding.rotate_queries_or_keys
      - timm==0.9.2                    Mlp, use_fused_attn
    These two are the "parity unpinned" slice: the reference itself holds no test for them.
"""
import sys
import types
import math
import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__all__ = [k for k in attrs if not k.startswith("_")]
    sys.modules[name] = m
    return m


class _RotaryEmbedding(nn.Module):
    """rotary-embedding-torch 0.3.2, lang freqs, theta=10000, no xpos, no interpolation."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def forward(self, t):
        freqs = torch.einsum("..., f -> ... f", t.type(self.freqs.dtype), self.freqs)
        return freqs.repeat_interleave(2, dim=-1)  # '... n -> ... (n r)', r=2

    @staticmethod
    def _rotate_half(x):
        x = x.reshape(*x.shape[:-1], -1, 2)
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        seq_len = t.shape[seq_dim]
        freqs = self.forward(torch.arange(seq_len, device=t.device))
        rot = freqs.shape[-1]
        t_mid, t_right = t[..., :rot], t[..., rot:]
        t_mid = t_mid * freqs.cos() + self._rotate_half(t_mid) * freqs.sin()
        return torch.cat((t_mid, t_right), dim=-1)


class _Mlp(nn.Module):
    """timm 0.9.2 layers.Mlp: fc1 -> act -> drop -> Identity -> fc2 -> drop."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class _Unused:  # names imported by dit.py but only touched by non-rotary model classes
    def __init__(self, *a, **k):
        raise RuntimeError("stub: not on the DiTRotary path")


def install():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # numeric stand-ins
    _mod("rotary_embedding_torch", RotaryEmbedding=_RotaryEmbedding)
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.vision_transformer", Attention=_Unused, Mlp=_Mlp)
    _mod("timm.models.vision_transformer_relpos", RelPosAttention=_Unused)
    _mod("timm.layers", Format=_Unused, nchw_to=_Unused, to_2tuple=lambda x: (x, x),
         _assert=lambda c, m="": None, RelPosBias=_Unused,
         use_fused_attn=lambda: hasattr(torch.nn.functional, "scaled_dot_product_attention"))

    # inert stubs
    class _Comm:
        rank = 0
        size = 1
        def Get_rank(self): return 0
        def Get_size(self): return 1
        def bcast(self, x, root=0): return x
        def gather(self, x, root=0): return [x]
        def Barrier(self): pass
    mpi = _mod("mpi4py.MPI", COMM_WORLD=_Comm())
    _mod("mpi4py", MPI=mpi)
    _mod("blobfile", BlobFile=open, exists=lambda p: False)
    _mod("wandb")
    _mod("mido")
    m21 = types.ModuleType("music21")
    m21.__all__ = []
    sys.modules["music21"] = m21
    _mod("pretty_midi", PrettyMIDI=_Unused, Instrument=_Unused, Note=_Unused)
    try:
        import PIL.Image  # noqa: F401  (real Pillow is present in this image)
    except Exception:
        _mod("PIL"); _mod("PIL.Image")

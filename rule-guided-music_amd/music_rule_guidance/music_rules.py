"""Built-in rule programs on a piano roll in [-1, 1] (N, C, 128, T) -- reference API
(music_rule_guidance/music_rules.py:23-94), arithmetic in librgm_hip.so (csrc/rules.hip).

Semantics kept from the reference, including the surprising ones:
  * only channel 0 is read; rows outside the piano range [21, 108] are set to -1 IN the caller's tensor
    (the reference's piano_like writes through a view), note_density also snaps values < -0.95 to -1 there;
  * batch size 1 squeezes the batch dimension of the result;
  * note_density returns [vertical windows..., horizontal windows...] with horizontal / horizontal_scale.
CPU tensors are accepted (the CLI evaluates final rolls from numpy): they are staged to the HIP device,
processed there, and the in-place writes are copied back -- the computation never runs on the CPU.
"""
import torch

from rgm import native as _rgm

VERTICAL_ND_BOUNDS = [1.29, 2.7578125, 3.61, 4.4921875, 5.28125, 6.1171875, 7.22]
VERTICAL_ND_CENTER = [0.56, 2.0239, 3.1839, 4.0511, 4.8867, 5.6992, 6.6686, 7.77]
HORIZONTAL_ND_BOUNDS = [1.8, 2.6, 3.2, 3.6, 4.4, 4.8, 5.8]
HORIZONTAL_ND_CENTER = [1.4, 2.2000, 2.9, 3.4, 4.0, 4.6, 5.3, 6.3]
MIN_PIANO, MAX_PIANO, OFF = 21, 108, -1


def _stage(piano_roll):
    """-> (device tensor the kernels may write, write-back callable)."""
    if piano_roll.dim() != 4 or piano_roll.shape[2] != 128:
        raise ValueError(f"piano roll must be (N, C, 128, T), got {tuple(piano_roll.shape)}")
    if piano_roll.is_cuda and piano_roll.is_contiguous() and piano_roll.dtype == torch.float32:
        return piano_roll, (lambda d: None)
    if not torch.cuda.is_available():
        raise _rgm.RgmError("rule kernels need a HIP device (no CPU fallback in the product path)")
    dev = piano_roll.device if piano_roll.is_cuda else torch.device("cuda", torch.cuda.current_device())
    d = piano_roll.detach().to(device=dev, dtype=torch.float32).contiguous()

    def back(dd):
        with torch.no_grad():
            piano_roll[:, :1].copy_(dd[:, :1].to(piano_roll.device, piano_roll.dtype))
    return d, back


def piano_like(x):
    x[:, :, :MIN_PIANO, :] = OFF
    x[:, :, MAX_PIANO + 1:, :] = OFF
    return x


def total_pitch_class_histogram(piano_roll):
    d, back = _stage(piano_roll)
    N, Cc, _, T = d.shape
    out = torch.empty((N, 12), dtype=torch.float32, device=d.device)
    scratch = torch.empty((N, 128), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _rgm.check(_rgm.lib.rgm_rule_pitch_hist(_rgm.ptr(d), _rgm.ptr(out), _rgm.ptr(scratch), N, Cc, T, _rgm.current_stream()))
    back(d)
    out = out.to(piano_roll.device)
    return out.squeeze(0) if N == 1 else out


def note_density(piano_roll, interval=128, quantize_factor=1, horizontal_scale=5):
    if quantize_factor != 1:
        raise NotImplementedError("quantize_factor != 1 is unused by the sampling configs")
    d, back = _stage(piano_roll)
    N, Cc, _, T = d.shape
    out = torch.empty((N, 2 * (T // interval)), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _rgm.check(_rgm.lib.rgm_rule_note_density(_rgm.ptr(d), _rgm.ptr(out), N, Cc, T, int(interval), float(horizontal_scale),
                                                  _rgm.current_stream()))
    back(d)
    out = out.to(piano_roll.device)
    return out.squeeze() if N == 1 else out


_BOUNDS = {}


def note_density_class(piano_roll, interval=128, quantize_factor=1, horizontal_scale=1):
    nd = note_density(piano_roll, interval=interval, quantize_factor=quantize_factor, horizontal_scale=horizontal_scale)
    dev = nd.device if nd.is_cuda else torch.device("cuda", torch.cuda.current_device())
    key = (str(dev), float(horizontal_scale))
    if key not in _BOUNDS:
        _BOUNDS[key] = (torch.tensor(VERTICAL_ND_BOUNDS, device=dev),
                        torch.tensor(HORIZONTAL_ND_BOUNDS, device=dev) / horizontal_scale)
    vb, hb = _BOUNDS[key]
    half = nd.shape[-1] // 2
    ndd = nd.to(dev)
    out = torch.empty(ndd.shape, dtype=torch.int64, device=dev)
    for lo, bounds in ((0, vb), (half, hb)):
        part = ndd[:, lo:lo + half].contiguous()
        res = torch.empty(part.shape, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _rgm.check(_rgm.lib.rgm_bucketize(_rgm.ptr(part), _rgm.ptr(bounds), bounds.numel(), _rgm.ptr(res), part.numel(),
                                              _rgm.current_stream()))
        out[:, lo:lo + half] = res
    return out.to(piano_roll.device)


_CHORD_BACKEND = None


def register_chord_backend(fn):
    """Install the host-side chord analyser (reference: piano_roll_to_chord.py via music21 + pretty_midi).
    It is symbolic-music analysis on the CPU, not GPU work, and its dependencies are not vendored."""
    global _CHORD_BACKEND
    _CHORD_BACKEND = fn


def get_chords(piano_roll_batch, given_key=None, fs=100, window_size=1.28, return_key=False):
    if _CHORD_BACKEND is None:
        raise ImportError("chord rules need a host plugin (music21 / mido based); install one with "
                          "music_rule_guidance.music_rules.register_chord_backend(fn)")
    return _CHORD_BACKEND(piano_roll_batch, given_key=given_key, fs=fs, window_size=window_size, return_key=return_key)

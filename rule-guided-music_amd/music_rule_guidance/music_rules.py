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
import numpy as np
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


# ---- chord rule (a11): device-side preamble + a host analyser.  The reference's analyser is symbolic-music code on music21
# (piano_roll_to_chord.py:307-359), which is not vendored and cannot be restated: it stays a plug-in with the reference's own
# per-excerpt signature, so `register_chord_backend(piano_roll_to_chords)` with the reference's function is all a user with
# music21 needs.  What IS arithmetic on the roll -- mask, background snap, 0..127 quantisation -- runs on the GPU.
KEY_DICT = {"D major": 0, "g minor": 1, "B- major": 2, "G major": 3, "d minor": 4, "c# minor": 5, "F major": 6, "E- major": 7,
            "e minor": 8, "f# minor": 9, "C major": 10, "F# major": 11, "g# minor": 12, "A major": 13, "a minor": 14,
            "B major": 15, "A- major": 16, "b- minor": 17, "E major": 18, "c minor": 19, "b minor": 20, "e- minor": 21,
            "f minor": 22, "C# major": 23, "no key": 24}      # the chord classifier's key classes (piano_roll_to_chord.py:15-18)
IND2KEY = {v: k for k, v in KEY_DICT.items()}

_CHORD_BACKEND = None
_CHORD_WORKERS = 4          # the reference chunks the batch over a 4-process pool (gaussian_diffusion.py:1365-1371)
_CHORD_POOL = None


def register_chord_backend(fn, workers=4):
    """fn(piano_roll (128,T) int array in [0,127], given_key=None, return_key=False, fs=100., window_size=1.28) ->
    {"chords": LongTensor (T/fs/window_size,), ["key": int, "correlationCoefficient": float]} -- the signature of the reference's
    piano_roll_to_chords (music21).  workers > 1 evaluates the excerpts of a batch in a persistent spawn-context process pool
    (fn must be picklable, i.e. a module-level function); 0/1 = in this process."""
    global _CHORD_BACKEND, _CHORD_WORKERS, _CHORD_POOL
    if _CHORD_POOL is not None:
        _CHORD_POOL.terminate()
        _CHORD_POOL = None
    _CHORD_BACKEND, _CHORD_WORKERS = fn, int(workers)


def chord_quantise(piano_roll_batch):
    """(N,C,128,T) roll -> (N,128,T) uint8 integer roll of channel 0 (get_chords' preamble, music_rules.py:100-110); writes the
    piano_like mask and the < -0.95 -> -1 snap into the caller's roll like the reference."""
    d, back = _stage(piano_roll_batch)
    N, Cc, _, T = d.shape
    q = torch.empty((N, 128, T), dtype=torch.uint8, device=d.device)
    with torch.cuda.device(d.device):
        _rgm.check(_rgm.lib.rgm_rule_chord_quantise(_rgm.ptr(d), _rgm.ptr(q), N, Cc, T, _rgm.current_stream()))
    back(d)
    return q


def _chord_job(args):
    fn, roll, kw = args
    return fn(roll, **kw)


def _chord_pool():
    """the persistent spawn-context worker pool (None: evaluate in the calling thread)"""
    global _CHORD_POOL
    if _CHORD_WORKERS <= 1:
        return None
    if _CHORD_POOL is None:
        import multiprocessing
        _CHORD_POOL = multiprocessing.get_context("spawn").Pool(_CHORD_WORKERS)
    return _CHORD_POOL


def _run_chord_jobs(jobs):
    outs = None
    if _CHORD_WORKERS > 1 and len(jobs) > 1:
        try:
            outs = _chord_pool().map(_chord_job, jobs)
        except (AttributeError, TypeError, ImportError) as e:      # an unpicklable backend (lambda / closure): run it here
            if "pickle" not in str(e).lower() and "local object" not in str(e).lower():
                raise
            outs = None
    if outs is None:
        outs = [_chord_job(j) for j in jobs]
    return outs


def _pack_chords(outs, return_key):
    chords = torch.stack([torch.as_tensor(o["chords"], dtype=torch.long) for o in outs], dim=0)
    if chords.shape[0] == 1:
        chords = chords.squeeze(0)
    if return_key:
        return chords, [o["key"] for o in outs], [o["correlationCoefficient"] for o in outs]
    return chords


# ---- the analyser OVERLAPPED with the GPU.  The reference blocks the step on pool.map (gaussian_diffusion.py:1363-1375);
# here get_chords_async enqueues the device preamble on the caller's stream, copies the uint8 rolls to pinned host memory on a side
# stream and hands the analysis to a driver thread (which waits for THAT copy only, then feeds the worker pool) -- the caller goes on
# enqueueing GPU work (the next chunk's decode, the other rules) and joins the answer where it needs it (ChordFuture.result()).
_CHORD_SIDE = {}            # device index -> side stream of the D2H copies
_CHORD_DRIVER = None        # one driver thread: futures complete in submission order


class ChordFuture:
    """result() -> what get_chords returns for the same roll (chords [, keys, correlation coefficients])."""

    def __init__(self, fut, return_key):
        self._fut, self._return_key = fut, return_key

    def done(self):
        return self._fut.done()

    def result(self):
        return _pack_chords(self._fut.result(), self._return_key)


def get_chords_async(piano_roll_batch, given_key=None, fs=100, window_size=1.28, return_key=False):
    """get_chords without the wait: (N,C,128,T) DEVICE roll -> ChordFuture.  The roll is read (and, like get_chords, written: mask +
    background snap) by a kernel on the current stream; the caller may go on using it on that stream at once."""
    global _CHORD_DRIVER
    if _CHORD_BACKEND is None:
        raise ImportError("chord rules need a host analyser (the reference's is music21-based and not vendored): "
                          "music_rule_guidance.music_rules.register_chord_backend(piano_roll_to_chords)")
    _rgm.require_cuda(piano_roll_batch)
    q = chord_quantise(piano_roll_batch)                                  # (N,128,T) uint8, current stream
    dev = q.device
    cur = torch.cuda.current_stream(dev)
    side = _CHORD_SIDE.get(dev.index)
    if side is None:
        side = _CHORD_SIDE[dev.index] = torch.cuda.Stream(device=dev)
    host = torch.empty(q.shape, dtype=torch.uint8, pin_memory=True)
    ready = torch.cuda.Event()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        host.copy_(q, non_blocking=True)
        ready.record(side)
    q.record_stream(side)                                                 # the allocator must not hand q out again before the copy ran
    kw = dict(given_key=given_key, fs=fs, window_size=window_size, return_key=return_key)
    fn = _CHORD_BACKEND

    def work():
        ready.synchronize()                                               # this copy only -- not the device
        rolls = host.numpy().astype(np.intc)
        return _run_chord_jobs([(fn, rolls[i], kw) for i in range(rolls.shape[0])])
    if _CHORD_DRIVER is None:
        from concurrent.futures import ThreadPoolExecutor
        _CHORD_DRIVER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rgm-chord")
    return ChordFuture(_CHORD_DRIVER.submit(work), return_key)


def get_chords(piano_roll_batch, given_key=None, fs=100, window_size=1.28, return_key=False):
    """FUNC_DICT['chord_progression']: (N,C,128,T) roll -> chords (N, windows) LongTensor [(windows,) when N == 1]
    (+ keys, correlation coefficients with return_key) -- reference music_rules.py:97-130.  Blocking, like the reference; the
    samplers' search step uses get_chords_async."""
    if _CHORD_BACKEND is None:
        raise ImportError("chord rules need a host analyser (the reference's is music21-based and not vendored): "
                          "music_rule_guidance.music_rules.register_chord_backend(piano_roll_to_chords)")
    rolls = chord_quantise(piano_roll_batch).cpu().numpy().astype(np.intc)
    kw = dict(given_key=given_key, fs=fs, window_size=window_size, return_key=return_key)
    return _pack_chords(_run_chord_jobs([(_CHORD_BACKEND, rolls[i], kw) for i in range(rolls.shape[0])]), return_key)

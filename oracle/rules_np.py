"""Rule-program oracle (numpy): pitch histogram, note density (+class), losses.

Test infrastructure (see oracle/__init__.py).  Restates:
  music_rule_guidance/music_rules.py:23-26   piano_like   (IN-PLACE write into the caller's roll)
  music_rule_guidance/music_rules.py:29-43   total_pitch_class_histogram
  music_rule_guidance/music_rules.py:46-83   note_density (quantize_factor=1 only)
  music_rule_guidance/music_rules.py:86-94   note_density_class
  music_rule_guidance/rule_maps.py:5-38      FUNC_DICT / LOSS_DICT (mse_loss_mean, zero_one_loss_mean)
Like the reference, the functions mutate channel 0 of the roll they are given (piano_like mask,
and the < -0.95 background threshold in note_density) -- later rules in the same SCG step see
those writes.  The chord rule (music21) cannot be restated: parity unpinned.
"""
from functools import partial
import numpy as np

F32 = np.float32
MIN_PIANO, MAX_PIANO = 21, 108
VERTICAL_ND_BOUNDS = [1.29, 2.7578125, 3.61, 4.4921875, 5.28125, 6.1171875, 7.22]
HORIZONTAL_ND_BOUNDS = [1.8, 2.6, 3.2, 3.6, 4.4, 4.8, 5.8]


def piano_like(x):
    x[:, :, :MIN_PIANO, :] = -1
    x[:, :, MAX_PIANO + 1:, :] = -1
    return x


def pitch_hist(roll):
    """(N,C,128,T) in [-1,1] -> (N,12)  [(12,) when N==1]."""
    pr = piano_like(roll[:, :1])                      # view: in-place mask
    pr = ((pr + 1) / F32(2.0))[:, 0]
    per_pitch = pr.sum(-1, dtype=F32)                 # (N,128)
    padded = np.concatenate((per_pitch, np.zeros((pr.shape[0], 4), dtype=F32)), axis=-1)
    hist = padded.reshape(-1, 11, 12).transpose(0, 2, 1).sum(-1, dtype=F32)
    hist = hist / (hist.sum(-1, keepdims=True) + F32(1e-12))
    return hist[0] if hist.shape[0] == 1 else hist


def pitch_hist_logp_grad(roll, target, scale=1.0):
    """rule_x0_mse_dummy on pitch_hist (condition_functions.py:122-126 over music_rules.py:29-43) with its gradient written out:
    log p = -scale * sum_c (hist_c - target_c)^2 (N,), and d log p / d roll (N,C,128,T).  hist = h / (sum h + 1e-12) with h_c the
    sum of (x+1)/2 over the piano rows of class c, so d/d x = 0.5 * (u_c - sum_k u_k hist_k) / (sum h + 1e-12), u = -2 scale (hist -
    target), on channel 0's piano rows [21,108] and 0 elsewhere (piano_like overwrites the other rows with a constant)."""
    pr = piano_like(roll[:, :1])
    h128 = ((pr + 1) / F32(2.0))[:, 0].sum(-1, dtype=np.float64)                    # (N,128)
    h = np.concatenate((h128, np.zeros((h128.shape[0], 4))), axis=-1).reshape(-1, 11, 12).sum(1)
    d = h.sum(-1, keepdims=True) + 1e-12
    hist = h / d
    e = hist - np.asarray(target, dtype=np.float64).reshape(hist.shape)
    logp = -scale * (e * e).sum(-1)
    u = -2.0 * scale * e
    dh = (u - (u * hist).sum(-1, keepdims=True)) / d                                 # (N,12)
    grad = np.zeros(roll.shape, dtype=np.float64)
    p = np.arange(MIN_PIANO, MAX_PIANO + 1)
    grad[:, 0, p, :] = 0.5 * dh[:, p % 12][:, :, None]
    return logp.astype(F32), grad.astype(F32)


def chord_quantise(roll):
    """get_chords' preamble (music_rules.py:97-110): piano_like + (< -0.95 -> -1) written into channel 0, then the integer roll
    clamp((x+1)/2*127, 0, 127) truncated that the music21 analyser is given -> (N,128,T) int32."""
    pr = piano_like(roll[:, :1])
    pr[pr < F32(-0.95)] = -1.0
    v = np.clip((pr + F32(1)) / F32(2) * F32(127), 0, 127)
    return v[:, 0].astype(np.intc)


def note_density(roll, interval=128, horizontal_scale=5):
    """(N,C,128,T) -> (N, 2*T/interval): [vertical windows..., horizontal windows...]."""
    pr = piano_like(roll[:, :1])
    n, T = pr.shape[0], pr.shape[-1]
    pr[pr < F32(-0.95)] = -1.0                        # in place, like the reference
    b = (pr + 1) / F32(2.0)
    b = (b >= F32(1e-2)).astype(F32)
    vert_col = b.sum(axis=2, dtype=F32)               # (N,1,T)
    bp = np.pad(b, ((0, 0), (0, 0), (0, 0), (1, 1)))
    d = np.diff(bp, axis=-1)
    d[d < 0] = 0
    hor_col = (d.sum(axis=2)[:, :, :-1] != 0).astype(F32)
    vert = vert_col.reshape(n, 1, -1, interval).mean(-1, dtype=F32)
    hor = hor_col.reshape(n, 1, -1, interval).sum(-1, dtype=F32) / F32(horizontal_scale)
    nd = np.concatenate((vert, hor), axis=-1)
    return nd.squeeze() if n == 1 else nd[:, 0]


def note_density_class(roll, interval=128, horizontal_scale=1):
    nd = note_density(roll, interval=interval, horizontal_scale=horizontal_scale)
    half = nd.shape[-1] // 2
    vb = np.asarray(VERTICAL_ND_BOUNDS, dtype=F32)
    hb = np.asarray(HORIZONTAL_ND_BOUNDS, dtype=F32) / F32(horizontal_scale)
    # torch.bucketize(right=False): first index i with bounds[i] >= v
    return np.concatenate((np.searchsorted(vb, nd[:, :half], side="left"),
                           np.searchsorted(hb, nd[:, half:], side="left")), axis=-1).astype(np.int64)


def mse_loss_mean(gen, y):
    return ((gen.astype(F32) - y.astype(F32)) ** 2).mean(-1, dtype=F32)


def zero_one_loss_mean(gen, y):
    return (y != gen).astype(F32).mean(-1, dtype=F32)


FUNC_DICT = {
    "pitch_hist": pitch_hist,
    "note_density": note_density,
    "note_density_hr_1": partial(note_density, horizontal_scale=1.0),
    "note_density_hr_2": partial(note_density, horizontal_scale=2.0),
    "note_density_class": note_density_class,
    "note_density_pixel": partial(note_density, interval=16),
}
LOSS_DICT = {
    "pitch_hist": mse_loss_mean,
    "note_density": mse_loss_mean,
    "note_density_hr_1": mse_loss_mean,
    "note_density_hr_2": mse_loss_mean,
    "note_density_class": zero_one_loss_mean,
    "note_density_pixel": mse_loss_mean,
}

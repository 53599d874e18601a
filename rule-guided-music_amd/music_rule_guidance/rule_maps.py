"""Plugin surface of rule guidance -- same dict API as the reference (music_rule_guidance/rule_maps.py:5-38):
FUNC_DICT[name](piano_roll (N,C,128,T) in [-1,1]) -> (N,K) and LOSS_DICT[name](gen (N,K), target (N,K)) -> (N,).
Add your own entries (plain torch functions work); the built-in names dispatch to HIP kernels."""
from functools import partial

import torch

from rgm import native as _rgm
from . import music_rules

FUNC_DICT = {
    "pitch_hist": music_rules.total_pitch_class_histogram,
    "note_density": music_rules.note_density,
    "note_density_hr_1": partial(music_rules.note_density, horizontal_scale=1.),
    "note_density_hr_2": partial(music_rules.note_density, horizontal_scale=2.),
    "note_density_class": music_rules.note_density_class,
    "chord_progression": music_rules.get_chords,
    "note_density_pixel": partial(music_rules.note_density, interval=16),     # lower time resolution
    "chord_progression_pixel": partial(music_rules.get_chords, fs=12.5),
}


def _row_loss(gen_rule, y_, zero_one):
    if gen_rule.shape != y_.shape:
        gen_rule, y_ = torch.broadcast_tensors(gen_rule, y_)
    dev = gen_rule.device if gen_rule.is_cuda else (y_.device if y_.is_cuda else None)
    if dev is None:
        if not torch.cuda.is_available():
            raise _rgm.RgmError("rule losses need a HIP device (no CPU fallback in the product path)")
        dev = torch.device("cuda", torch.cuda.current_device())
    a = gen_rule.to(device=dev, dtype=torch.float32).contiguous()
    b = y_.to(device=dev, dtype=torch.float32).contiguous()
    K = a.shape[-1]
    rows = a.numel() // K
    out = torch.empty(a.shape[:-1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _rgm.check(_rgm.lib.rgm_row_loss(_rgm.ptr(a), _rgm.ptr(b), _rgm.ptr(out), rows, K, int(zero_one), _rgm.current_stream()))
    return out.to(gen_rule.device)


def mse_loss_mean(gen_rule, y_):
    return _row_loss(gen_rule, y_, False)


def zero_one_loss_mean(gen_rule, y_):
    return _row_loss(gen_rule, y_, True)


def zero_one_loss_sum(gen_rule, y_):
    return _row_loss(gen_rule, y_, True) * gen_rule.shape[-1]


# used by SCG to select the best candidate, and reported as the loss
LOSS_DICT = {
    "pitch_hist": mse_loss_mean,
    "note_density": mse_loss_mean,
    "note_density_hr_1": mse_loss_mean,
    "note_density_hr_2": mse_loss_mean,
    "note_density_class": zero_one_loss_mean,
    "chord_progression": zero_one_loss_mean,
    "note_density_pixel": mse_loss_mean,
    "chord_progression_pixel": zero_one_loss_mean,
}

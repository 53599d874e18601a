"""Config loading, final decode to the uint8 piano roll and rule-loss reporting -- reference API
(guided_diffusion/midi_util.py:26-64, :96-124).  MIDI writing / plotting of the reference need mido +
the vendored pretty_midi fork (pure I/O, out of scope); save_piano_roll_midi stores .npy rolls instead and
hands over to an optional user hook."""
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch
import yaml

from music_rule_guidance.rule_maps import FUNC_DICT, LOSS_DICT
from music_rule_guidance.music_rules import IND2KEY, KEY_DICT, MAX_PIANO, MIN_PIANO  # noqa: F401  (re-exported like the reference)
from rgm import native as _rgm


def dict_to_obj(d):
    """Nested dicts -> attribute namespaces (lists keep their element types)."""
    if isinstance(d, list):
        return [dict_to_obj(x) if isinstance(x, dict) else x for x in d]
    if isinstance(d, dict):
        return SimpleNamespace(**{k: dict_to_obj(v) for k, v in d.items()})
    return d


def load_config(filename):
    with open(filename, "r") as f:
        return dict_to_obj(yaml.safe_load(f))


# The final decode is the one place where a float becomes an INTEGER (truncating cast + background threshold): a roll value within the
# decoder's arithmetic error of a boundary flips.  It runs once per sample, outside the step loop, so it always takes the exact-fp32
# MFMA decoder (v_mfma_f32_32x32x2_f32: the reference's own arithmetic up to summation order), whatever the sampling loop ran in --
# measured on the reference's 50-step golden: 250 -> the fp32 level of mismatching entries of 786 432 (docs/rounds/r06.md).
# RGM_FINAL_DECODE_EXACT=0 / exact=False: the process-wide arithmetic instead (A/B runs).
FINAL_DECODE_EXACT = os.environ.get("RGM_FINAL_DECODE_EXACT", "1") != "0"


@torch.no_grad()
def decode_sample_for_midi(sample, embed_model=None, scale_factor=1., threshold=-0.95, exact=None):
    """Latent batch -> uint8 piano roll (B, 128, T, 3) in [0, 127].

    With the native AutoencoderKL the whole chain (1/scale_factor, square gather, decoder, background
    threshold, (x+1)*63.5 clamp, truncating cast, layout change) is one decode call; other embed models are
    driven through tensor views and only the final quantisation runs in the HIP kernel.
    exact (default FINAL_DECODE_EXACT = True): run this decode in exact-fp32 arithmetic (reference midi_util.py:42-64 is fp32)."""
    if exact is None:
        exact = FINAL_DECODE_EXACT
    if embed_model is not None and hasattr(embed_model, "decode_latent") and sample.shape[-2] >= sample.shape[-1]:
        with _rgm.gemm_precision_scope("fp32" if exact else None):
            return embed_model.decode_latent(sample, scale_factor, want_u8=True, threshold=threshold, want_float=False)
    sample = sample / scale_factor
    if embed_model is not None:
        h, w = sample.shape[-2:]
        if h > w:
            sample = sample.permute(0, 1, 3, 2)
        n_lat = sample.shape[-1] // sample.shape[-2]
        if h >= w:
            sample = torch.cat(torch.chunk(sample, n_lat, dim=-1), dim=0)
        sample = embed_model.decode(sample)
        if h >= w:
            sample = torch.cat(torch.chunk(sample, n_lat, dim=0), dim=-1)
    _rgm.require_cuda(sample)
    roll = sample.to(torch.float32).contiguous()
    B, Cc, P, T = roll.shape
    assert Cc == 3 and P == 128, "quantise kernel expects (B,3,128,T)"
    out = torch.empty((B, 128, T, 3), dtype=torch.uint8, device=roll.device)
    with torch.cuda.device(roll.device):
        _rgm.check(_rgm.lib.rgm_quantise_roll(_rgm.ptr(roll), _rgm.ptr(out), B, T, float(threshold), _rgm.current_stream()))
    return out


# note-density class boundaries / centres used to shift an extracted density by whole classes (reference midi_util.py:20-23)
VERTICAL_ND_BOUNDS = [1.29, 2.7578125, 3.61, 4.4921875, 5.28125, 6.1171875, 7.22]
VERTICAL_ND_CENTER = [0.56, 2.0239, 3.1839, 4.0511, 4.8867, 5.6992, 6.6686, 7.77]
HORIZONTAL_ND_BOUNDS = [1.8, 2.6, 3.2, 3.6, 4.4, 4.8, 5.8]
HORIZONTAL_ND_CENTER = [1.4, 2.2000, 2.9, 3.4, 4.0, 4.6, 5.3, 6.3]

_MIDI_WRITER = None
_MIDI_READER = None


def register_midi_reader(fn):
    """fn(path, fs) -> piano roll (3,128,T) with values in [0,127] (the reference: pretty_midi + get_full_piano_roll,
    scripts/edit.py:170-171) -- plug in a MIDI reader for the editing CLI."""
    global _MIDI_READER
    _MIDI_READER = fn


def read_midi_piano_roll(path, fs=100):
    """MIDI file -> (3,128,T) [velocity | onset | pedal] roll.  Default: the built-in SMF reader + get_full_piano_roll's logic
    (music_rule_guidance.piano_roll_to_chord; all three channels pinned to the reference's pretty_midi fork, tests/golden/midi_rolls.npz);
    register_midi_reader replaces it, e.g. with the fork itself."""
    if _MIDI_READER is not None:
        return np.asarray(_MIDI_READER(path, fs), dtype=np.float32)
    from music_rule_guidance.piano_roll_to_chord import SimpleMIDI, midi_to_full_piano_roll
    return midi_to_full_piano_roll(SimpleMIDI(path), fs=fs)


def register_midi_writer(fn):
    """fn(piano_roll_uint8 (C,128,T), path, fs) -- plug in a MIDI writer (the reference uses its pretty_midi fork)."""
    global _MIDI_WRITER
    _MIDI_WRITER = fn


def save_piano_roll_midi(sample, save_dir, fs=100, y=None, save_piano_roll=False, save_ind=0):
    """sample: (B, 3, 128, T) uint8 array (or (B,128,T) / (B,2,128,T)).  Writes sample_<i>[_y_<label>].midi per sample (names
    and event extraction as in the reference :67-93, incl. the forced onsets in the first column) plus the raw roll as
    .npy; register_midi_writer replaces the built-in SMF writer.  save_piano_roll (PNG plots) is not provided."""
    from music_rule_guidance.piano_roll_to_chord import piano_roll_to_pretty_midi
    os.makedirs(save_dir, exist_ok=True)
    for i in range(sample.shape[0]):
        stem = f"sample_{i + save_ind}" + (f"_y_{int(y[i])}" if y is not None else "")
        np.save(os.path.join(save_dir, stem + ".npy"), sample[i])
        if _MIDI_WRITER is not None:
            _MIDI_WRITER(sample[i], os.path.join(save_dir, stem + ".midi"), fs)
            continue
        cur = np.array(sample[i])
        if cur.ndim == 3 and cur.shape[0] == 3:            # a note sounding in the first column starts there
            cur[1, np.nonzero(cur[0, :, 0])[0], 0] = 127
        piano_roll_to_pretty_midi(cur.astype(np.float32), fs=fs).write(os.path.join(save_dir, stem + ".midi"))


def eval_rule_loss(generated_samples, target_rules):
    """DataFrame with <rule>.target_rule / .gen_rule / .loss columns, one row per sample."""
    results = {}
    B = generated_samples.shape[0]
    for name, target in target_rules.items():
        tl = target.tolist()
        results[name + ".target_rule"] = [tl] if B == 1 else tl
        target = target.to(generated_samples.device)
        if "chord" in name:
            gen, key, corr = FUNC_DICT[name](generated_samples, return_key=True)
            results[name + ".key_str"] = [IND2KEY.get(k, k) for k in key]
            results[name + ".key_corr"] = corr
        else:
            gen = FUNC_DICT[name](generated_samples)
        loss = LOSS_DICT[name](gen, target)
        gl = gen.tolist()
        results[name + ".gen_rule"] = [gl] if B == 1 else gl
        results[name + ".loss"] = loss.tolist() if loss.dim() else [loss.item()]
    return pd.DataFrame(results)

"""Replacement-based editing CLI -- the MI355X-native counterpart of the reference's scripts/edit.py.

Reference flow (scripts/edit.py:45-323): load eps-network + VAE (+ classifiers), read a source piano roll, ENCODE it to the
latent (`_encode`, gaussian_diffusion.py:1382-1395), build a mask that frees latent rows [l_start, l_end), noise the
ground truth to `noise_level` and run the reverse chain from there with `edit_kwargs`: every step's x0 estimate is
overwritten by the ground truth outside the editable rows (p_mean_variance :293-298), guidance / SCG only look at the
editable excerpt.  Rule targets are absolute values from the YAML, a shift of the rule extracted from the source
(integers), or the source's own rule (Null).

Sources: `edit.source` may be a `.npy` piano roll ((3,128,T) float in [-1,1] or (128,T,3) uint8), a MIDI file (built-in
SMF reader; guided_diffusion.midi_util.register_midi_reader swaps in another, e.g. pretty_midi), or `synthetic`
(a seeded sparse roll, for smoke runs).  `source: dataset` needs the reference's data loader and is not provided.
"""
import argparse
import os
import sys
from functools import partial

import numpy as np
import pandas as pd
import torch as th
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
sys.path.insert(0, _HERE)

from guided_diffusion import dist_util, logger, midi_util                                   # noqa: E402
from guided_diffusion.gaussian_diffusion import _encode, _extract_rule                      # noqa: E402
from guided_diffusion.midi_util import (HORIZONTAL_ND_BOUNDS, HORIZONTAL_ND_CENTER, VERTICAL_ND_BOUNDS,  # noqa: E402
                                        VERTICAL_ND_CENTER)
import sample_rule as _sr                                                                                         # noqa: E402
from sample_rule import build_pipeline, setup_chord_backend, create_argparser as _sample_argparser              # noqa: E402


def synthetic_roll(seed, T):
    """A seeded sparse piano roll (1,3,128,T) in [-1,1]: background at -1, held notes with onsets."""
    rng = np.random.RandomState(seed)
    r = -np.ones((1, 3, 128, T), dtype=np.float32)
    for _ in range(60 * T // 1024 + 8):
        p, s, L = rng.randint(30, 100), rng.randint(0, T - 8), rng.randint(8, 120)
        r[0, 0, p, s:s + L] = rng.uniform(-0.2, 1.0)
        r[0, 1, p, s] = 1.0
    return th.from_numpy(r)


def load_source(source, T, fs, device, allow_synthetic=False):
    """-> ground-truth roll (1,3,128,T) float32 in [-1,1], right-padded with background (reference :169-174)."""
    if source == "dataset":
        # the reference draws a test-set excerpt here (edit.py :140-168); the dataset loader is out of scope
        if not allow_synthetic:
            raise RuntimeError("edit.source 'dataset' needs the reference's data loader (out of scope): give a .npy roll or a MIDI "
                               "file as edit.source, or opt in to a seeded synthetic roll with --allow_synthetic_source True")
        logger.log("WARNING: --allow_synthetic_source: edit.source 'dataset' replaced by the seeded synthetic roll")
        source = "synthetic"
    if source == "synthetic":
        gt = synthetic_roll(0, T)
    elif str(source).endswith(".npy"):
        a = np.load(source)
        if a.dtype == np.uint8:                         # (128,T,3) as written by decode_sample_for_midi
            a = a.transpose(2, 0, 1).astype(np.float32) / 63.5 - 1
        gt = th.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None]
    else:
        gt = th.from_numpy(midi_util.read_midi_piano_roll(source, fs)).float()[None] / 63.5 - 1
    gt = gt[..., :T]
    return F.pad(gt, (0, T - gt.shape[3]), "constant", -1).to(device)


def edit_target_rules(target_rules, gt_partial, batch_size, device):
    """Targets for the editable excerpt (reference :187-246) -> ({rule: (B,K) tensor}, {rule: original rule})."""
    out, orig = {}, {}
    for rule_name, val in target_rules.items():
        if "horizontal" in rule_name:
            continue
        if "vertical" in rule_name:
            hr_nd = target_rules[rule_name.replace("vertical", "horizontal")]
            if "_hr_" in rule_name:
                tag = rule_name.split("_hr_")[-1]
                horizontal_scale, rule_name = int(tag), f"note_density_hr_{tag}"
            else:
                horizontal_scale, rule_name = 5, "note_density"
            orig_rule = _extract_rule(rule_name, gt_partial.clone())
            if orig_rule.dim() == 1:
                orig_rule = orig_rule.reshape(1, -1)
            if isinstance(val, int) or val is None:      # shift the extracted density by whole classes
                vt_bounds = th.tensor(VERTICAL_ND_BOUNDS, device=device)
                hr_bounds = th.tensor(HORIZONTAL_ND_BOUNDS, device=device) / horizontal_scale
                vt_center = th.tensor(VERTICAL_ND_CENTER, device=device)
                hr_center = th.tensor(HORIZONTAL_ND_CENTER, device=device) / horizontal_scale
                if isinstance(val, int):
                    vertical_rand, horizontal_rand = val, hr_nd
                else:
                    vertical_rand = th.randint(-1, 2, size=(orig_rule.shape[0], 1), device=device)
                    horizontal_rand = th.randint(-1, 2, size=(orig_rule.shape[0], 1), device=device)
                half = orig_rule.shape[-1] // 2
                vt_cls = th.bucketize(orig_rule[:, :half].contiguous(), vt_bounds) + vertical_rand
                hr_cls = th.bucketize(orig_rule[:, half:].contiguous(), hr_bounds) + horizontal_rand
                target = th.concat((vt_center[vt_cls.clamp_(min=0, max=7)], hr_center[hr_cls.clamp_(min=0, max=7)]), dim=-1)
            else:
                target = th.tensor(list(val) + [x / horizontal_scale for x in hr_nd], device=device)
        elif "pitch" in rule_name and val is not None:
            orig_rule = _extract_rule(rule_name, gt_partial.clone())
            v = th.tensor(val, device=device)
            target = v / (th.sum(v) + 1e-12)
        else:
            orig_rule = _extract_rule(rule_name, gt_partial.clone())
            target = th.tensor(val, device=device) if val is not None else orig_rule
        out[rule_name] = target.reshape(1, -1).float().repeat(batch_size, 1) if target.dim() == 1 or target.shape[0] == 1 \
            else target.float()
        orig[rule_name] = orig_rule
    return out, orig


def main(argv=None):
    args = create_argparser().parse_args(argv)
    root = "edit_demo/"
    tail = args.config_path.split("configs/")[-1] if "configs/" in args.config_path else os.path.basename(args.config_path)
    args.dir = root + os.path.splitext(tail)[0] + f"_cls_{args.class_label}"
    from rgm import native as _native
    _native.set_gemm_precision(args.gemm_precision)
    comm = dist_util.setup_dist(port=args.port)
    config = setup_chord_backend(args, midi_util.load_config(args.config_path))
    substituted = vars(config.edit).get("source", "synthetic") == "dataset" and args.allow_synthetic_source
    if _sr.DROPPED_RULES:
        args.dir += "_nochord"
    if substituted:
        args.dir += "_synthsrc"
    logger.configure(args=args, comm=comm)
    if config.sampling.use_ddim:
        args.timestep_respacing = config.sampling.timestep_respacing
    device = dist_util.dev()
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if args.vae is None:
        raise ValueError("editing needs the VAE (the source is encoded to the latent)")
    P = build_pipeline(args, config, device)
    diffusion, embed_model, gen_shape = P.diffusion, P.embed_model, P.gen_shape
    classes = None
    if args.class_cond:
        classes = th.ones(size=(args.batch_size,), device=device, dtype=th.int) * args.class_label

    save_dir = logger.get_dir()
    save_dir_gt = os.path.join(save_dir, "gt")
    os.makedirs(os.path.expanduser(save_dir_gt), exist_ok=True)
    sample_fn = partial(diffusion.ddim_sample_loop, eta=1.) if config.sampling.use_ddim else diffusion.p_sample_loop

    edit_kwargs = dict(vars(config.edit))
    edit_kwargs["l_start_pix"], edit_kwargs["l_end_pix"] = edit_kwargs["l_start"] * 8, edit_kwargs["l_end"] * 8
    gt = load_source(edit_kwargs.get("source", "synthetic"), gen_shape[2] * 8, args.fs, device, allow_synthetic=args.allow_synthetic_source)
    if args.save_files and rank0:
        _sr.write_run_metadata(save_dir, args, {"source": edit_kwargs.get("source", "synthetic"), "source_substituted_by_synthetic": bool(substituted)})
    gt_latent = _encode(gt, embed_model, scale_factor=args.scale_factor)
    mask = th.ones_like(gt_latent)
    mask[:, :, edit_kwargs["l_start"]:edit_kwargs["l_end"], :] = 0.
    edit_kwargs["gt"], edit_kwargs["mask"] = gt_latent, mask

    logger.log("sampling...")
    gt_partial = gt[:, :, :, edit_kwargs["l_start_pix"]:edit_kwargs["l_end_pix"]]
    rules, orig = edit_target_rules(vars(config.target_rules), gt_partial, args.batch_size, device)
    model_kwargs = {"rule": rules}
    if classes is not None:
        model_kwargs["y"] = classes
    use_scg = bool(getattr(config.guidance, "scg", getattr(config.guidance, "beam", False)))

    all_results = pd.DataFrame()
    count_samples = 0
    while count_samples < args.num_samples:
        sample = sample_fn(
            P.model_fn, gen_shape, clip_denoised=args.clip_denoised, model_kwargs=model_kwargs, device=device, cond_fn=P.cond_fn,
            embed_model=embed_model if config.guidance.vae else None, scale_factor=args.scale_factor,
            guidance_kwargs=config.guidance, scg_kwargs=vars(config.scg) if use_scg else None, edit_kwargs=edit_kwargs,
            t_end=config.sampling.t_end, record=args.record, progress=args.progress)
        sample = midi_util.decode_sample_for_midi(sample, embed_model=embed_model, scale_factor=args.scale_factor, threshold=-0.95)
        arr = sample.cpu().numpy().transpose(0, 3, 1, 2)
        arr_gt = ((gt + 1) * 63.5).clamp(0, 127).to(th.uint8).cpu().numpy()
        if args.save_files and rank0:
            lab = classes.cpu().numpy() if classes is not None else None
            midi_util.save_piano_roll_midi(arr, save_dir, args.fs, y=lab, save_ind=count_samples)
            midi_util.save_piano_roll_midi(arr_gt, save_dir_gt, args.fs, y=lab[:1] if lab is not None else None, save_ind=count_samples)
        generated = th.from_numpy(arr.astype(np.float32)) / 63.5 - 1
        generated = generated[:, :, :, edit_kwargs["l_start_pix"]:edit_kwargs["l_end_pix"]]   # only the edited excerpt is scored
        results = midi_util.eval_rule_loss(generated, model_kwargs["rule"])
        for name, o in orig.items():
            results[name + ".orig_rule"] = [o.reshape(-1).cpu().tolist()] * len(results)
        all_results = pd.concat([all_results, results], ignore_index=True)
        if args.save_files and rank0:
            all_results.to_csv(os.path.join(save_dir, "results.csv"), index=False)
        count_samples += args.batch_size

    if args.save_files and rank0:
        loss_cols = [c for c in all_results.columns if ".loss" in c]
        stats = pd.DataFrame([{"Attr": c, "Mean": all_results[c].mean(), "Std": all_results[c].std()} for c in loss_cols],
                             columns=["Attr", "Mean", "Std"])
        stats.to_csv(os.path.join(save_dir, "summary.csv"))
        print(stats)
    logger.log("sampling complete")
    return all_results, sample


def create_argparser():
    parser = _sample_argparser()
    parser.add_argument("--allow_synthetic_source", default=False, type=lambda v: str(v).lower() in ("yes", "true", "t", "y", "1"),
                        help="edit.source 'dataset' needs the reference's data loader; True substitutes a seeded synthetic roll "
                             "(marked in the output directory name and run_metadata.json)")
    return parser


if __name__ == "__main__":
    main()

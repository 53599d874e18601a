"""Rule-guided sampling CLI -- same flags, YAML schema and outputs as the reference's scripts/sample_rule.py
(:40-314): builds the eps-network, the diffusion, the VAE and the guidance classifiers, turns the YAML's
target rules into model_kwargs, samples batches with DDPM / stochastic DDIM (+ classifier guidance and/or SCG,
+ DiffCollage for long sequences), decodes to uint8 piano rolls and reports per-rule losses in results.csv /
summary.csv.  Run from this directory's parent (rule-guided-music_amd/) exactly like the reference:

    python scripts/sample_rule.py --config_path scripts/configs/cond_demo/demo2.yml --model DiTRotary_XL_8 \
        --model_path <ema.pt> --vae_path <vae.ckpt> --image_size 128 16 --in_channels 4 --scale_factor 1.2465 \
        --class_cond True --num_classes 3 --class_label 1 --batch_size 4 --num_samples 20

Multi-GPU: `torchrun --nproc-per-node 8 scripts/sample_rule.py ...` shards the SCG candidates over the ranks
(RCCL all-gather of the rule log-probs); rank 0 writes the outputs.  --synthetic_weights skips the checkpoint
files (no network here) and uses rgm.synth weights: for smoke runs and benchmarking only.
"""
import argparse
import os
import sys
from functools import partial

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch as th  # noqa: E402

from guided_diffusion import dist_util, midi_util, logger  # noqa: E402
from guided_diffusion.dit import DiT_models  # noqa: E402
from guided_diffusion.script_util import (  # noqa: E402
    NUM_CLASSES, model_and_diffusion_defaults, create_diffusion, add_dict_to_argparser, args_to_dict)
from guided_diffusion.gaussian_diffusion import _extract_rule  # noqa: E402
from guided_diffusion.condition_functions import model_fn, dc_model_fn, composite_nn_zt, composite_rule  # noqa: E402
from load_utils import load_model  # noqa: E402
import diff_collage as dc  # noqa: E402


KEEP_FLOAT_ROLLS = None   # tests only: a list -> main() appends every batch's FLOAT decoded roll (B,3,128,T) next to the uint8 one
NOISE_FN = None     # tests only: callable(shape, device) -> tensor installed as diffusion.noise_fn (teacher-forced parity runs)


DROPPED_RULES = []    # what setup_chord_backend removed under --skip_chord_rules (run_metadata.json, output directory suffix)


def setup_chord_backend(args, config):
    """Chord rules (`chord_progression*`) need a host analyser with the signature of the reference's piano_roll_to_chords
    (music21; not installable here).  --chord_backend module:function registers one (music_rules.register_chord_backend).
    Without one a config that asks for chord conditioning is an ERROR -- a run that silently lacks the requested rule would write
    tables that look like a fully conditioned one.  The explicit opt-in `--skip_chord_rules True` removes the chord entries from
    target_rules, from the SCG weights and from the cond_fn lists instead; what was removed is logged, recorded in
    run_metadata.json and marked in the output directory name (`_nochord`)."""
    from music_rule_guidance import music_rules
    del DROPPED_RULES[:]
    if getattr(args, "chord_backend", ""):
        import importlib
        mod, _, fn = args.chord_backend.partition(":")
        music_rules.register_chord_backend(getattr(importlib.import_module(mod), fn), workers=int(args.chord_workers))
        return config
    if music_rules._CHORD_BACKEND is not None:
        return config
    tr = vars(config.target_rules)
    cf = getattr(config.guidance, "cond_fn", None)
    wanted = [f"target_rules.{k}" for k in tr if "chord" in k]
    if cf is not None:
        wanted += [f"cond_fn.{r}" for r in cf.rule_names if "chord" in r]
    if not wanted:
        return config
    if not getattr(args, "skip_chord_rules", False):
        raise RuntimeError("this config conditions on chord rules (" + ", ".join(wanted) + ") but no chord analyser is registered: "
                           "pass --chord_backend module:function (signature of the reference's piano_roll_to_chords), or opt in to "
                           "running WITHOUT them with --skip_chord_rules True")
    dropped = []
    for k in [k for k in tr if "chord" in k]:
        tr.pop(k)
        dropped.append(f"target_rules.{k}")
    if getattr(config, "scg", None) is not None:
        for k in [k for k in vars(config.scg) if "chord" in k]:
            vars(config.scg).pop(k)
    if cf is not None and any("chord" in r for r in cf.rule_names):
        keep = [i for i, r in enumerate(cf.rule_names) if "chord" not in r]
        dropped += [f"cond_fn.{cf.rule_names[i]}" for i in range(len(cf.rule_names)) if i not in keep]
        for name in ("fns", "classifier_scales", "rule_names"):
            setattr(cf, name, [getattr(cf, name)[i] for i in keep])
        cc = getattr(cf, "classifiers", None)
        if cc is not None:
            for name in ("names", "paths", "num_classes"):
                if hasattr(cc, name):
                    v = getattr(cc, name)
                    setattr(cc, name, [v[i] for i in keep if i < len(v)])
        if not keep:
            config.guidance.cond_fn = None
            config.guidance.nn = False
    DROPPED_RULES.extend(dropped)
    logger.log("WARNING: --skip_chord_rules: running WITHOUT " + ", ".join(dropped))
    return config


def write_run_metadata(save_dir, args, extra=None):
    """run_metadata.json next to results.csv: what this run was NOT conditioned on / substituted (empty lists for a faithful run)."""
    import json
    meta = {"config_path": args.config_path, "dropped_rules": list(DROPPED_RULES), "synthetic_weights": bool(args.synthetic_weights),
            "targets_npz": getattr(args, "targets_npz", "") or None}
    meta.update(extra or {})
    with open(os.path.join(save_dir, "run_metadata.json"), "w") as f:
        json.dump(meta, f, indent=1)


def targets_from_npz(path, target_rules, batch_size, device):
    """`target_rules: Null` (reference :147-168: the targets are the rules of a dataset batch).  The dataset loader is out of scope;
    the batch itself is not: --targets_npz gives either `gt`, the ground-truth rolls ((B,3,128,T) float32 in [-1,1], what the
    reference's load_data yields) from which every rule is extracted with _extract_rule exactly like the reference does, or one
    array per rule ((B,K) or (K,)).  vertical_nd / horizontal_nd collapse into the fused note_density rule as in the reference."""
    names = list(target_rules)
    if "vertical_nd" in names:
        names = [k for k in names if k not in ("vertical_nd", "horizontal_nd")] + ["note_density"]
    z = np.load(path)
    out = {}
    if "gt" in z.files:
        gt = th.from_numpy(np.ascontiguousarray(z["gt"], dtype=np.float32)).to(device)
        if gt.shape[0] < batch_size:
            raise ValueError(f"{path}: gt holds {gt.shape[0]} rolls, batch_size is {batch_size}")
        gt = gt[:batch_size].contiguous()
        with th.no_grad():
            for k in names:
                out[k] = _extract_rule(k, gt)
        return out
    for k in names:
        if k not in z.files:
            raise KeyError(f"{path}: no array for rule '{k}' (has {z.files}); give `gt` or one array per rule")
        v = th.from_numpy(np.ascontiguousarray(z[k])).to(device)
        v = v.float() if v.is_floating_point() else v
        if v.dim() == 1:
            v = v.repeat(batch_size, 1)
        if v.shape[0] < batch_size:
            raise ValueError(f"{path}: '{k}' holds {v.shape[0]} rows, batch_size is {batch_size}")
        out[k] = v[:batch_size].contiguous()
    return out


def output_dir_for(config_path, class_label):
    """cond_demo/<config path below cond_table/ or cond_demo/>_cls_<label>   (reference :42-46)."""
    root = "cond_demo/"
    key = "cond_table/" if "cond_table/" in config_path else root
    return root + os.path.splitext(config_path.split(key)[-1])[0] + f"_cls_{class_label}"


def build_target_rules(target_rules, batch_size, device):
    """YAML target lists -> {rule: (B,K) tensor}.  vertical_nd + horizontal_nd fuse into one note_density
    target [vertical..., horizontal/scale...] (scale 5, or N for the *_hr_N variants); pitch_hist is
    normalised to sum 1 (reference :170-193)."""
    rules = dict(target_rules)
    for key in list(rules):
        if "vertical_nd" in key:
            if "_hr_" in key:
                tag = key.split("_hr_")[-1]
                scale, hkey, out = int(tag), f"horizontal_nd_hr_{tag}", f"note_density_hr_{tag}"
            else:
                scale, hkey, out = 5, "horizontal_nd", "note_density"
            rules[out] = list(rules[key]) + [v / scale for v in rules[hkey]]
            rules.pop(key)
            rules.pop(hkey)
            break
    out = {}
    for key, val in rules.items():
        v = th.tensor(val, device=device)
        if key == "pitch_hist":
            v = v / (th.sum(v) + 1e-12)
        out[key] = v.repeat(batch_size, 1)
    return out


def build_pipeline(args, config, device):
    """eps-network, diffusion, VAE, guidance cond_fn and the (possibly DiffCollage-wrapped) model_fn, as both CLIs
    (sample_rule.py :52-137, edit.py :55-138 of the reference) set them up.  Returns a SimpleNamespace."""
    from types import SimpleNamespace
    logger.log("creating model and diffusion...")
    model = DiT_models[args.model](input_size=args.image_size, in_channels=args.in_channels,
                                   num_classes=args.num_classes, learn_sigma=args.learn_sigma)
    diffusion = create_diffusion(**args_to_dict(args, ["learn_sigma", "diffusion_steps", "noise_schedule", "timestep_respacing",
                                                       "use_kl", "predict_xstart", "rescale_timesteps", "rescale_learned_sigmas"]))
    if args.synthetic_weights:
        from rgm import synth
        arch = dict(depth=model.depth, hidden=model.hidden_size, heads=model.num_heads, patch=model.patch_size,
                    in_ch=args.in_channels, out_ch=model.out_channels,
                    num_classes=model._n_embed, class_dropout=False)
        model.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / model.hidden_size ** 0.5, device=device, **arch))
    else:
        model.load_state_dict(dist_util.load_state_dict(args.model_path, map_location="cpu"), strict=False)
    model.to(device)
    if args.use_fp16:
        raise NotImplementedError("the reference's DiTRotary has no convert_to_fp16 either; sampling is fp32")
    model.eval()

    embed_model = None
    if args.vae is not None:
        embed_model = load_model(args.vae, None if args.synthetic_weights else args.vae_path)
        if args.synthetic_weights:
            from rgm import synth
            embed_model.load_state_dict(synth.vae_state_dict(2, device=device, encoder=True))
        embed_model.to(device)
        embed_model.eval()

    cond_fn_config = config.guidance.cond_fn
    classifiers = []
    if config.guidance.nn:
        logger.log("loading classifier...")
        cc = cond_fn_config.classifiers
        # the reference loads one classifier per entry of `names` (:88-104; cond_demo/demo3.yml lists three cond_fns but two
        # names, so two run); its pixel-space ablation configs carry no names at all: inferred from the cond_fn here
        names = getattr(cc, "names", None) or ["DiTRotary-S/8-chord-cls" if "chord" in f else "DiTRotary-S/8-cls" for f in cond_fn_config.fns]
        for i, name in enumerate(names):
            clf = DiT_models[name](input_size=args.image_size, in_channels=args.in_channels, num_classes=cc.num_classes[i])
            if args.synthetic_weights:
                from rgm import synth
                arch = dict(depth=clf.depth, hidden=clf.hidden_size, heads=clf.num_heads, patch=clf.patch_size,
                            in_ch=args.in_channels, classifier=True, cls_classes=cc.num_classes[i], chord=clf.chord)
                clf.load_state_dict(synth.dit_state_dict(3 + i, device=device, **arch))
            else:
                clf.load_state_dict(dist_util.load_state_dict(cc.paths[i], map_location="cpu"))
            clf.to(device)
            clf.eval()
            classifiers.append(clf)

    if cond_fn_config is None:
        cond_fn_used = None
    elif config.guidance.nn:
        cond_fn_used = partial(composite_nn_zt, fns=cond_fn_config.fns, classifier_scales=cond_fn_config.classifier_scales,
                               classifiers=classifiers, rule_names=cond_fn_config.rule_names)
    else:
        cond_fn_used = partial(composite_rule, fns=cond_fn_config.fns, classifier_scales=cond_fn_config.classifier_scales,
                               rule_names=cond_fn_config.rule_names)

    if config.sampling.diff_collage:
        def eps_fn(x, t, y=None):          # the backbone takes (4, time, pitch)
            return model(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
        img_shape = (args.in_channels, args.image_size[1], args.image_size[0])            # 4 x 16 x 128
        if config.dc.type == "circle":       # the circle needs one more window than the line
            worker = dc.CondIndCircle(img_shape, eps_fn, config.dc.num_img + 1, overlap_size=config.dc.overlap_size)
        else:
            worker = dc.CondIndSimple(img_shape, eps_fn, config.dc.num_img, overlap_size=config.dc.overlap_size)
        gen_shape = (args.batch_size, worker.shape[0], worker.shape[2], worker.shape[1])
        model_fn_used = partial(dc_model_fn, model=worker.eps_scalar_t_fn, num_classes=args.num_classes,
                                class_cond=args.class_cond, cfg=args.cfg, w=args.w)
    else:
        gen_shape = (args.batch_size, args.in_channels, args.image_size[0], args.image_size[1])
        model_fn_used = partial(model_fn, model=model, num_classes=args.num_classes, class_cond=args.class_cond,
                                cfg=args.cfg, w=args.w)

    return SimpleNamespace(model=model, diffusion=diffusion, embed_model=embed_model, cond_fn=cond_fn_used,
                           model_fn=model_fn_used, gen_shape=gen_shape, classifiers=classifiers)


def main(argv=None):
    args = create_argparser().parse_args(argv)
    args.dir = output_dir_for(args.config_path, args.class_label)
    from rgm import native as _native
    _native.set_gemm_precision(args.gemm_precision)      # "bf16x3_presplit" / "bf16x3" (fast, fp32-grade) or "fp32" (exact fp32 MFMA)
    comm = dist_util.setup_dist(port=args.port)
    config = setup_chord_backend(args, midi_util.load_config(args.config_path))
    if DROPPED_RULES:
        args.dir += "_nochord"
    logger.configure(args=args, comm=comm)
    if config.sampling.use_ddim:
        args.timestep_respacing = config.sampling.timestep_respacing
    device = dist_util.dev()
    rank0 = int(os.environ.get("RANK", "0")) == 0

    P = build_pipeline(args, config, device)
    diffusion, embed_model, cond_fn_used, model_fn_used, gen_shape = P.diffusion, P.embed_model, P.cond_fn, P.model_fn, P.gen_shape
    diffusion.noise_fn = NOISE_FN

    target_rules = vars(config.target_rules)
    if any(v is None for v in list(target_rules.values())[:1]):
        if not args.targets_npz:
            raise NotImplementedError("target rules 'Null' are the rules of a dataset batch in the reference (:147-168); the dataset "
                                      "loader is out of scope -- pass that batch with --targets_npz (`gt` rolls or one array per "
                                      "rule), or give the targets in the YAML")
        model_kwargs = {"rule": targets_from_npz(args.targets_npz, target_rules, args.batch_size, device)}
    else:
        model_kwargs = {"rule": build_target_rules(target_rules, args.batch_size, device)}
    classes = None
    if args.class_cond:
        classes = th.ones(size=(args.batch_size,), device=device, dtype=th.int) * args.class_label
        model_kwargs["y"] = classes

    save_dir = logger.get_dir()
    os.makedirs(os.path.expanduser(save_dir), exist_ok=True)
    if args.save_files and rank0:
        write_run_metadata(save_dir, args)
    sample_fn = partial(diffusion.ddim_sample_loop, eta=1.) if config.sampling.use_ddim else diffusion.p_sample_loop
    use_scg = bool(getattr(config.guidance, "scg", getattr(config.guidance, "beam", False)))

    logger.log("sampling...")
    count_samples = 0
    all_results = pd.DataFrame()
    while count_samples < args.num_samples:
        sample = sample_fn(
            model_fn_used, gen_shape, clip_denoised=args.clip_denoised, model_kwargs=model_kwargs, device=device,
            cond_fn=cond_fn_used, embed_model=embed_model if config.guidance.vae else None, scale_factor=args.scale_factor,
            guidance_kwargs=config.guidance, scg_kwargs=vars(config.scg) if use_scg else None,
            t_end=config.sampling.t_end, record=args.record, progress=args.progress)
        if KEEP_FLOAT_ROLLS is not None:        # parity tests: the float roll explains every uint8 difference (boundary adjacency)
            from guided_diffusion.gaussian_diffusion import _decode
            with _native.gemm_precision_scope("fp32" if midi_util.FINAL_DECODE_EXACT else None):   # the final decode's own arithmetic
                KEEP_FLOAT_ROLLS.append(_decode(sample, embed_model, scale_factor=args.scale_factor).float().cpu().numpy())
        sample = midi_util.decode_sample_for_midi(sample, embed_model=embed_model, scale_factor=args.scale_factor, threshold=-0.95)
        arr = sample.cpu().numpy().transpose(0, 3, 1, 2)                                   # (B, 3, 128, T) uint8
        if args.save_files and rank0:
            midi_util.save_piano_roll_midi(arr, save_dir, args.fs, y=classes.cpu().numpy() if classes is not None else None,
                                           save_ind=count_samples)
        generated = th.from_numpy(arr.astype(np.float32)) / 63.5 - 1
        results = midi_util.eval_rule_loss(generated, model_kwargs["rule"])
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
    if args.record and rank0:
        import pickle
        for name in ("log_probs", "loss_std", "loss_range", "each_loss"):
            with open(os.path.join(save_dir, name + ".pkl"), "wb") as f:
                pickle.dump(dict(getattr(diffusion, name)) if name == "each_loss" else getattr(diffusion, name), f)
    logger.log("sampling complete")
    return all_results


def create_argparser():
    defaults = dict(
        project="music-sampling", dir="", data_dir="", config_path="", model="DiTRotary_XL_8", model_path="",
        vae="kl/f8-all-onset", vae_path="taming-transformers/checkpoints/all_onset/epoch_14.ckpt",
        clip_denoised=False, num_samples=128, batch_size=16, scale_factor=1., fs=100, num_classes=0, class_label=1,
        cfg=False, w=4., classifier_scale=1.0, record=False, save_files=True, training=False, deterministic=False,
        port=None,
        # additions of this implementation
        synthetic_weights=False, progress=True, gemm_precision="bf16x3_presplit", chord_backend="", chord_workers=4,
        skip_chord_rules=False, targets_npz="",
    )
    defaults.update(model_and_diffusion_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


if __name__ == "__main__":
    main()

"""Unguided / classifier-free-guided sampling, data-parallel over the GPUs of a node -- the reference's scripts/cfg_sample.py
(:26-127) on the native kernels.

Every rank draws its own batches (class-conditional, the null label, or classifier-free guidance (1+w) eps(x,y) - w eps(x,null)
evaluated as ONE forward of 2B rows: condition_functions.model_fn), decodes them to uint8 piano rolls on the device, and the
rolls -- 0.4 MB per sample -- are all-gathered (RCCL; the only collective) so that rank 0 writes the MIDI / .npy files.
Flags and defaults are the reference's (:130-157); additions: --synthetic_weights (no checkpoints offline), --seed,
--gemm_precision, --progress.  Ranks use different Philox streams (seed + rank): unlike the sharded SCG of sample_rule.py,
where all ranks must share one stream, here they must NOT produce the same samples.

    torchrun --nnodes=1 --nproc-per-node 8 scripts/cfg_sample.py --model_path ... --cfg True --w 4. --class_label 1
"""
import argparse
import os
import sys

import numpy as np
import torch as th
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial  # noqa: E402

from guided_diffusion import dist_util, logger, midi_util  # noqa: E402
from guided_diffusion.condition_functions import model_fn  # noqa: E402
from guided_diffusion.dit import DiT_models  # noqa: E402
from guided_diffusion.gaussian_diffusion import PhiloxNoise  # noqa: E402
from guided_diffusion.script_util import add_dict_to_argparser, args_to_dict, create_diffusion, model_and_diffusion_defaults  # noqa: E402
from load_utils import load_model  # noqa: E402


def main(argv=None):
    args = create_argparser().parse_args(argv)
    from rgm import native as _native
    _native.set_gemm_precision(args.gemm_precision)
    comm = dist_util.setup_dist(port=args.port)
    logger.configure(args=args, comm=comm)
    device = dist_util.dev()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0

    logger.log("creating model and diffusion...")
    model = DiT_models[args.model](input_size=args.image_size, in_channels=args.in_channels, num_classes=args.num_classes,
                                   learn_sigma=args.learn_sigma)
    diffusion = create_diffusion(**args_to_dict(args, ["learn_sigma", "diffusion_steps", "noise_schedule", "timestep_respacing",
                                                       "use_kl", "predict_xstart", "rescale_timesteps", "rescale_learned_sigmas"]))
    diffusion.batch_shard = diffusion.scg_shard = False     # data-parallel sampler: every rank draws its OWN batches (no shared chain to shard)
    if args.synthetic_weights:
        from rgm import synth
        arch = dict(depth=model.depth, hidden=model.hidden_size, heads=model.num_heads, patch=model.patch_size,
                    in_ch=args.in_channels, out_ch=model.out_channels, num_classes=model._n_embed, class_dropout=False)
        model.load_state_dict(synth.dit_state_dict(1, final_std=0.3 / model.hidden_size ** 0.5, device=device, **arch))
    else:
        model.load_state_dict(dist_util.load_state_dict(args.model_path, map_location="cpu"), strict=False)
    model.to(device)
    if args.use_fp16:
        raise NotImplementedError("the reference's DiTRotary has no convert_to_fp16 either; sampling is fp32")
    model.eval()
    eps_fn = partial(model_fn, model=model, num_classes=args.num_classes, class_cond=args.class_cond, cfg=args.cfg, w=args.w)

    embed_model = load_model(args.embed_model_name, None if args.synthetic_weights else args.embed_model_ckpt)
    if args.synthetic_weights:
        from rgm import synth
        embed_model.load_state_dict(synth.vae_state_dict(2, device=device), strict=False)
    embed_model.to(device)
    embed_model.eval()
    diffusion.noise = PhiloxNoise(seed=int(args.seed) + rank)

    logger.log("sampling...")
    save_dir = os.path.join(logger.get_dir(), f"gen_cls_{args.class_label}{args.save_name}")
    os.makedirs(os.path.expanduser(save_dir), exist_ok=True)
    shape = (args.batch_size, args.in_channels, args.image_size[0], args.image_size[1])
    sample_fn = diffusion.p_sample_loop if not args.use_ddim else diffusion.ddim_sample_loop
    rolls, labels = [], []
    while len(rolls) * args.batch_size < args.num_samples:
        model_kwargs = {}
        classes = None
        if args.class_cond:
            classes = th.ones(size=(args.batch_size,), device=device, dtype=th.int) * args.class_label   # one class per run
            model_kwargs["y"] = classes
        sample = sample_fn(eps_fn, shape, clip_denoised=args.clip_denoised, model_kwargs=model_kwargs, device=device,
                           progress=args.progress)
        u8 = midi_util.decode_sample_for_midi(sample, embed_model=embed_model, scale_factor=args.scale_factor, threshold=-0.95)
        gathered, gl = gather_batch(u8, classes, world)
        rolls.extend(g.cpu().numpy() for g in gathered)
        labels.extend(g.cpu().numpy() for g in gl)
        logger.log(f"created {len(rolls) * args.batch_size} samples")

    arr, label_arr = assemble(rolls, labels, args.num_samples)
    if rank == 0:
        midi_util.save_piano_roll_midi(arr, save_dir, args.fs, y=label_arr)
    if world > 1:
        dist.barrier()
    logger.log("sampling complete")
    return arr


def gather_batch(u8, classes, world):
    """One batch of every rank, in rank order (reference scripts/cfg_sample.py:102-109: all_gather of the uint8 rolls -- 0.4 MB per
    sample -- and of the labels): -> ([rolls of rank 0, rank 1, ...], [labels ...]); one rank: the batch itself."""
    if world > 1:
        gathered = [th.zeros_like(u8) for _ in range(world)]
        dist.all_gather(gathered, u8.contiguous())
    else:
        gathered = [u8]
    gl = []
    if classes is not None:
        if world > 1:
            gl = [th.zeros_like(classes) for _ in range(world)]
            dist.all_gather(gl, classes)
        else:
            gl = [classes]
    return gathered, gl


def assemble(rolls, labels, num_samples):
    """[(B,128,T,C) uint8 per rank and round] -> the first num_samples rolls as (n,C,128,T) (reference :111-117), labels alike."""
    arr = np.concatenate(rolls, axis=0)                                    # (n, 128, T, C) uint8
    arr = arr.squeeze(axis=-1) if arr.shape[-1] == 1 else arr.transpose(0, 3, 1, 2)
    arr = arr[:num_samples]
    label_arr = np.concatenate(labels, axis=0)[:num_samples] if labels else None
    return arr, label_arr


def create_argparser():
    defaults = dict(
        project="music-sampling", save_name="", dir="", model="DiTRotary_XL_8", embed_model_name="kl/f8-all-onset",
        embed_model_ckpt="taming-transformers/checkpoints/all_onset/epoch_14.ckpt", clip_denoised=False, num_samples=128,
        batch_size=16, use_ddim=False, model_path="", scale_factor=1., fs=100, num_classes=3, class_label=1, cfg=False, w=4.,
        training=False, port=None,
        # additions of this implementation
        synthetic_weights=False, progress=True, gemm_precision="bf16x3_presplit", seed=0,
    )
    defaults.update(model_and_diffusion_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


if __name__ == "__main__":
    main()

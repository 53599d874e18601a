"""Defaults, factories and argparse helpers of the sampling CLI -- reference API
(guided_diffusion/script_util.py:13-26, :74-126, :462-531).  UNet / super-res / classifier-UNet
factories of the reference are out of scope."""
import argparse

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps

NUM_CLASSES = 3   # number of datasets (composer groups)


def diffusion_defaults():
    return dict(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="",
                use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def model_and_diffusion_defaults():
    """Flag defaults of the piano-roll scripts (values identical to the reference's dict)."""
    res = dict(image_size=128, in_channels=1, num_channels=128, num_res_blocks=2, num_heads=4,
               num_heads_upsample=-1, num_head_channels=-1, attention_resolutions="32,16,8", channel_mult="",
               dropout=0.0, class_cond=False, use_checkpoint=False, use_scale_shift_norm=True,
               resblock_updown=False, use_fp16=False, use_new_attention_order=False)
    res.update(diffusion_defaults())
    return res


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing if timestep_respacing else [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
        model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def create_diffusion(learn_sigma, diffusion_steps, noise_schedule, timestep_respacing, use_kl, predict_xstart,
                     rescale_timesteps, rescale_learned_sigmas):
    return create_gaussian_diffusion(steps=diffusion_steps, learn_sigma=learn_sigma, noise_schedule=noise_schedule,
                                     use_kl=use_kl, predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
                                     rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=timestep_respacing)


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")


def add_dict_to_argparser(parser, default_dict):
    """One --flag per key; bools via str2bool, None as str, image_size takes one or more values."""
    for k, v in default_dict.items():
        typ = str if v is None else (str2bool if isinstance(v, bool) else type(v))
        extra = {"nargs": "+"} if k == "image_size" else {}
        parser.add_argument(f"--{k}", default=v, type=typ, **extra)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}

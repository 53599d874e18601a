"""Drop-in `guided_diffusion` package for the rule-guided sampling hot path on MI355X.

Same import paths and call signatures as the reference's guided_diffusion/  for:
dit.DiT_models, gaussian_diffusion.GaussianDiffusion, respace.SpacedDiffusion,
script_util.create_diffusion, condition_functions.*, midi_util.{load_config,decode_sample_for_midi,
eval_rule_loss}.  Arithmetic runs in librgm_hip.so (hand-written HIP, gfx950).
"""

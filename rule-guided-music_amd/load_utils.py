"""load_model(name, ckpt) -- reference API (load_utils.py:20-25) without OmegaConf / Lightning:
reads taming-transformers/configs/pr/<name>.yaml with PyYAML (same keys: model.params.{embed_dim, ddconfig}),
builds the native AutoencoderKL and restores the Lightning checkpoint's ["state_dict"]."""
import os

import yaml

from taming.models.klvae_pedal import AutoencoderKL

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_model(name, ckpt):
    cfg_path = f"taming-transformers/configs/pr/{name}.yaml"
    if not os.path.exists(cfg_path):
        cfg_path = os.path.join(_HERE, cfg_path)
    with open(cfg_path) as f:
        params = yaml.safe_load(f)["model"]["params"]
    model = AutoencoderKL(ddconfig=params.get("ddconfig"), embed_dim=params.get("embed_dim", 4))
    if ckpt is not None:
        model.init_from_ckpt(ckpt)
    model.eval()
    return model

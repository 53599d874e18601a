"""Timestep re-spacing ("ddim50", "250", ...) -- reference API (guided_diffusion/respace.py:7-128).

space_timesteps picks which of the original T steps are kept; SpacedDiffusion rebuilds the betas of the
shortened chain (beta_i = 1 - abar_i / abar_prev_kept) and maps the chain's indices back to original
timesteps before the model / cond_fn sees them.  Host-side only; the map lives on the device as one
cached int64 tensor per device instead of being re-created on every call.
"""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Set of kept original timesteps.  "ddimN": fixed integer stride giving exactly N steps;
    otherwise comma-separated (or list of) per-section counts with rounded fractional strides."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                kept = range(0, num_timesteps, stride)
                if len(kept) == want:
                    return set(kept)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion chain that visits only `use_timesteps` of a base chain."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        abar = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64))
        self.timestep_map = [i for i in range(len(abar)) if i in self.use_timesteps]
        prev, betas = 1.0, []
        for i in self.timestep_map:
            betas.append(1 - abar[i] / prev)
            prev = abar[i]
        kwargs["betas"] = np.array(betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def condition_mean(self, cond_fn, *args, **kwargs):
        return super().condition_mean(self._wrap_model(cond_fn), *args, **kwargs)

    def condition_score(self, cond_fn, *args, **kwargs):
        return super().condition_score(self._wrap_model(cond_fn), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t        # scaling (if any) happens inside the wrapped model


class _WrappedModel:
    """Callable that re-maps chain indices to original timesteps before calling `model`."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._maps = {}

    def map_ts(self, ts):
        """chain index -> the timestep the wrapped model is called with"""
        key = (str(ts.device), ts.dtype)
        if key not in self._maps:
            self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = self._maps[key][ts]
        return new_ts.float() * (1000.0 / self.original_num_steps) if self.rescale_timesteps else new_ts

    def __call__(self, x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        if key not in self._maps:
            self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = self._maps[key][ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)

"""SimpleWork: (shape, eps function) holder -- reference API (diff_collage/generic_sampler.py:17-20)."""


class SimpleWork:
    def __init__(self, shape, eps_scalar_t_fn):
        self.shape = shape
        self.eps_scalar_t_fn = eps_scalar_t_fn

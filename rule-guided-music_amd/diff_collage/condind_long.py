"""CondIndSimple: eps of a long sequence from overlapping 128-wide windows under a conditional-independence
factorisation -- reference API (diff_collage/condind_long.py:8-51).

eps_long = fold_sum_i eps(window_i)  -  fold_sum_{i<n-1} eps(right half of window_i)
Two eps-network batches per call (B*n full windows at T=256 tokens, B*n half windows at T=128); the
split and the subtract-and-fold are one HIP kernel each."""
import torch as th

from .generic_sampler import SimpleWork
from .w_img import split_windows, merge_windows


class CondIndSimple(SimpleWork):
    circle = False

    def __init__(self, shape, eps_scalar_t_fn, num_img, overlap_size=32):
        c, h, w = shape
        assert overlap_size == w // 2
        self.overlap_size = overlap_size
        self.num_img = num_img
        super().__init__((c, h, self._final_width(w, num_img)), self.get_eps_t_fn(eps_scalar_t_fn))

    def _final_width(self, w, n):
        return w * n - self.overlap_size * (n - 1)

    def loss(self, x):
        a, b = x[:-1], x[1:]
        return th.sum((a[..., -self.overlap_size:] - b[..., :self.overlap_size]).abs() ** 2, dim=(1, 2, 3))

    def get_eps_t_fn(self, eps_scalar_t_fn):
        n, ov = self.num_img, self.overlap_size

        def eps_t_fn(long_x, scalar_t, y=None):
            xs, halves = split_windows(long_x, n, ov, want_halves=True)       # circular reads cover the circle variant
            yy = None if y is None else y.repeat_interleave(n)
            tt = scalar_t.repeat_interleave(n)
            full_eps = eps_scalar_t_fn(xs, tt, y=yy)                           # (B*n, c, h, 128)
            half_eps = eps_scalar_t_fn(halves, tt, y=yy)                       # (B*n, c, h, overlap)
            return merge_windows(full_eps, half_eps, ov, n, circle=self.circle, is_avg=False)
        return eps_t_fn

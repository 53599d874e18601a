"""CondIndSimple: eps of a long sequence from overlapping 128-wide windows under a conditional-independence
factorisation -- reference API (diff_collage/condind_long.py:8-51).

eps_long = fold_sum_i eps(window_i)  -  fold_sum_{i<n-1} eps(right half of window_i)
Two eps-network batches per call (B*n full windows at T=256 tokens, B*(n-1) half windows at T=128 -- the
reference also evaluates the last window's half and discards it); the split and the subtract-and-fold are one HIP kernel each."""
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

        def sharded(long_x, scalar_t, y, R, r):
            """the 7 + 6 window forwards of ONE replicated call shared out over R ranks (rgm/batch_shard.py WINDOW_SHARD): window i of
            the list [full 0..B n-1 | halves of windows 0..n-2 of every sample] goes to rank i % R; one all-reduce completes them"""
            from rgm import batch_shard
            xs, halves = split_windows(long_x, n, ov, want_halves=True)
            Bn = xs.shape[0]
            yy = None if y is None else y.repeat_interleave(n)
            tt = scalar_t.repeat_interleave(n)
            keep = th.arange(Bn, device=long_x.device).view(-1, n)[:, :n - 1].reshape(-1)
            rf, rh = batch_shard.window_share(Bn, keep.numel(), R, r)        # halves continue the numbering behind the full windows
            mine_f = th.tensor(list(rf), dtype=th.long, device=long_x.device)
            mine_h = keep[th.tensor(list(rh), dtype=th.long, device=long_x.device)]
            buf = th.zeros(Bn * xs[0].numel() + Bn * halves[0].numel(), dtype=th.float32, device=long_x.device)
            full_eps = buf[:Bn * xs[0].numel()].view(xs.shape)
            half_eps = buf[Bn * xs[0].numel():].view(halves.shape)
            # the zero-filled buffers assume eps has its input's shape (a learn_sigma network returns 2C channels: the caller keeps such
            # models replicated -- gaussian_diffusion._search_partition); checked here so that a mismatch fails at its cause
            if mine_f.numel():
                e = eps_scalar_t_fn(xs[mine_f].contiguous(), tt[mine_f].contiguous(), y=None if yy is None else yy[mine_f].contiguous()).float()
                assert e.shape[1:] == xs.shape[1:], f"window sharding needs eps of the input's shape, got {tuple(e.shape)} for {tuple(xs.shape)}"
                full_eps[mine_f] = e
            if mine_h.numel():
                e = eps_scalar_t_fn(halves[mine_h].contiguous(), tt[mine_h].contiguous(), y=None if yy is None else yy[mine_h].contiguous()).float()
                assert e.shape[1:] == halves.shape[1:], f"window sharding needs eps of the input's shape, got {tuple(e.shape)} for {tuple(halves.shape)}"
                half_eps[mine_h] = e
            batch_shard.reduce_windows(buf)
            return merge_windows(full_eps, half_eps, ov, n, circle=False, is_avg=False)

        def eps_t_fn(long_x, scalar_t, y=None):
            if not self.circle and n > 1:
                from rgm import batch_shard
                R, r = batch_shard.window_world()
                if R > 1:
                    return sharded(long_x, scalar_t, y, R, r)
            xs, halves = split_windows(long_x, n, ov, want_halves=True)       # circular reads cover the circle variant
            yy = None if y is None else y.repeat_interleave(n)
            tt = scalar_t.repeat_interleave(n)
            full_eps = eps_scalar_t_fn(xs, tt, y=yy)                           # (B*n, c, h, 128)
            if self.circle or n == 1:
                half_eps = eps_scalar_t_fn(halves, tt, y=yy)                   # (B*n, c, h, overlap)
            else:
                # linear collage: the reference evaluates the right half of EVERY window and then sets the last one to zero
                # (condind_long.py:37-44 `half_eps[-1] = 0`) -- that forward is skipped here: n - 1 half windows per sample
                keep = th.arange(long_x.shape[0] * n, device=long_x.device).view(-1, n)[:, :n - 1].reshape(-1)
                part = eps_scalar_t_fn(halves[keep].contiguous(), tt[keep].contiguous(), y=None if yy is None else yy[keep].contiguous())
                half_eps = th.zeros((halves.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
                half_eps[keep] = part
            return merge_windows(full_eps, half_eps, ov, n, circle=self.circle, is_avg=False)
        return eps_t_fn

"""DiffCollage oracle (numpy): window split / fold-merge and the conditional-independence eps.

Test infrastructure (see oracle/__init__.py).  Restates:
  diff_collage/w_img.py:8-24     split_wimg   (unfold into n windows of width 128)
  diff_collage/w_img.py:26-48    avg_merge_wimg (fold-sum, optional divide by coverage count)
  diff_collage/condind_long.py:24-51     CondIndSimple.get_eps_t_fn
  diff_collage/condind_circle.py:41-84   CondIndCircle.get_eps_t_fn
  guided_diffusion/condition_functions.py:30-42  dc_model_fn (H<->W transpose around the call)
"""
import numpy as np

F32 = np.float32
BASE = 128


def split_wimg(wimg, n_img):
    """(B,C,h,W) -> (B*n,C,h,128), window i of sample b at row b*n+i; returns (windows, overlap)."""
    b, c, h, w = wimg.shape
    ov = (n_img * BASE - w) // (n_img - 1) if n_img > 1 else 0
    assert n_img * BASE - ov * (n_img - 1) == w
    st = BASE - ov
    wins = np.stack([wimg[..., i * st:i * st + BASE] for i in range(n_img)], axis=1)
    return wins.reshape(b * n_img, c, h, BASE).astype(F32), ov


def merge_wimg(imgs, overlap, n, is_avg=True):
    """(B*n,C,h,w) -> (B,C,h,n*w-(n-1)*overlap) by summation (and division by coverage)."""
    bn, c, h, w = imgs.shape
    b = bn // n
    st = w - overlap
    out = np.zeros((b, c, h, n * w - (n - 1) * overlap), dtype=F32)
    cnt = np.zeros_like(out)
    iv = imgs.reshape(b, n, c, h, w)
    for i in range(n):
        out[..., i * st:i * st + w] += iv[:, i]
        cnt[..., i * st:i * st + w] += 1
    return (out / cnt).astype(F32) if is_avg else out


def condind_eps(long_x, t, eps_fn, num_img, overlap, y=None, circle=False):
    """eps of a long latent (B,C,h,W) from window evaluations; eps_fn(x, t, y=...) on (.,C,h,w)."""
    x = np.concatenate((long_x, long_x[..., :overlap]), axis=-1) if circle else long_x
    xs, _ = split_wimg(x, num_img)
    tt = np.repeat(np.asarray(t), num_img)
    yy = None if y is None else np.repeat(np.asarray(y), num_img)
    b = long_x.shape[0]
    full = eps_fn(xs, tt, y=yy).astype(F32).reshape((b, num_img) + xs.shape[1:]).copy()
    half = eps_fn(np.ascontiguousarray(xs[..., -overlap:]), tt, y=yy).astype(F32)
    half = half.reshape((b, num_img) + half.shape[1:]).copy()
    half[:, -1] = 0
    full[..., -overlap:] -= half
    long_eps = merge_wimg(full.reshape((b * num_img,) + xs.shape[1:]), overlap, num_img, is_avg=False)
    if not circle:
        return long_eps
    return np.concatenate(((long_eps[..., :overlap] + long_eps[..., -overlap:]) / F32(2.0),
                           long_eps[..., overlap:-overlap]), axis=-1).astype(F32)

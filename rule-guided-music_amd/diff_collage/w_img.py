"""Window split / merge of a long latent along its last axis -- reference API (diff_collage/w_img.py:8-48),
HIP kernels underneath (csrc/collage.hip).  Windows are 128 wide (the eps-network's native length)."""
import torch as th

from rgm import native as _rgm

BASE = 128


def _overlap(w, n_img):
    ov = (n_img * BASE - w) // (n_img - 1) if n_img > 1 else 0
    assert n_img * BASE - ov * (n_img - 1) == w, f"width {w} is not {n_img} windows of {BASE}"
    return ov


def split_windows(wimg, n_img, overlap, want_halves=False):
    """(B,C,h,W) -> (B*n,C,h,128) [and the right `overlap` columns (B*n,C,h,overlap)]; reads wrap around W."""
    _rgm.require_cuda(wimg)
    wimg = wimg.float().contiguous()
    B, Cc, h, W = wimg.shape
    wins = th.empty((B * n_img, Cc, h, BASE), dtype=th.float32, device=wimg.device)
    halves = th.empty((B * n_img, Cc, h, overlap), dtype=th.float32, device=wimg.device) if want_halves else None
    with th.cuda.device(wimg.device):
        _rgm.check(_rgm.lib.rgm_collage_split(_rgm.ptr(wimg), _rgm.ptr(wins), _rgm.ptr(halves), B, Cc, h, W, n_img, overlap,
                                              _rgm.current_stream()))
    return (wins, halves) if want_halves else wins


def split_wimg(wimg, n_img, rtn_overlap=True):
    if wimg.ndim == 3:
        wimg = wimg[None]
    ov = _overlap(wimg.shape[-1], n_img)
    img = split_windows(wimg, n_img, ov)
    return (img, ov) if rtn_overlap else img


def merge_windows(full, half, overlap_size, n, circle=False, is_avg=False):
    _rgm.require_cuda(full, half)
    full = full.float().contiguous()
    half = None if half is None else half.float().contiguous()
    bn, Cc, h, w = full.shape
    assert w == BASE
    B = bn // n
    Wl = n * BASE - (n - 1) * overlap_size
    out = th.empty((B, Cc, h, Wl - overlap_size if circle else Wl), dtype=th.float32, device=full.device)
    with th.cuda.device(full.device):
        _rgm.check(_rgm.lib.rgm_collage_merge(_rgm.ptr(full), _rgm.ptr(half), _rgm.ptr(out), B, Cc, h, n, overlap_size,
                                              int(circle), int(is_avg), _rgm.current_stream()))
    return out


def avg_merge_wimg(imgs, overlap_size, n=None, is_avg=True):
    if n is None:
        n = imgs.shape[0]
    return merge_windows(imgs, None, overlap_size, n, circle=False, is_avg=is_avg)

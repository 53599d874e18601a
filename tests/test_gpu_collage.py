"""-m gpu: DiffCollage split / merge kernels and the CondIndSimple / CondIndCircle eps against the goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from rgm import synth

pytestmark = pytest.mark.gpu
SM = dict(depth=2, hidden=384, heads=6, patch=8, in_ch=4, out_ch=4, num_classes=3)


def test_split_merge_and_condind_eps(precision):
    from gpu_util import dev, rel, load_module
    from guided_diffusion.dit import DiTRotary
    import diff_collage as dc
    from oracle import collage_np as ocl
    g = load_golden("collage")
    w = g["w"]
    xs, ov = dc.split_wimg(dev(w), 7)
    oxs, _ = ocl.split_wimg(w, 7)
    assert ov == 64 and np.array_equal(xs.cpu().numpy(), oxs)                       # pure gather: bit-exact
    assert rel(dc.avg_merge_wimg(xs, 64, n=7, is_avg=True).cpu().numpy(), g["merge_avg"]) < 1e-6
    m = DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=2, num_heads=6, num_classes=3, learn_sigma=False)
    m = load_module(m, synth.dit_state_dict(11, **SM))

    def eps_fn(x, t, y=None):                                                        # scripts/sample_rule.py:120-122
        return m(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
    t, y = dev(g["t"]), dev(g["y"])
    lin = dc.CondIndSimple((4, 16, 128), eps_fn, 7, overlap_size=64)
    assert lin.shape == (4, 16, 512)
    assert rel(lin.eps_scalar_t_fn(dev(w), t, y=y).cpu().numpy(), g["eps_linear"]) < 2e-4
    cir = dc.CondIndCircle((4, 16, 128), eps_fn, 8, overlap_size=64)
    assert cir.shape == (4, 16, 512)
    assert rel(cir.eps_scalar_t_fn(dev(w), t, y=y).cpu().numpy(), g["eps_circle"]) < 2e-4
    # linearity of the composition: eps of windows that are all zero except one is that window's placement
    z = torch.zeros(7, 4, 16, 128, device="cuda")
    z[3] = 1.0
    from diff_collage.w_img import merge_windows
    out = merge_windows(z, None, 64, 7)
    assert out.shape == (1, 4, 16, 512) and float(out[..., 192:320].min()) == 1.0 and float(out.sum()) == 4 * 16 * 128


def test_replicated_collage_forward_shares_its_windows_out_over_the_ranks(monkeypatch):
    """BASELINE config 5 on 8 GPUs is ONE long sample: the x_t forward of a search step has no rows to share out, so the linear collage
    shares out its windows (rgm/batch_shard.py WINDOW_SHARD: window i of [7 full | 6 halves] on rank i % R, one all-reduce of the
    zero-filled window eps).  Replayed as 'rank r of 4' and 'of 8' with a stand-in all-reduce that adds what the other ranks would have
    computed: every rank ends on the unsharded eps (up to the batch-size dependence of the GEMM tiles), each window computed exactly once."""
    import diff_collage as dc
    from gpu_util import dev, rel
    from rgm import batch_shard
    from test_gpu_sampler import SM, _dit
    m = _dit(SM, 11)
    calls = []

    def eps_fn(x, t, y=None):
        calls.append(tuple(x.shape))
        return m(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
    worker = dc.CondIndSimple((4, 16, 128), eps_fn, 7, overlap_size=64)
    rng = np.random.RandomState(3)
    w = dev(rng.randn(1, 4, 16, 512).astype(np.float32))
    t = dev(np.array([300], dtype=np.int64))
    y = dev(np.array([2], dtype=np.int64))
    ref = worker.eps_scalar_t_fn(w, t, y=y)
    assert calls == [(7, 4, 16, 128), (6, 4, 16, 64)]          # the last window's half is not evaluated (the reference zeroes it)
    for world in (4, 8):
        bufs = {}
        for rank in range(world):                              # pass 1: what every rank contributes
            monkeypatch.setattr(batch_shard, "window_world", lambda world=world, rank=rank: (world, rank))
            monkeypatch.setattr(batch_shard, "reduce_windows", lambda tns, rank=rank: bufs.__setitem__(rank, tns.clone()) or tns)
            worker.eps_scalar_t_fn(w, t, y=y)
        stack = torch.stack([bufs[r] for r in range(world)])
        assert int(((stack != 0).sum(0) > 1).sum()) == 0       # every element has one contributor
        total = stack.sum(0)
        for rank in (0, world - 1):                            # pass 2: the completed buffer gives the unsharded eps
            monkeypatch.setattr(batch_shard, "window_world", lambda world=world, rank=rank: (world, rank))
            monkeypatch.setattr(batch_shard, "reduce_windows", lambda tns: tns.copy_(total))
            out = worker.eps_scalar_t_fn(w, t, y=y)
            assert rel(out.cpu().numpy(), ref.cpu().numpy()) < 2e-5

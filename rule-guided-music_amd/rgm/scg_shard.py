"""SCG candidate sharding over torch.distributed ranks (RCCL over xGMI on the GPU box, gloo in CPU tests).

SCG's n candidates of a step are independent between the noise draw and the argmax
(reference guided_diffusion/gaussian_diffusion.py:509-540), so rank r of R owns the contiguous block
k in [r*n/R, (r+1)*n/R) -- contiguous so that "first maximum wins" is preserved by concatenating the
per-rank (n/R, B) log-prob tables in rank order.  The only exchange is ONE all-gather of n*B floats per
guided step (latency-bound; nothing is per-link bandwidth bound on the point-to-point xGMI mesh).  Every
rank then runs the same argmax on the same table and regenerates the winning candidate from the shared
counter-based noise stream, so no latent ever crosses the fabric.

Pure host logic: no HIP calls here, so the N>1 control flow is testable with world_size-2 gloo on CPU.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def partition(n, world_size=None, rank=None):
    """(first candidate, local count, sharded?) for this rank; unsharded when n does not divide evenly."""
    if world_size is None:
        world_size, rank = world()
    if world_size <= 1 or n % world_size != 0:
        return 0, n, False
    per = n // world_size
    return rank * per, per, True


def gather_totals(total_local):
    """(n/R, B) float32 per rank -> (n, B) on every rank, rank-major (== candidate order)."""
    R, _ = world()
    total_local = total_local.contiguous()
    if total_local.is_cuda and dist.get_backend() == "nccl":   # RCCL: one all-gather straight into the (n, B) table
        out = torch.empty((R * total_local.shape[0],) + tuple(total_local.shape[1:]), dtype=total_local.dtype, device=total_local.device)
        dist.all_gather_into_tensor(out, total_local)
        return out
    parts = [torch.empty_like(total_local) for _ in range(R)]
    dist.all_gather(parts, total_local)
    return torch.cat(parts, dim=0)


def first_argmax(total_all):
    """Host-side restatement of the selection rule for tests: first maximum over dim 0, NaN counts as max."""
    return torch.argmax(total_all, dim=0)

"""Batch sharding of the non-SCG work over torch.distributed ranks ("classifier guidance and unguided steps are
batch-parallel: shard B across ranks with an all-gather of x_{t-1}").

A reverse step is independent per sample everywhere outside SCG's candidate search: the eps-network forward, the classifier's
value-and-gradient, DPS, the fused step update and the noise draw (counter-based: row b of a draw is the same numbers on any
rank).  Rank r of R therefore computes rows [r*B/R, (r+1)*B/R) of the step and ONE all-gather of the new latents (and the x0
estimates: 2 x 32 KiB per sample) gives every rank the full batch for the next step.  SCG search steps keep their own split
(candidates, rgm/scg_shard.py) for the search itself; the x_t forward and the classifier gradient that come BEFORE the search are
shared out by rows too (partition_rows: one all-gather of eps (+ gradient), 32 KiB per sample each), instead of every rank
repeating them.  Batches that do not divide by the world size stay replicated.

Pure host logic (no HIP calls): the N > 1 control flow is testable with world_size-2 gloo on CPU.
"""
import threading
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def partition(B, world_size=None, rank=None):
    """(first row, local rows, sharded?) of this rank; replicated when B does not divide evenly."""
    if world_size is None:
        world_size, rank = world()
    if world_size <= 1 or B % world_size != 0:
        return 0, B, False
    per = B // world_size
    return rank * per, per, True


def partition_rows(B, world_size=None, rank=None):
    """Rows of the x_t forward (and classifier gradient) of an SCG search step this rank computes: (first row, rows) or None =
    replicated.  B % R == 0: B / R rows each, in rank order.  R % B == 0 (more ranks than samples, e.g. B = 4 on 8 GPUs): one row
    each, rank r takes row r % B -- the all-gather then holds rows 0..B-1 in its first B entries (ranks 0..B-1) and copies after."""
    if world_size is None:
        world_size, rank = world()
    if world_size <= 1 or B <= 0:
        return None
    if B % world_size == 0:
        per = B // world_size
        return rank * per, per
    if world_size % B == 0:
        return rank % B, 1
    return None


_orig_partition_rows = partition_rows        # (tests replace partition_rows with one rank's view and still need the rule itself)


def partition_roles(B, world_size=None, rank=None):
    """At least twice as many ranks as samples (B = 4 on 8 GPUs -- the north star's C4) AND a classifier gradient to compute: the two
    per-sample forwards of a search step do not depend on each other (the guidance gradient is a function of x_t and t alone,
    condition_functions.py:58-64), so they go to DIFFERENT ranks instead of one after the other on the same one.  Returns
    (row, role) -- role 0: the eps-network forward of that row, role 1: its classifier gradient -- or None when the rule does not apply.
    Ranks [0, B) take role 0, ranks [B, 2B) role 1, further ranks repeat the pattern (their results are not read)."""
    if world_size is None:
        world_size, rank = world()
    if world_size <= 1 or B <= 0 or world_size % B != 0 or world_size < 2 * B:
        return None
    return rank % B, (rank // B) % 2


_orig_partition_roles = partition_roles


# ---- window sharding of a replicated DiffCollage forward (BASELINE config 5: ONE long sample on 8 ranks)
# The x_t forward of a search step is per-sample work; with B = 1 every rank would repeat all 7 + 6 windows of the collage.  While
# the window-shard flag is set (set_window_shard, by gaussian_diffusion._search_step_inputs around that forward only) diff_collage's CondIndSimple evaluates the
# windows whose index is congruent to this rank and completes the rest with ONE all-reduce of the (zero-filled) window eps: every element
# has exactly one non-zero contributor, so the sum is exact and identical on every rank.
# Per THREAD (advisor, round 4): a collage model evaluated on another host thread while this one is inside its sharded forward must not enter
# the all-reduce -- if only some ranks took that path the collective would deadlock.
_window_tls = threading.local()


def window_shard_on():
    """True while THIS thread is inside the sharded x_t forward of a search step"""
    return getattr(_window_tls, "on", False)


def set_window_shard(on):
    _window_tls.on = bool(on)


def window_ranks():
    """ranks a replicated collage forward can be shared out over (bench.py --simulate-ranks replaces this and window_world)"""
    return world()[0]


def window_world():
    """(world size, rank) the collage worker shards its windows over -- (1, 0) outside a sharded x_t forward"""
    return world() if window_shard_on() else (1, 0)


def window_share(n_full, n_half, world_size, rank):
    """This rank's share of the window forwards of one collage evaluation: item i of the list [n_full full windows | n_half half windows]
    goes to rank i % world_size.  Returns (indices into the full windows, indices into the half windows) as range objects."""
    first_h = min((rank - n_full) % world_size, n_half)
    return range(min(rank, n_full), n_full, world_size), range(first_h, n_half, world_size)


def reduce_windows(t):
    """sum over the ranks, in place: ONE collective per collage forward"""
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


PER_SAMPLE_VECTORS = ("y", "t", "timesteps")      # the only 1-D tensors that carry one entry per sample


def slice_rows(obj, B, b0, nb, key=None):
    """Rows [b0, b0+nb) of every per-sample tensor inside a (nested) dict / list / tuple; other values pass through (per-call scalars,
    namespaces, tensors broadcast over the batch such as a (1,C,H,W) edit mask).  Per-sample means: at least 2-D with leading
    dimension B (latents, (B,K) rule targets, masks), or a 1-D tensor of length B stored under one of PER_SAMPLE_VECTORS (class labels,
    timesteps) -- an unbatched 1-D tensor that merely happens to have B entries (a (16,) target with B = 16) is NOT cut."""
    if torch.is_tensor(obj):
        if obj.dim() >= 2 and obj.shape[0] == B:
            return obj[b0:b0 + nb].contiguous()
        if obj.dim() == 1 and obj.shape[0] == B and key in PER_SAMPLE_VECTORS:
            return obj[b0:b0 + nb].contiguous()
        return obj
    if isinstance(obj, dict):
        return {k: slice_rows(v, B, b0, nb, k) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(slice_rows(v, B, b0, nb, key) for v in obj)
    return obj


def gather_rows(tensors):
    """[(nb, ...)] per rank, all of one shape / dtype -> [(R*nb, ...)] on every rank, rank-major (= row order): ONE collective."""
    R, _ = world()
    local = torch.stack([t.contiguous() for t in tensors], dim=1).contiguous()        # (nb, k, ...)
    if local.is_cuda and dist.get_backend() == "nccl":
        out = torch.empty((R * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
    else:
        parts = [torch.empty_like(local) for _ in range(R)]
        dist.all_gather(parts, local)
        out = torch.cat(parts, dim=0)
    return [out[:, i].contiguous() for i in range(len(tensors))]

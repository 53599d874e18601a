"""Process-group bootstrap -- replaces the reference's MPI rendezvous (guided_diffusion/dist_util.py:21-104).

One process per GPU, launched by torchrun (or plain python for a single GPU): RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT come from the environment, the backend is "nccl" (= RCCL over xGMI on ROCm) when a
GPU is present and "gloo" otherwise.  There is no MPI in this stack; checkpoints are read by every rank from
its local filesystem (the reference broadcasts them over MPI in 1 GiB chunks)."""
import os

import torch as th
import torch.distributed as dist


def setup_dist(port=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns None (there is no MPI comm)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if th.cuda.is_available():
        th.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % th.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if port is not None:
            os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl" if th.cuda.is_available() else "gloo", init_method="env://")
    return None


def dev():
    if th.cuda.is_available():
        return th.device("cuda", th.cuda.current_device())
    return th.device("cpu")


def load_state_dict(path, **kwargs):
    return th.load(path, **kwargs)


def sync_params(params):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for p in params:
            with th.no_grad():
                dist.broadcast(p, 0)

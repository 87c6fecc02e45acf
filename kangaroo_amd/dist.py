"""Multi-GPU plumbing: one process per GPU, independent herds, no collective in the jump path.

The reference runs one host thread + one GPUEngine per GPU with fully independent herds that only
share the host DP table (core/kangaroo.cpp:780-786, :475).  Here the unit of sharding is the same —
a disjoint herd per rank — and torch.distributed (RCCL on GPUs, gloo on CPU) is used only for the
start-up broadcast of the run configuration and for barriers / the max-over-ranks of timings.
"""
import os

import numpy as np

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _device():
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def barrier():
    _, _, world = env_world()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()


def max_over_ranks(value):
    _, _, world = env_world()
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    _, _, world = env_world()
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_config(arr):
    """Rank 0's uint64 array (jump table + dpMask + wild offset, a few KiB) to every rank — the one-time
    broadcast SURVEY §8e allows; nothing moves between GPUs per jump."""
    _, _, world = env_world()
    if world == 1:
        return arr
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy()).to(_device())
    dist.broadcast(t, src=0)
    return t.cpu().numpy().view(np.uint64).reshape(arr.shape)


def herd_shard(rank, world, n_per_rank, base=0x1234567, stride=0x9E3779B97F4A7C15F39CC0605CEDC835):
    """Start scalar of rank's synthetic herd: ranks own disjoint arithmetic progressions of distances."""
    assert 0 <= rank < world
    return (base + rank * n_per_rank * stride) % N_ORDER, stride

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


def init(backend=None, timeout_s=600.0):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world).
    timeout_s bounds the rendezvous and every collective: a rank that died must not leave the others waiting for
    the default half hour (the only collectives here are barriers and reductions of a few bytes)."""
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
            import datetime

            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=timeout_s))
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


def all_gather_array(arr):
    """A fixed-size uint64 array from every rank -> [world, len] on every rank (control words of the solve loop)."""
    rank, _, world = env_world()
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    if world == 1:
        return arr.reshape(1, -1).copy()
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(arr.view(np.int64).copy()).to(_device())
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in outs])


def all_gather_varlen(arr):
    """A uint64 array of ANY length from every rank -> the concatenation in rank order on every rank (the
    same-herd collisions of one launch: usually none, no fixed bound)."""
    rank, _, world = env_world()
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    if world == 1:
        return arr.copy()
    lens = all_gather_array(np.array([len(arr)], dtype=np.uint64))[:, 0].astype(np.int64)
    longest = int(lens.max())
    if longest == 0:
        return arr[:0].copy()
    buf = np.zeros(longest, dtype=np.uint64)
    buf[: len(arr)] = arr
    rows = all_gather_array(buf)
    return np.concatenate([rows[r][: int(lens[r])] for r in range(world)])


def all_gather_items(items):
    """The DP exchange: every rank's distinguished points of one launch (structured array, any length) ->
    the concatenation over ranks, in rank order, on every rank.

    This is the one exchange step the reference's multi-GPU solve has — all GPU threads feed one DP table
    (core/kangaroo.cpp:473-490).  With one process per GPU the table is sharded by x across the ranks
    (solve.solve_key_multi) and the DP stream is all-gathered: 72 B per DP, a few hundred thousand DPs per
    second and GPU at the reference's DP sizes, i.e. tens of MB/s against xGMI links of 150 GB/s.  Lengths
    differ per rank, so the counts go first and the payload is padded to the longest."""
    rank, _, world = env_world()
    items = np.ascontiguousarray(items)
    if world == 1:
        return items.copy()
    import torch
    import torch.distributed as dist

    dev = _device()
    cnt = torch.tensor([len(items)], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [int(c.item()) for c in cnts]
    longest = max(cnts)
    if longest == 0:
        return items[:0].copy()
    size = items.dtype.itemsize
    buf = torch.zeros(longest * size, dtype=torch.uint8, device=dev)
    if len(items):
        buf[: len(items) * size] = torch.from_numpy(items.view(np.uint8).reshape(-1).copy()).to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = [outs[r][: cnts[r] * size].cpu().numpy().view(items.dtype) for r in range(world)]
    return np.concatenate(parts)

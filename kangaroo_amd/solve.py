"""Host loop around the jump engine: the per-GPU thread of the reference, `Kangaroo::SolveKeyGPU`
(core/kangaroo.cpp:397-508), restated on the product's own host helpers (kangaroo_amd/host.py) so that a
key can be solved on a machine that holds nothing but this repository.

Sequence kept from the reference: create nbThread herds of 128 alternating tame/wild kangaroos
(:423-433), SetWildOffset(rangeWidth/2) (:436), SetParams(dpMask, jump table) (:437), SetKangaroos (:438),
callKernel (:447); then per iteration Launch (:458), feed every DP to the table (:473-490) and re-seed the
kangaroos whose DP collided inside their own herd with SetKangaroo (:478-486).
"""
import time

from . import host
from .engine import GPUEngine


def dp_mask_of(dp_bits):
    """Kangaroo::SetDP (core/kangaroo.cpp:110-120): the top dp_bits of x.bits64[3]."""
    if dp_bits <= 0:
        return 0
    dp_bits = min(dp_bits, 64)
    return ((1 << dp_bits) - 1) << (64 - dp_bits)


def suggested_dp(range_power, n_kangaroo):
    """Kangaroo::Run (core/kangaroo.cpp:742-744): rangePower/2 - log2(kangaroos) - 1, floored at 0."""
    import math

    return max(0, int(range_power / 2.0 - math.log2(n_kangaroo) - 1))


def solve_key(range_start, range_end, pub, grid=(0, 0), gpu_id=0, dp_bits=None, seed=1, max_launches=None,
              nb_run=None, progress=None):
    """Returns (private_key or None, stats).  pub = (x, y) affine point; ranges are Python ints."""
    sv = host.Solver(range_start, range_end, pub)
    gx, gy = GPUEngine.GetGridSize(gpu_id, grid[0], grid[1])
    eng = GPUEngine(gx, gy, gpu_id, 65536 * 2)
    try:
        n = eng.nbKangaroo
        if dp_bits is None:
            # the reference's suggestion, raised if needed so that a launch cannot overflow the DP ring
            # (the reference would exit with "items lost", engine.cu:506-510)
            import math

            ring = max(65536 * 2, min(2 * n, 1 << 22))
            per_launch = n * (nb_run or 256)
            dp_bits = max(suggested_dp(sv.range_power, n), math.ceil(math.log2(per_launch / (ring / 2))))
        rng = host.Rng(seed)
        # CreateHerd (core/kangaroo.cpp:521-573): the distances with the reference's RNG draws on the host,
        # the points d*G (+ key for wild) on the GPU (kg_herd_from_distances)
        d = host.herd_distances(rng, sv.range_power, sv.range_width, n, 0)
        jd, jx, jy, _ = host.create_jump_table(sv.range_power)
        eng.SetWildOffset(sv.wild_offset)
        eng.SetParams(dp_mask_of(dp_bits), jd, jx, jy)
        if nb_run:
            eng.SetNbRun(nb_run)
        eng.HerdFromDistances(d, sv.key)
        del d
        steps = eng.nb_run
        t0 = time.perf_counter()
        eng.callKernel()
        launches = 0
        n_dp = 0
        while sv.found is None and (max_launches is None or launches < max_launches):
            items = eng.Launch()
            launches += 1
            n_dp += len(items)
            if len(items):
                _, dead = sv.add_items(items)
                for di in dead:
                    k_idx = int(items[di]["kIdx"])
                    nx, ny, nd = host.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, k_idx % 2)
                    eng.SetKangaroo(k_idx, nx[0], ny[0], nd[0])
            if progress:
                progress(launches, n_dp, sv.table_size)
        eng.Collect()  # drain the launch still in flight
        dt = time.perf_counter() - t0
        stats = {"kangaroos": n, "launches": launches, "jumps": n * steps * launches, "seconds": dt, "dps": n_dp,
                 "table": sv.table_size, "dead": sv.dead, "dp_bits": dp_bits, "range_power": sv.range_power,
                 "device": eng.deviceName}
        return sv.found, stats
    finally:
        eng.close()


MAX_DEAD_PER_LAUNCH = 250  # same-herd collisions reported per rank and launch (the rest stay as they are)


def solve_key_multi(range_start, range_end, pub, grid=(0, 0), dp_bits=None, seed=1, max_launches=None, nb_run=None,
                    progress=None, engine_factory=None):
    """The reference's multi-GPU solve (one GPUEngine per device feeding one DP table,
    core/kangaroo.cpp:780-786 and :473-490) with one PROCESS per GPU, started by torchrun.

    Every rank walks its own herd (own RNG stream).  After each launch the DP items of all ranks are
    all-gathered (dist.all_gather_items: the path's one exchange step); the DP table is sharded over the
    ranks by the low bits of x, so every rank inserts only its share, and a collision is found by the rank
    that owns the x.  A fixed-size control record per rank (found flag, key, same-herd collisions to re-seed)
    is all-gathered after the inserts; every rank returns the key.  world_size 1 runs the same code without
    torch.distributed.  engine_factory(local_rank) -> engine lets tests drive the loop without a GPU."""
    import numpy as np

    from . import dist

    rank, local_rank, world = dist.init()
    sv = host.Solver(range_start, range_end, pub)
    if engine_factory is not None:
        eng = engine_factory(local_rank)
    else:
        gx, gy = GPUEngine.GetGridSize(local_rank, grid[0], grid[1])
        eng = GPUEngine(gx, gy, local_rank, 65536 * 2)
    try:
        n = eng.nbKangaroo
        if dp_bits is None:
            import math

            ring = max(65536 * 2, min(2 * n, 1 << 22))
            dp_bits = max(suggested_dp(sv.range_power, n * world), math.ceil(math.log2(n * (nb_run or 256) / (ring / 2))))
        rng = host.Rng(seed + 0x9E3779B1 * rank)
        d = host.herd_distances(rng, sv.range_power, sv.range_width, n, 0)
        jd, jx, jy, _ = host.create_jump_table(sv.range_power)
        eng.SetWildOffset(sv.wild_offset)
        eng.SetParams(dp_mask_of(dp_bits), jd, jx, jy)
        if nb_run:
            eng.SetNbRun(nb_run)
        eng.HerdFromDistances(d, sv.key)
        del d
        steps = eng.nb_run
        t0 = time.perf_counter()
        eng.callKernel()
        launches = 0
        n_dp = 0
        key = None
        while key is None and (max_launches is None or launches < max_launches):
            items = eng.Launch()
            launches += 1
            items["kIdx"] += np.uint64(rank * n)  # n is even: tame/wild parity survives
            everyone = dist.all_gather_items(items)
            n_dp += len(everyone)
            mine = everyone[(everyone["x"][:, 0] % np.uint64(world)) == rank] if world > 1 else everyone
            ctrl = np.zeros(6 + MAX_DEAD_PER_LAUNCH, dtype=np.uint64)
            if len(mine):
                _, dead = sv.add_items(mine)
                dead = dead[:MAX_DEAD_PER_LAUNCH]
                ctrl[5] = len(dead)
                ctrl[6:6 + len(dead)] = mine["kIdx"][dead]
            if sv.found is not None:
                ctrl[0] = 1
                ctrl[1:5] = host.limbs(sv.found)
            for c in dist.all_gather_array(ctrl):
                if c[0] and key is None:
                    key = host.to_int(c[1:5])
                for g in c[6:6 + int(c[5])]:
                    g = int(g)
                    if g // n == rank:
                        nx, ny, nd = host.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, g % 2)
                        eng.SetKangaroo(g % n, nx[0], ny[0], nd[0])
            if progress and rank == 0:
                progress(launches, n_dp, sv.table_size)
        eng.Collect()
        dt = time.perf_counter() - t0
        stats = {"kangaroos": n * world, "launches": launches, "jumps": n * world * steps * launches, "seconds": dt,
                 "dps": n_dp, "table_this_rank": sv.table_size, "dead_this_rank": sv.dead, "dp_bits": dp_bits,
                 "range_power": sv.range_power, "world": world, "device": eng.deviceName}
        return key, stats
    finally:
        eng.close()

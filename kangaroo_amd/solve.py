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

"""Host loop around the jump engine: the per-GPU thread of the reference, `Kangaroo::SolveKeyGPU`
(core/kangaroo.cpp:397-508), and the part of `Kangaroo::Run` that starts one such thread per GPU and loops over
the keys of the input file (core/kangaroo.cpp:767-791), restated on the product's own host helpers
(kangaroo_amd/host.py) so that keys can be solved on a machine that holds nothing but this repository.

Sequence kept from the reference, per GPU thread: a fresh GPUEngine per key (:408), nbThread herds of 128
alternating tame/wild kangaroos or the kangaroos of a work file (:416-433, storage/backup.cpp:135-150),
SetWildOffset(rangeWidth/2) (:436), SetParams(dpMask, jump table) (:437), SetKangaroos (:438), callKernel (:447);
then per iteration Launch (:458), feed every DP to the ONE table all GPU threads share (:473-490), re-seed the
kangaroos whose DP collided inside their own herd with SetKangaroo (:478-486), and on a save request hand the
herd over with GetKangaroos and park until the file is written (:493-500, storage/backup.cpp:327-428).
"""
import math
import threading
import time

import numpy as np

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
    return max(0, int(range_power / 2.0 - math.log2(n_kangaroo) - 1))


RING_MAX = 1 << 25  # KG_RING_MAX (engine_capi.hip): the most items one launch can return


def _ring_safe_dp(n, nb_run):
    """Smallest dp_bits at which one launch of n kangaroos fills at most half of the largest DP ring the engine
    allocates (the ring follows the DP size up to RING_MAX items; beyond it the reference's fate applies: "items
    lost", engine.cu:506-510)."""
    return max(0, math.ceil(math.log2(n * (nb_run or 256) / (RING_MAX / 2))))


class _Search:
    """What the GPU threads of one key share: the fields of the reference's Kangaroo object they touch
    (hashTable + ghMutex -> host.Solver, endOfSearch, saveRequest / isWaiting / saveMutex, counters)."""

    def __init__(self, sv, n_threads):
        self.sv = sv
        self.end = threading.Event()            # endOfSearch
        self.save_request = threading.Event()   # saveRequest
        self.save_done = threading.Event()      # release of saveMutex
        self.lock = threading.Lock()
        self.waiting = 0                        # threads with isWaiting set
        self.herds = [None] * n_threads         # what GetKangaroos returned at the last save request
        self.launches = [0] * n_threads
        self.dps = [0] * n_threads
        self.started = [False] * n_threads      # hasStarted
        self.errors = []
        self.stats = [None] * n_threads
        self.n_threads = n_threads


def _gpu_thread(st, th_id, gpu_id, grid, dp_bits, seed, nb_run, herd, keep_herd, max_launches, engine_factory):
    """Kangaroo::SolveKeyGPU (core/kangaroo.cpp:397-508) for thread th_id on device gpu_id.
    herd: (px, py, d) loaded from a work file (already n entries) or None -> CreateHerd."""
    sv = st.sv
    eng = None
    try:
        if engine_factory is not None:
            eng = engine_factory(gpu_id)
        else:
            gx, gy = GPUEngine.GetGridSize(gpu_id, grid[0], grid[1])
            eng = GPUEngine(gx, gy, gpu_id, 65536 * 2)
        n = eng.nbKangaroo
        rng = host.Rng(seed + 0x9E3779B1 * th_id)
        jd, jx, jy, _ = host.create_jump_table(sv.range_power)
        eng.SetWildOffset(sv.wild_offset)
        eng.SetParams(dp_mask_of(dp_bits), jd, jx, jy)
        if nb_run:
            eng.SetNbRun(nb_run)
        if herd is not None:
            eng.SetKangaroos(*herd)  # kangaroos of the work file (FetchWalks)
        elif hasattr(eng, "HerdFromDistances"):
            # CreateHerd (core/kangaroo.cpp:521-573): the distances with the reference's RNG draws on the host,
            # the points d*G (+ key for wild) on the GPU (kg_herd_from_distances)
            eng.HerdFromDistances(host.herd_distances(rng, sv.range_power, sv.range_width, n, 0), sv.key)
        else:
            eng.SetKangaroos(*host.create_herd(rng, sv.range_power, sv.range_width, sv.key, n, 0))
        herd = None
        t0 = time.perf_counter()
        eng.callKernel()
        st.started[th_id] = True
        while not st.end.is_set():
            items = eng.Launch(copy=False)  # consumed below, before the next Launch
            st.launches[th_id] += 1
            st.dps[th_id] += len(items)
            if len(items):
                rc, dead = sv.add_items(items)  # thread-safe, sharded (the reference: one ghMutex for the batch)
                for di in dead:
                    k_idx = int(items[di]["kIdx"])
                    nx, ny, nd = host.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, k_idx % 2)
                    eng.SetKangaroo(k_idx, nx[0], ny[0], nd[0])
                if sv.found is not None:
                    st.end.set()
            if max_launches is not None and st.launches[th_id] >= max_launches:
                st.end.set()
            if st.save_request.is_set() and not st.end.is_set():
                # core/kangaroo.cpp:493-500: hand the herd over, then park until the file is written
                if keep_herd:
                    st.herds[th_id] = eng.GetKangaroos()  # waits for the launch in flight
                with st.lock:
                    st.waiting += 1
                st.save_done.wait()
                with st.lock:
                    st.waiting -= 1
        eng.Collect()  # drain the launch still in flight
        st.stats[th_id] = {"kangaroos": n, "launches": st.launches[th_id], "dps": st.dps[th_id],
                           "jumps": n * eng.nb_run * st.launches[th_id], "seconds": time.perf_counter() - t0,
                           "device": eng.deviceName, "gpu_id": gpu_id}
    except BaseException as ex:  # noqa: BLE001 — reported by the caller's thread
        st.errors.append(ex)
        st.end.set()
        st.started[th_id] = True
    finally:
        if eng is not None:
            eng.close()


def solve_key_threads(range_start, range_end, pub, gpu_ids=(0,), grid=(0, 0), dp_bits=None, seed=1, max_launches=None,
                      nb_run=None, progress=None, work_file=None, save_period=60.0, save_kangaroos=True,
                      stop_after_saves=None, resume_from=None, engine_factory=None, poll=0.02):
    """The reference's multi-GPU solve of ONE key (Kangaroo::Run core/kangaroo.cpp:767-791 for a single key): one
    host thread + one GPUEngine per entry of gpu_ids (an id may repeat, as `-gpuId 0,0` does in the reference),
    fully independent herds, ONE shared DP table (host.Solver, thread-safe) — no collective anywhere.

    work_file / save_period / save_kangaroos: Kangaroo::SaveWork every save_period seconds (the reference's -w /
    -wi / -ws; storage/backup.cpp:327-428).  stop_after_saves: end the search after that many saves (an interrupted
    run).  resume_from: a work file written earlier; its table and kangaroos are loaded (LoadWork + FetchWalks,
    storage/backup.cpp:101-150), kangaroos missing from the file are created.
    Returns (private_key or None, stats)."""
    gpu_ids = list(gpu_ids)
    sv = host.Solver(range_start, range_end, pub)
    st = _Search(sv, len(gpu_ids))

    # engine geometry is needed up front (nbKangaroo per thread, Kangaroo::Run core/kangaroo.cpp:712-719)
    def kangaroos_of(g):
        if engine_factory is not None:
            e = engine_factory(g)
            try:
                return e.nbKangaroo
            finally:
                e.close()
        gx, gy = GPUEngine.GetGridSize(g, grid[0], grid[1])
        return 128 * gx * gy

    n_per = [kangaroos_of(g) for g in gpu_ids]
    total = sum(n_per)
    offset_count, offset_time, herds = 0, 0.0, [None] * len(gpu_ids)
    info = host.work_info(resume_from) if resume_from else None
    if dp_bits is None and info is not None:
        dp_bits = info["dp_bits"]  # LoadWork: "if(initDPSize < 0) initDPSize = payload.dpSize" (storage/backup.cpp:110)
    if dp_bits is None:
        dp_bits = max(suggested_dp(sv.range_power, total), max(_ring_safe_dp(n, nb_run) for n in n_per))
    if resume_from:
        offset_count, offset_time = info["total_count"], info["total_time"]
        px, py, d = sv.load_work(resume_from, total)
        got, pos = len(px), 0
        fill = host.Rng(seed ^ 0x5A5A5A5A)
        for i, n in enumerate(n_per):  # FectchKangaroos: thread after thread takes its share of the file
            take = max(0, min(n, got - pos))
            hx, hy, hd = px[pos:pos + take], py[pos:pos + take], d[pos:pos + take]
            pos += take
            if take < n:  # FetchWalks :145-149: fill with new kangaroos (TAME first)
                cx, cy, cd = host.create_herd(fill, sv.range_power, sv.range_width, sv.key, n - take, 0)
                hx, hy, hd = np.concatenate([hx, cx]), np.concatenate([hy, cy]), np.concatenate([hd, cd])
            herds[i] = (np.ascontiguousarray(hx), np.ascontiguousarray(hy), np.ascontiguousarray(hd))
    keep_herd = bool(work_file) and save_kangaroos
    threads = [threading.Thread(target=_gpu_thread,
                                args=(st, i, g, grid, dp_bits, seed, nb_run, herds[i], keep_herd, max_launches,
                                      engine_factory), daemon=True)
               for i, g in enumerate(gpu_ids)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    herds = None
    saves = 0
    last_save = time.perf_counter()
    # Kangaroo::Process (core/thread.cpp:150-271): the stats / save loop of the main thread
    while not st.end.is_set():
        time.sleep(poll)
        if progress:
            progress(sum(st.launches), sum(st.dps), sv.table_size)
        if work_file and all(st.started) and time.perf_counter() - last_save >= save_period and not st.end.is_set():
            st.save_done.clear()
            st.save_request.set()
            while st.waiting < len(threads) and not st.end.is_set():  # wait until every worker has parked
                time.sleep(poll / 4)
            if not st.end.is_set():
                count = offset_count + sum(n * l for n, l in zip(n_per, st.launches)) * (nb_run or 256)
                herd = None
                if keep_herd:
                    herd = tuple(np.concatenate([h[c] for h in st.herds]) for c in range(3))
                sv.save_work(work_file, dp_bits, count, offset_time + time.perf_counter() - t0, herd)
                saves += 1
            st.save_request.clear()
            if stop_after_saves is not None and saves >= stop_after_saves:
                st.end.set()
            st.save_done.set()
            # every parked worker must have left its wait before the next request may go out: a worker that has not
            # yet woken would otherwise be counted as parked for the NEXT save, which would then run while that
            # worker is already launching and inserting again
            while st.waiting > 0 and any(t.is_alive() for t in threads):
                time.sleep(poll / 4)
            last_save = time.perf_counter()
    st.save_done.set()
    for t in threads:  # the reference forgets to join its GPU threads; this loop does not
        t.join()
    if st.errors:
        raise st.errors[0]
    dt = time.perf_counter() - t0
    per = [s for s in st.stats if s]
    stats = {"kangaroos": total, "launches": sum(st.launches), "jumps": sum(s["jumps"] for s in per), "seconds": dt,
             "dps": sum(st.dps), "table": sv.table_size, "dead": sv.dead, "unstored": sv.unstored, "dp_bits": dp_bits,
             "range_power": sv.range_power, "gpus": gpu_ids, "saves": saves, "per_thread": per,
             "device": per[0]["device"] if per else ""}
    return sv.found, stats


def solve_key(range_start, range_end, pub, grid=(0, 0), gpu_id=0, **kw):
    """One key on one GPU: solve_key_threads with a single GPU thread.  Returns (private_key or None, stats)."""
    return solve_key_threads(range_start, range_end, pub, gpu_ids=(gpu_id,), grid=grid, **kw)


def solve_keys(range_start, range_end, pubs, gpu_ids=(0,), **kw):
    """The key loop of Kangaroo::Run (core/kangaroo.cpp:767-791): the keys of the input file one after the other,
    each with InitSearchKey (a new table, the key shifted by -rangeStart*G) and fresh engines — SolveKeyGPU builds
    its GPUEngine per key (:408) and the unique_ptr destroys it when the key ends.
    Returns [(private_key or None, stats), ...] in input order."""
    out = []
    for i, pub in enumerate(pubs):
        k = dict(kw)
        k["seed"] = kw.get("seed", 1) + 0x01000193 * i  # the reference keeps drawing from one RNG stream across keys
        out.append(solve_key_threads(range_start, range_end, pub, gpu_ids=gpu_ids, **k))
    return out


def solve_key_multi(range_start, range_end, pub, grid=(0, 0), dp_bits=None, seed=1, max_launches=None, nb_run=None,
                    progress=None, engine_factory=None):
    """OPTIONAL variant with one PROCESS per GPU (started by torchrun) for set-ups that want process isolation; the
    north-star shape is solve_key_threads (threads, one shared table, no collective).

    Every rank walks its own herd (own RNG stream).  After each launch the DP items of all ranks are
    all-gathered (dist.all_gather_items); the DP table is sharded over the ranks by the low bits of x, so every
    rank inserts only its share, and a collision is found by the rank that owns the x.  A control record per rank
    (status, key, same-herd collisions to re-seed — any number of them) is all-gathered after the inserts; every
    rank returns the key.  A rank that fails publishes that in its control record before raising, so the others
    stop with it instead of waiting in the next all-gather.  world_size 1 runs the same code without
    torch.distributed.  engine_factory(local_rank) -> engine lets tests drive the loop without a GPU."""
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
            dp_bits = max(suggested_dp(sv.range_power, n * world), _ring_safe_dp(n, nb_run))
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
            error = None
            try:
                items = eng.Launch()
                items["kIdx"] += np.uint64(rank * n)  # n is even: tame/wild parity survives
            except Exception as ex:  # noqa: BLE001 — published below, raised after the exchange
                error = ex
                items = np.zeros(0, dtype=host.ITEM_DTYPE)
            launches += 1
            everyone = dist.all_gather_items(items)
            n_dp += len(everyone)
            mine = everyone[(everyone["x"][:, 0] % np.uint64(world)) == rank] if world > 1 else everyone
            dead_idx = np.zeros(0, dtype=np.uint64)
            if len(mine) and error is None:
                try:
                    _, dead = sv.add_items(mine)
                    dead_idx = np.ascontiguousarray(mine["kIdx"][dead], dtype=np.uint64)
                except Exception as ex:  # noqa: BLE001
                    error = ex
            ctrl = np.zeros(5, dtype=np.uint64)  # [status: 0 running / 1 found / 2 failed, key limbs]
            if error is not None:
                ctrl[0] = 2
            elif sv.found is not None:
                ctrl[0] = 1
                ctrl[1:5] = host.limbs(sv.found)
            ctrls = dist.all_gather_array(ctrl)
            all_dead = dist.all_gather_varlen(dead_idx)
            if error is not None:
                raise error
            failed = [r for r, c in enumerate(ctrls) if int(c[0]) == 2]
            if failed:
                raise RuntimeError(f"solve_key_multi: rank(s) {failed} failed, stopping rank {rank}")
            for c in ctrls:
                if int(c[0]) == 1 and key is None:
                    key = host.to_int(c[1:5])
            for g in all_dead:
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

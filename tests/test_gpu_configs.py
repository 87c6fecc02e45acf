"""BASELINE configs 3 and 5 in miniature on one MI355X, through the C ABI and the product's host loop
(kangaroo_amd/solve.py): the key loop with a fresh engine per key (core/kangaroo.cpp:767-791, :408), the DP = 8
drain with zero loss (engine.cu:506-510, core/kangaroo.cpp:473-490), -w save / -i resume through
GetKangaroos / SetKangaroos (core/kangaroo.cpp:416,438,493-500; storage/backup.cpp:135-150,327-428) checked
against an uninterrupted oracle walk, and a 72-bit solve by two GPU threads sharing one table at the DP size the
reference would pick (core/kangaroo.cpp:742-744, :780-786)."""
import os

import numpy as np
import pytest

import kgoracle as K
from gpu_common import dp_multiset, make_case, oracle_dps
from kangaroo_amd import GPUEngine, host, solve

pytestmark = pytest.mark.gpu


def test_multi_key_loop_fresh_engine_per_key():
    """Config 5's multi-key input: three keys of one 40-bit range, one after the other, each with its own engine."""
    privs = [0xA5B3C1D2E7, 0x1F2E3D4C5B, 0xC0DE12345F]
    pubs = [K.ec_mul_g(p) for p in privs]
    res = solve.solve_keys(0, (1 << 40) - 1, pubs, gpu_ids=(0,), grid=(8, 32), dp_bits=8, seed=3, max_launches=2000)
    assert [r[0] for r in res] == privs
    for _, st in res:
        assert st["jumps"] == st["kangaroos"] * 256 * st["launches"] and st["table"] > 0
    # an engine created after the previous one was destroyed works on the same device, and a search with an
    # offset range start in the same process too
    lo = 1 << 70
    priv = lo + 0xB3C1D2E7F0A1
    (found, st), = solve.solve_keys(lo, lo + (1 << 48) - 1, [K.ec_mul_g(priv)], gpu_ids=(0,), grid=(16, 32), dp_bits=8,
                                    seed=5, max_launches=2000)
    assert found == priv


def test_dp8_drain_without_loss():
    """Config 5's low DP size: at DP = 8 every kangaroo reports about one DP per launch of 256 steps.  Nothing may
    be lost between the kernel's append, the pinned ring, kg_launch and the table: table size + dead == DPs drained."""
    lo, hi = 0, (1 << 110) - 1  # far too wide to be solved: the table only grows
    pub = host.ec_mul_g(0x2545F4914F6CDD1D0123456789ABCDEF % hi)
    sv = host.Solver(lo, hi, pub)
    eng = GPUEngine(128, 128, 0, 65536 * 2)
    n = eng.nbKangaroo
    assert n == 2097152
    jd, jx, jy, _ = host.create_jump_table(sv.range_power)
    eng.SetWildOffset(sv.wild_offset)
    eng.SetParams(solve.dp_mask_of(8), jd, jx, jy)
    eng.HerdFromDistances(host.herd_distances(host.Rng(7), sv.range_power, sv.range_width, n, 0), sv.key)
    rng = host.Rng(8)
    launches, n_dp, reseeded = 12, 0, 0
    eng.callKernel()
    for i in range(launches):
        items = eng.Launch() if i < launches - 1 else eng.Collect()
        assert eng.lost == 0
        assert not np.any(items["x"][:, 3] >> np.uint64(56))  # every item is a DP: top 8 bits of x.bits64[3] clear
        n_dp += len(items)
        _, dead = sv.add_items(items)
        for di in dead:
            k_idx = int(items[di]["kIdx"])
            nx, ny, nd = host.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, k_idx % 2)
            eng.SetKangaroo(k_idx, nx[0], ny[0], nd[0])
            reseeded += 1
    eng.close()
    expect = launches * n * 256 / 256
    assert abs(n_dp - expect) < 6 * expect ** 0.5
    assert sv.found is None
    assert sv.table_size + sv.dead == n_dp and sv.dead == reseeded
    # the items are on their tracks: x == (d*G (+ key for wild)).x for a sample of the last launch
    some = items[np.random.default_rng(5).integers(0, len(items), 256)]
    hx, hy = host.ec_mul_g_batch(np.ascontiguousarray(some["d"]))
    for j, it in enumerate(some):
        pt = (host.to_int(hx[j]), host.to_int(hy[j]))
        if int(it["kIdx"]) & 1:
            pt = host.ec_add(pt, sv.key)
        assert pt[0] == host.to_int(it["x"])


def test_save_resume_equals_uninterrupted_oracle_walk(tmp_path):
    """Walk, GetKangaroos, SaveWork with the herd; then a NEW engine and a NEW table load the file, SetKangaroos,
    and walk on.  The DPs of both halves together, the final herd and the final table must equal those of ONE
    uninterrupted oracle walk of the same herd."""
    case = make_case(64, 128 * 2 * 32, seed=321, dp_bits=6)
    n, steps = 128 * 2 * 32, 32
    pub = K.ec_mul_g(case["priv"])
    lo, hi = 0, case["width"]

    def engine():
        e = GPUEngine(2, 32, 0, 65536 * 2)
        e.SetWildOffset(case["width"] >> 1)
        e.SetParams(case["dp_mask"], case["jd"], case["jx"], case["jy"])
        e.SetNbRun(steps)
        return e

    path = str(tmp_path / "save.kg")
    sv1 = host.Solver(lo, hi, pub)
    e1 = engine()
    e1.SetKangaroos(case["px"], case["py"], case["d"])
    e1.callKernel()
    first = [e1.Launch(), e1.Launch(), e1.Collect()]
    for it in first:
        sv1.add_items(it)
    herd = e1.GetKangaroos()  # core/kangaroo.cpp:495-496
    sv1.save_work(path, 6, total_count=3 * n * steps, total_time=1.5, herd=herd)
    e1.close()
    info = host.work_info(path)
    assert info["kangaroos"] == n and info["items"] == sv1.table_size and info["total_count"] == 3 * n * steps

    sv2 = host.Solver(lo, hi, pub)
    px, py, d = sv2.load_work(path, n)  # LoadWork + FetchWalks
    assert len(px) == n and sv2.table_size == sv1.table_size
    e2 = engine()
    e2.SetKangaroos(px, py, d)  # core/kangaroo.cpp:438
    e2.callKernel()
    second = [e2.Launch(), e2.Launch(), e2.Collect()]
    for it in second:
        sv2.add_items(it)
    gx, gy, gd = e2.GetKangaroos()
    e2.close()

    ox, oy, od = case["px"].copy(), case["py"].copy(), case["d"].copy()
    ref = oracle_dps(case, ox, oy, od, 6 * steps)
    got = sorted(t for it in first + second for t in dp_multiset(it))
    assert len(ref) > 10000 and got == ref
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
    # the table after save + load + second half == the table of all oracle DPs inserted in one go
    sv3 = host.Solver(lo, hi, pub)
    for k, x, dd in ref:
        sv3.add(x, dd, k)
    # (entries that went through the file keep 128 bits of x, as the reference's format does, memory/hashtable.cpp:13-21)
    low = lambda dump: sorted((h, x & ((1 << 128) - 1), dd) for h, x, dd in dump)
    if sv3.found is None and sv2.found is None:
        assert low(sv2.dump()) == low(sv3.dump())


def test_interrupted_solve_resumes_and_finds_key(tmp_path):
    """-w / -i end to end on the GPU: run 1 is stopped right after its first save (table + herd on disk, through
    GetKangaroos while a launch is in flight), run 2 loads the file and recovers the key."""
    lo = 1 << 70
    priv = lo + 0xB3C1D2E7F0A1
    pub = K.ec_mul_g(priv)
    path = str(tmp_path / "w.kg")
    key, st1 = solve.solve_key(lo, lo + (1 << 48) - 1, pub, grid=(16, 32), dp_bits=8, seed=3, nb_run=64, work_file=path,
                               save_period=0.0, stop_after_saves=1)
    assert key is None and st1["saves"] == 1
    info = host.work_info(path)
    assert info["kangaroos"] == 128 * 16 * 32 and info["items"] == st1["table"] and info["range_start"] == lo
    key, st2 = solve.solve_key(lo, lo + (1 << 48) - 1, pub, grid=(16, 32), dp_bits=8, seed=4, resume_from=path,
                               max_launches=3000)
    assert key == priv and st2["table"] >= st1["table"]


def test_72bit_solve_two_gpu_threads_one_table():
    """Config 3 in miniature: independent herds on two engines (two host threads, here both on device 0 as the
    reference's `-gpuId 0,0` would), ONE shared table, 2^23 kangaroos in all, DP size from the reference's formula
    (72/2 - 23 - 1 = 12).  About 2^37.3 jumps expected."""
    priv = 0x9F3C5A7E1B2D4C6F81 % (1 << 72)
    pub = K.ec_mul_g(priv)
    assert solve.suggested_dp(72, 2 * 128 * 256 * 128) == 12
    found, st = solve.solve_key_threads(0, (1 << 72) - 1, pub, gpu_ids=(0, 0), grid=(256, 128), seed=21, max_launches=1500)
    print(st)
    assert found == priv and st["dp_bits"] == 12 and st["kangaroos"] == 1 << 23
    assert all(t["launches"] > 0 and t["dps"] > 0 for t in st["per_thread"])
    assert st["jumps"] < 64 * 2 * (1 << 36) + st["kangaroos"] * 256 * 8


def test_config5_multi_key_low_dp_two_threads_with_saves(tmp_path):
    """BASELINE config 5 in miniature, all at once: two keys in a row, DP = 8 (about one DP per kangaroo and launch),
    two GPU threads feeding one table, and -w saves with the herds (GetKangaroos on both engines, table + 262 144
    kangaroos on disk) while the search runs.  Nothing may be lost (a lost item raises), both keys must be found,
    and the last work file must describe the last key."""
    lo = 1 << 70
    bits = 58  # about 2^30 jumps per key: some 16 rounds of the two 131 072-kangaroo herds; the main loop polls every
    # 2 ms, so that the first save request goes out during the first launches
    privs = [lo + 0x0C0FFEE123456789 % (1 << bits), lo + 0x0DEADBEEF9876543 % (1 << bits)]
    pubs = [K.ec_mul_g(p) for p in privs]
    path = str(tmp_path / "c5.kg")
    res = solve.solve_keys(lo, lo + (1 << bits) - 1, pubs, gpu_ids=(0, 0), grid=(8, 128), dp_bits=8, seed=31,
                           work_file=path, save_period=0.0, max_launches=4000, poll=0.002)
    assert [r[0] for r in res] == privs
    for _, st in res:
        print(st["saves"], st["launches"], st["dps"], st["table"], st["dead"], st["unstored"])
        assert st["saves"] >= 1 and st["kangaroos"] == 2 * 128 * 8 * 128
        # every DP either went into the table, marked a dead kangaroo, or is the one that solved the key / arrived after it
        assert st["table"] + st["dead"] + st["unstored"] == st["dps"] and st["table"] > 100000
        assert 1 <= st["unstored"] < 20000
        assert all(t["launches"] > 0 for t in st["per_thread"])
    info = host.work_info(path)
    assert info["dp_bits"] == 8 and info["kangaroos"] == 2 * 128 * 8 * 128 and info["items"] > 100000
    assert info["range_start"] == lo and info["range_end"] == lo + (1 << bits) - 1
    sv = host.Solver(lo, lo + (1 << bits) - 1, pubs[1])
    assert info["key"] == pubs[1] or info["key"] == sv.key  # the file carries the key as the reference writes it


def test_config5_default_grid_dp8_drain_without_loss():
    """BASELINE config 5's DP size on the engine's DEFAULT grid: 8 388 608 kangaroos at DP = 8 report about 8.4 M
    distinguished points per launch.  The caller's maxFound is the reference's 131 072 (core/kangaroo.cpp:408), with
    which the reference exits at the first launch (engine.cu:506-510); here the DP ring follows the DP mask
    (kg_dp_capacity).  Five launches: nothing lost, every item a DP, table + dead == DPs drained, and the kernel is not
    slowed down much by the reporting (round 2: -27 % at this DP size; now the walk kernel only chains the item to its
    kangaroo and the distances are filled in by the history kernel on the second stream)."""
    gx, gy = GPUEngine.GetGridSize(0)
    lo, hi = 0, (1 << 110) - 1
    pub = host.ec_mul_g(0x2545F4914F6CDD1D0123456789ABCDEF % hi)
    sv = host.Solver(lo, hi, pub)
    eng = GPUEngine(gx, gy, 0, 65536 * 2)
    n = eng.nbKangaroo
    assert n == 8388608
    jd, jx, jy, _ = host.create_jump_table(sv.range_power)
    eng.SetWildOffset(sv.wild_offset)
    eng.SetParams(solve.dp_mask_of(32), jd, jx, jy)
    eng.HerdFromDistances(host.herd_distances(host.Rng(17), sv.range_power, sv.range_width, n, 0), sv.key)
    eng.callKernel()  # two launches at DP = 32 first: the kernel time without DP traffic
    eng.Launch()
    eng.Collect()
    ms_quiet = eng.LastKernelMs()
    eng.SetParams(solve.dp_mask_of(8), jd, jx, jy)
    assert eng.DpCapacity() >= 2 * n
    rng = host.Rng(18)
    launches, n_dp, reseeded, ms, ms_sum = 5, 0, 0, [], []
    eng.callKernel()
    for i in range(launches):
        items = eng.Launch(copy=False) if i < launches - 1 else eng.Collect(copy=False)
        ms.append(eng.LastKernelMs())
        ms_sum.append(eng.LastSumKernelMs())
        assert eng.lost == 0
        assert not np.any(items["x"][:, 3] >> np.uint64(56))
        n_dp += len(items)
        _, dead = sv.add_items(items)
        for di in dead:
            k_idx = int(items[di]["kIdx"])
            nx, ny, nd = host.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, k_idx % 2)
            eng.SetKangaroo(k_idx, nx[0], ny[0], nd[0])
            reseeded += 1
    expect = launches * n
    assert abs(n_dp - expect) < 6 * expect ** 0.5
    assert sv.found is None and sv.unstored == 0
    assert sv.table_size + sv.dead == n_dp and sv.dead == reseeded
    some = items[np.random.default_rng(5).integers(0, len(items), 512)].copy()
    eng.close()
    hx, hy = host.ec_mul_g_batch(np.ascontiguousarray(some["d"]))
    for j, it in enumerate(some):
        pt = (host.to_int(hx[j]), host.to_int(hy[j]))
        if int(it["kIdx"]) & 1:
            pt = host.ec_add(pt, sv.key)
        assert pt[0] == host.to_int(it["x"])
    print(f"walk kernel: DP=32 {ms_quiet:.2f} ms, DP=8 {sorted(ms)[len(ms) // 2]:.2f} ms (8.4 M DPs per launch); history kernel "
          f"at DP=8 {sorted(ms_sum)[len(ms_sum) // 2]:.2f} ms, on the second stream underneath the next walk kernel")
    assert sorted(ms)[len(ms) // 2] < 1.12 * ms_quiet


@pytest.mark.timeout(700)
def test_config3_80bit_solve_two_gpu_threads_one_table():
    """BASELINE config 3 at its stated range on one GPU: an 80-bit known key, two independent herds on two engines (two
    host threads, both on device 0 as `-gpuId 0,0`; core/kangaroo.cpp:780-786), ONE shared table, 2^23 kangaroos in
    all, the DP size the reference's formula picks (80/2 - 23 - 1 = 16, core/kangaroo.cpp:742-744).  About 2^41.3
    jumps, two minutes."""
    priv = (0xD1B54A32D192ED03C0FFEE1234567890ABCDEF % (1 << 79)) | (1 << 79)
    pub = K.ec_mul_g(priv)
    assert solve.suggested_dp(80, 2 * 128 * 256 * 128) == 16
    found, st = solve.solve_key_threads(0, (1 << 80) - 1, pub, gpu_ids=(0, 0), grid=(256, 128), seed=11, max_launches=9000)
    print(st)
    assert found == priv and st["dp_bits"] == 16 and st["kangaroos"] == 1 << 23 and st["range_power"] == 80
    assert all(t["launches"] > 0 and t["dps"] > 0 for t in st["per_thread"])
    assert st["table"] + st["dead"] + st["unstored"] == st["dps"]
    assert st["jumps"] < 16 * 2 * (1 << 40)


@pytest.mark.timeout(500)
def test_config3_80bit_solve_at_dp20():
    """BASELINE config 3 with the DP size it names: an 80-bit known key at DP = 20.  With 20 DP bits the herd has to be
    small for the DP overhead (kangaroos x 2^20 jumps) to stay near the 2^41 jumps of the search itself: 2^20 kangaroos
    (grid 128 x 64), one engine.  2^40.7 jumps, 1.7 M DPs, 110 s at 16 G jumps/s (profiles/r03_solves.txt)."""
    priv = (0xD1B54A32D192ED03C0FFEE1234567890ABCDEF % (1 << 79)) | (1 << 79)
    pub = K.ec_mul_g(priv)
    found, st = solve.solve_key(0, (1 << 80) - 1, pub, grid=(128, 64), dp_bits=20, seed=11, max_launches=30000)
    print(st)
    assert found == priv and st["dp_bits"] == 20 and st["kangaroos"] == 1 << 20
    assert st["table"] + st["dead"] + st.get("unstored", 0) == st["dps"]
    assert st["jumps"] < 8 * 2 * (1 << 40)

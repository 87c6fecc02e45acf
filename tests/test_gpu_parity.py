"""GPU parity tests: the HIP engine, driven through the C ABI, against the oracle on the same seeded
herd and jump table.  Bit-exact: DP multiset (kIdx, x, distance), and the whole herd state."""
import os

import numpy as np
import pytest

import kgoracle as K
from gpu_common import dp_multiset, make_case, oracle_dps

pytestmark = pytest.mark.gpu


def _engine(groups, per_group, case, nb_run=None, k=None, mode=None, ring_max=0, ws_limit=0):
    from kangaroo_amd import GPUEngine

    e = GPUEngine(groups, per_group, 0, 65536 * 2, mode=mode, k_per_lane=k or 0, ring_max=ring_max, ws_limit=ws_limit)  # kg_create_tuned
    e.SetWildOffset(case["width"] >> 1)  # core/kangaroo.cpp:436
    e.SetParams(case["dp_mask"], case["jd"], case["jx"], case["jy"])
    if nb_run:
        e.SetNbRun(nb_run)
    e.SetKangaroos(case["px"], case["py"], case["d"])
    return e


# (kernel form, kangaroos per lane): the shipped register-resident instantiations and the streaming kernel at several
# herd depths; 64 per lane leaves half of the only workgroup padding
VARIANTS = [("reg", 16), ("stream", 2), ("stream", 4), ("stream", 6), ("stream", 10), ("stream", 16),
            ("stream", 64)]


@pytest.mark.parametrize("mode,k", VARIANTS)
def test_dp_stream_and_state_match_oracle(mode, k):
    """BASELINE config 2 in miniature: 64-bit range, same seed -> same DPs as the CPU walk.
    Odd and even step counts per launch exercise both pass directions of the fused pass."""
    case = make_case(64, 128 * 2 * 32, seed=1000 + k, dp_bits=6)
    e = _engine(2, 32, case, nb_run=64, k=k, mode=mode)
    assert e.KernelInfo()["k_per_lane"] in (k, -(-k // 4) * 4)  # the streaming kernel rounds up to its group (pairs or quads)
    assert ("stream" in e.KernelName()) == (mode == "stream")
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    for launch in range(3):
        got = dp_multiset(e.Launch())
        ref = oracle_dps(case, px, py, d, 64)
        assert len(ref) > 50
        assert got == ref, f"launch {launch}"
    # the 4th launch is in flight; collect it, then an odd-length launch
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 64)
    e.SetNbRun(7)
    e.callKernel()
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 7)
    e.SetNbRun(1)
    e.callKernel()
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 1)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()


@pytest.mark.parametrize("mode,groups,per_group,k", [("stream", 3, 7, None), ("stream", 1, 1, None), ("stream", 5, 3, 6),
                                                     ("reg", 3, 7, 16), ("reg", 1, 1, 16)])
def test_ragged_herd_sizes(mode, groups, per_group, k):
    """Herds that do not fill the last workgroup (2688, 128, 1920 kangaroos): padding lanes walk a copy of
    kangaroo 0 and must never report; every real kangaroo matches the oracle."""
    n = 128 * groups * per_group
    case = make_case(64, n, seed=500 + n, dp_bits=4)
    e = _engine(groups, per_group, case, nb_run=33, k=k, mode=mode)
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    got = e.Launch()
    assert int(got["kIdx"].max()) < n
    assert dp_multiset(got) == oracle_dps(case, px, py, d, 33)
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 33)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()


@pytest.mark.parametrize("mode", ["stream", "reg"])
def test_default_nb_run_256_and_125_bit_range(mode):
    """NB_RUN = 256 (constants.h:21) with the benchmark's 125-bit jump table (3-word distance delta)."""
    case = make_case(125, 128 * 1 * 64, seed=77, dp_bits=8)
    e = _engine(1, 64, case, mode=mode)
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    got = dp_multiset(e.Collect())
    ref = oracle_dps(case, px, py, d, 256)
    assert got == ref and len(ref) > 1000
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()


@pytest.mark.parametrize("mode", ["stream", "reg"])
def test_wide_jump_distances_use_full_width_delta(mode):
    """Jump distances wider than 96 bits select the 8-word delta kernel; distances stay exact."""
    case = make_case(250, 128 * 1 * 64, seed=5, dp_bits=5)
    e = _engine(1, 64, case, nb_run=32, mode=mode)
    assert e.KernelInfo()["delta_words"] == 8
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    got = dp_multiset(e.Collect())
    ref = oracle_dps(case, px, py, d, 32)
    assert got == ref and len(ref) > 100
    e.close()


@pytest.mark.parametrize("mode", ["stream", "reg"])
def test_set_kangaroo_between_launches(mode):
    """SetKangaroo (dead-kangaroo reset, core/kangaroo.cpp:484-485) lands before the next launch."""
    case = make_case(64, 128 * 1 * 64, seed=9, dp_bits=4)
    e = _engine(1, 64, case, nb_run=16, mode=mode)
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    items = e.Launch()  # launch 2 already queued
    assert dp_multiset(items) == oracle_dps(case, px, py, d, 16)
    # replace kangaroos 5 (wild) and 8 (tame) with fresh ones while launch 2 is in flight
    nx, ny, nd = K.create_herd(case["rng"], 64, case["width"], case["key"], 2, 1)
    e.SetKangaroo(5, nx[0], ny[0], nd[0])
    e.SetKangaroo(8, nx[1], ny[1], nd[1])
    ref2 = oracle_dps(case, px, py, d, 16)  # launch 2 ran on the old herd
    assert dp_multiset(e.Launch()) == ref2
    px[5], py[5], d[5] = nx[0], ny[0], nd[0]
    px[8], py[8], d[8] = nx[1], ny[1], nd[1]
    assert dp_multiset(e.Launch()) == oracle_dps(case, px, py, d, 16)  # launch 3 sees the new ones
    e.close()


@pytest.mark.parametrize("mode", ["stream", "reg"])
def test_dp_zero_bits_every_point_and_overflow_report(mode):
    """dpSize = 0: every jump is a DP (core/kangaroo.cpp:110-120).  The DP ring follows the configuration: 262 144
    items per launch fit although the caller's maxFound is 131 072 (the reference would exit, engine.cu:506-510).
    With the ring capped at that maxFound the overflow is reported, not hidden."""
    from kangaroo_amd import KgError

    case = make_case(64, 128 * 1 * 64, seed=3, dp_bits=0)
    e = _engine(1, 64, case, nb_run=2, mode=mode)
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 2)
    assert len(oracle_dps(case, px.copy(), py.copy(), d.copy(), 1)) == 8192
    e.SetNbRun(32)  # 8192 * 32 = 262144 DPs per launch: the ring grows to hold them (twice that + slack)
    assert e.DpCapacity() >= 2 * 262144
    e.callKernel()
    got = e.Launch()  # the second launch is queued into the other buffer
    assert e.lost == 0 and len(got) == 262144
    assert dp_multiset(got) == oracle_dps(case, px, py, d, 32)
    got = e.Collect()
    assert e.lost == 0 and dp_multiset(got) == oracle_dps(case, px, py, d, 32)
    e.close()

    c = _engine(1, 64, case, nb_run=32, mode=mode, ring_max=131072)
    assert c.DpCapacity() == 131072
    c.callKernel()
    with pytest.raises(KgError) as ei:
        c.Collect()
    assert "items lost" in str(ei.value) and c.lost == 8192 * 32 - 131072
    c.close()


def test_two_engines_in_one_process_from_two_threads():
    """The reference runs one GPUEngine per host thread (core/kangaroo.cpp:780-786); two engines on one device,
    driven concurrently, must not disturb each other (own stream, own buffers, per-thread error state)."""
    import threading

    results = {}

    def run(tag, seed, mode):
        case = make_case(64, 128 * 1 * 64, seed=seed, dp_bits=5)
        e = _engine(1, 64, case, nb_run=48, mode=mode)
        px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
        ok = True
        e.callKernel()
        for _ in range(3):
            ok &= dp_multiset(e.Launch()) == oracle_dps(case, px, py, d, 48)
        ok &= dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 48)
        gx, gy, gd = e.GetKangaroos()
        ok &= np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
        e.close()
        results[tag] = ok

    th = [threading.Thread(target=run, args=("a", 21, "stream")), threading.Thread(target=run, args=("b", 22, "stream"))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert results == {"a": True, "b": True}


def test_full_size_bench_herd_sampled_walks_and_point_distance_invariant():
    """BASELINE config 2 at full size: 512 x 128 grid = 8 388 608 kangaroos, NB_RUN = 256, 125-bit jump table.
    Every kangaroo walks alone, so re-walking a sample of them with the oracle pins the launch at any herd size:
    final x, y, distance and every reported DP of the sampled kangaroos, bit for bit.  Over the whole herd the
    size-independent invariant is P_i = d_i * G (the synthetic herd starts on it and every jump keeps it), checked on
    a second, larger sample with the host EC code, and every DP anyone reported must satisfy the mask."""
    from kangaroo_amd import GPUEngine, host

    e = GPUEngine(512, 128, 0, 1 << 17)
    n = e.nbKangaroo
    assert n == 8388608 and "stream" in e.KernelName()
    threads = n // e.KernelInfo()["k_per_lane"]
    px, py, d = host.synthetic_herd(n)
    jd, jx, jy, _ = host.create_jump_table(125)
    dp_mask = 0xFFFF000000000000  # 16 bits: ~32768 DPs per launch
    e.SetWildOffset(((1 << 125) - 1) >> 1)
    e.SetParams(dp_mask, jd, jx, jy)
    e.SetKangaroos(px, py, d)
    e.callKernel()
    items = np.concatenate([e.Launch(), e.Collect()])  # two launches: both pass directions, state carried over
    gx, gy, gd = e.GetKangaroos()
    e.close()

    rng = np.random.default_rng(12345)
    edges = [0, 1, 63, 64, 255, 256, threads - 1, threads, threads + 1, 2 * threads - 1, n // 2 - 1, n // 2, n - threads,
             n - 257, n - 256, n - 65, n - 64, n - 2, n - 1]
    sample = np.unique(np.concatenate([np.array(edges), rng.integers(0, n, 300)])).astype(np.int64)
    sx, sy, sd = px[sample].copy(), py[sample].copy(), d[sample].copy()
    ref = K.walk(sx, sy, sd, 512, dp_mask, jd, jx, jy, cap=1 << 16)
    assert np.array_equal(gx[sample], sx) and np.array_equal(gy[sample], sy) and np.array_equal(gd[sample], sd)
    ref_dps = sorted((int(sample[k]), x, dd) for _, k, x, dd in ref)
    pos = {int(k): True for k in sample}
    got_dps = [t for t in dp_multiset(items[np.isin(items["kIdx"], sample)]) if t[0] in pos]
    assert got_dps == ref_dps

    # all DPs: mask, index range, plausible count (binomial, p = 2^-16 over 2 * 256 * n jumps)
    assert int(items["kIdx"].max()) < n
    assert not np.any(items["x"][:, 3] & np.uint64(dp_mask))
    expect = 2 * 256 * n / 65536
    assert abs(len(items) - expect) < 6 * expect ** 0.5
    # P = d * G on 4096 kangaroos and on 512 of the reported DPs
    chk = rng.integers(0, n, 4096)
    hx, _ = host.ec_mul_g_batch(gd[chk])
    assert np.array_equal(hx, gx[chk])
    some = items[rng.integers(0, len(items), 512)]
    hx, _ = host.ec_mul_g_batch(np.ascontiguousarray(some["d"]))
    assert np.array_equal(hx, some["x"])


def test_512_workgroups_every_kangaroo_matches_oracle():
    """Two workgroups per CU on every CU (512 workgroups, 2 kangaroos per lane, 262 144 kangaroos), every kangaroo
    compared.  This is the smallest herd that showed the wide-store data hazard fixed by ws_hold()
    (walk_kernel_stream.h): y of a few waves' lanes 12-15 mod 16 came back wrong after ONE step, and only in
    workgroups 256 and up, which no herd of the other tests reaches."""
    from kangaroo_amd import GPUEngine, host

    e = GPUEngine(16, 128, 0, 1 << 16, mode="stream", k_per_lane=2)
    kk = e.KernelInfo()["k_per_lane"]
    if kk != 2:  # a build whose kangaroos go in quads: the same 512 workgroups at 4 per lane
        e.close()
        e = GPUEngine(16 * kk // 2, 128, 0, 1 << 16, mode="stream", k_per_lane=kk)
    n = e.nbKangaroo
    assert n == 131072 * kk and e.KernelInfo()["grid"] == 512 and "stream" in e.KernelName()
    px, py, d = host.synthetic_herd(n)
    jd, jx, jy, _ = host.create_jump_table(125)
    dp_mask = 0xFFF0000000000000
    e.SetWildOffset(((1 << 125) - 1) >> 1)
    e.SetParams(dp_mask, jd, jx, jy)
    e.SetNbRun(5)
    e.SetKangaroos(px, py, d)
    e.callKernel()
    items = np.concatenate([e.Launch(), e.Launch(), e.Collect()])
    gx, gy, gd = e.GetKangaroos()
    e.close()
    ref = K.walk(px, py, d, 15, dp_mask, jd, jx, jy, cap=1 << 16)
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    assert dp_multiset(items) == sorted((k, x, dd) for _, k, x, dd in ref)


def test_nb_run_growth_and_arbitrary_dp_mask():
    """kg_set_nb_run may grow between launches (the history part of the workspace is reallocated), and SetParams takes
    any 64-bit mask on x.bits64[3] (the reference only ever builds top-bit masks, core/kangaroo.cpp:110-120): here a
    mask with bits in both 32-bit halves."""
    case = make_case(64, 128 * 1 * 64, seed=808, dp_bits=0)
    case["dp_mask"] = 0x7000000030000000  # 5 bits: one DP in 32 jumps
    e = _engine(1, 64, case, nb_run=8, mode="stream")
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 8)
    e.SetNbRun(96)
    e.callKernel()
    got = dp_multiset(e.Launch())
    ref = oracle_dps(case, px, py, d, 96)
    assert got == ref and len(ref) > 10000
    e.SetNbRun(200)  # takes effect with the launch after the one already queued
    assert dp_multiset(e.Launch()) == oracle_dps(case, px, py, d, 96)
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 200)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()


def test_dp_mask_lowered_while_a_launch_is_in_flight_grows_the_ring():
    """SetParams with a much lower DP size while a launch is in flight: the launch under way keeps the old mask, the DP
    ring of the launch queued next is grown at that kick (the other buffer still holds the launch in flight), nothing is
    lost and every launch equals the oracle's walk with the mask that was current when it was queued."""
    case = make_case(64, 128 * 4 * 64, seed=515, dp_bits=12)
    hi_mask = case["dp_mask"]
    lo_mask = 0xC000000000000000  # DP size 2: 8192 * ... a quarter of all points
    e = _engine(4, 64, case, nb_run=24, mode="stream")
    n = 128 * 4 * 64
    cap0 = e.DpCapacity()
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()                      # launch 1, old mask
    r1 = e.Launch()                     # launch 2 queued (old mask), launch 1 returned
    assert dp_multiset(r1) == oracle_dps(case, px, py, d, 24)
    e.SetParams(lo_mask, case["jd"], case["jx"], case["jy"])  # waits for launch 2, then the mask changes
    assert e.DpCapacity() >= 2 * n * 24 // 4 > cap0
    r2 = e.Launch()                     # launch 3 queued with the new mask into a ring grown now; launch 2 returned
    assert dp_multiset(r2) == oracle_dps(case, px, py, d, 24)
    case["dp_mask"] = lo_mask
    r3 = e.Launch()
    assert e.lost == 0 and abs(len(r3) - n * 24 / 4) < 6 * (n * 24 / 4) ** 0.5
    assert dp_multiset(r3) == oracle_dps(case, px, py, d, 24)
    r4 = e.Collect()
    assert e.lost == 0 and dp_multiset(r4) == oracle_dps(case, px, py, d, 24)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()


def test_launch_segments_equal_one_launch():
    """A herd whose streaming workspace exceeds what one kernel launch may address (4 GiB = 44.7 M kangaroos on the
    hardware; here kg_tuning.ws_limit_bytes forces it on 16 384 kangaroos) is walked in several launch segments per
    Launch: 12 kangaroos per lane as 3 segments of 4.  DPs (with their kIdx), the whole herd after three launches and a
    SetKangaroo in between must equal the oracle's, exactly as with one segment."""
    case = make_case(64, 128 * 2 * 64, seed=616, dp_bits=5)
    lanes = 1536  # ceil(16384 / 12) rounded up to 256
    e = _engine(2, 64, case, nb_run=40, k=12, mode="stream", ws_limit=4 * lanes * 96 + 1)
    assert e.KernelInfo()["k_per_lane"] == 12
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    e.callKernel()
    r1 = e.Launch()
    ref1 = oracle_dps(case, px, py, d, 40)
    assert dp_multiset(r1) == ref1 and len(ref1) > 10000
    # kangaroos of every segment report: kIdx covers the whole herd
    assert {int(k) * 3 // (128 * 2 * 64) for k in r1["kIdx"]} == {0, 1, 2}
    nx, ny, nd = K.create_herd(case["rng"], 64, case["width"], case["key"], 2, 1)
    e.SetKangaroo(7, nx[0], ny[0], nd[0])          # first segment
    e.SetKangaroo(16001, nx[1], ny[1], nd[1])      # last segment
    assert dp_multiset(e.Launch()) == oracle_dps(case, px, py, d, 40)  # launch 2 ran on the old herd
    px[7], py[7], d[7] = nx[0], ny[0], nd[0]
    px[16001], py[16001], d[16001] = nx[1], ny[1], nd[1]
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 40)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()
    with pytest.raises(Exception) as ei:  # a limit that cannot hold two kangaroos per lane
        _engine(2, 64, case, k=12, mode="stream", ws_limit=lanes * 96)
    assert "ws_limit_bytes" in str(ei.value)


def test_herd_beyond_one_descriptor_runs_in_segments():
    """67 108 864 kangaroos (6.4 GB of state; the streaming workspace of one launch ends at 44.7 M): two launch segments
    of 256 kangaroos per lane.  Round 2 ran such a herd on the spilling register-resident kernel.  Checked on a sample:
    start points from the GPU herd kernel, 24 steps, re-walked by the oracle kangaroo by kangaroo (a walk only depends
    on its own start), and every DP item of the sample."""
    from kangaroo_amd import GPUEngine, host

    e = GPUEngine(4096, 128, 0, 65536 * 2)
    n = e.nbKangaroo
    assert n == 1 << 26 and "stream" in e.KernelName() and e.KernelInfo()["k_per_lane"] == 512
    jd, jx, jy, _ = K.create_jump_table(125)
    width = (1 << 125) - 1
    key = K.ec_mul_g(0x0A5B3C1D2E7F9081726354453627180F % width)
    dist = host.herd_distances(host.Rng(99), 125, width, n, 0)
    e.SetWildOffset(width >> 1)
    dp_mask = 0xFFFC000000000000  # DP size 14
    e.SetParams(dp_mask, jd, jx, jy)
    e.SetNbRun(24)
    e.HerdFromDistances(dist, key)
    sx, sy, sd = e.GetKangaroos()
    rnd = np.random.default_rng(3)
    sample = np.unique(np.concatenate([rnd.integers(0, n, 3000), np.arange(0, 64), np.arange(n - 64, n),
                                       np.arange(n // 2 - 64, n // 2 + 64)]))
    ox, oy, od = (np.ascontiguousarray(a[sample]) for a in (sx, sy, sd))
    del sx, sy, sd
    e.callKernel()
    items = e.Collect()
    assert e.lost == 0 and abs(len(items) - n * 24 / 2 ** 14) < 6 * (n * 24 / 2 ** 14) ** 0.5
    ref = K.walk(ox, oy, od, 24, dp_mask, jd, jx, jy)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx[sample], ox) and np.array_equal(gy[sample], oy) and np.array_equal(gd[sample], od)
    want = sorted((int(sample[k]), x, dd) for _, k, x, dd in ref)
    pos = {int(v): i for i, v in enumerate(sample)}
    got = sorted(t for t in dp_multiset(items) if t[0] in pos)
    assert got == want
    # the two halves of the herd both report (kIdx below and above the segment boundary)
    assert (items["kIdx"] < n // 2).any() and (items["kIdx"] >= n // 2).any()
    e.close()

"""DP ingest: the sharded, threaded batch insert must leave exactly the table, the dead-kangaroo list
and the solution that inserting the same DPs one by one in arrival order leaves (the reference's order,
core/kangaroo.cpp:473-490)."""
import numpy as np

import kgoracle as K
from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE


def _items(n, seed, dup_every=7, same_herd_every=11):
    """Random DPs with planted duplicates and same-herd collisions (never a tame/wild collision)."""
    rnd = np.random.default_rng(seed)
    it = np.zeros(n, dtype=ITEM_DTYPE)
    it["x"] = rnd.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    it["d"] = rnd.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    it["d"][:, 2:] = 0
    it["kIdx"] = rnd.integers(0, 1 << 20, size=n, dtype=np.uint64)
    for i in range(dup_every, n, dup_every):        # exact duplicate of an earlier item
        it[i] = it[i - dup_every]
    for i in range(same_herd_every, n, same_herd_every):  # same x, same herd, other distance
        j = i - same_herd_every
        it["x"][i] = it["x"][j]
        it["kIdx"][i] = (it["kIdx"][j] & 1) | (int(it["kIdx"][i]) & ~1)
    return it


def test_threaded_batch_equals_arrival_order():
    pub = K.ec_mul_g(0x1234567)
    items = _items(50000, 3)
    a = host.Solver(0, (1 << 64) - 1, pub)
    a.set_threads(1)
    rc_a, dead_a = a.add_items(items)
    b = host.Solver(0, (1 << 64) - 1, pub)
    b.set_threads(6)
    rc_b, dead_b = b.add_items(items)
    c = host.Solver(0, (1 << 64) - 1, pub)
    for it in items:  # one by one
        c.add(host.to_int(it["x"]), host.to_int(it["d"]), int(it["kIdx"]))
    assert rc_a == rc_b == 0
    assert len(dead_a) > 5000 and np.array_equal(dead_a, dead_b)
    assert a.table_size == b.table_size == c.table_size
    assert a.dead == b.dead == c.dead == len(dead_a)
    assert a.dump() == b.dump() == c.dump()


def test_threaded_batch_finds_planted_key():
    """A tame and a wild DP with the same x and distances that add up to the key, buried in a large batch."""
    priv = 0xA5B3C1D2E7
    pub = K.ec_mul_g(priv)
    sv = host.Solver(0, (1 << 40) - 1, pub)
    sv.set_threads(4)
    items = _items(20000, 9)
    td = 0x3141592653
    wd = (td - priv) % K.N      # tame at td*G, wild at key + wd*G: same point when td = priv + wd
    x = K.ec_mul_g(td)[0]
    for pos, d, kidx in ((5000, td, 100), (15000, wd, 101)):
        items["x"][pos] = [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
        items["d"][pos] = [(d >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
        items["kIdx"][pos] = kidx
    rc, _ = sv.add_items(items)
    assert rc == 2 and sv.found == priv


def test_concurrent_producers_equal_one_by_one():
    """Eight producer threads (one per GPU in a solve, core/kangaroo.cpp:780-786) feed ONE table at the same time, each
    its own batches, through the solver's worker pool.  x values never repeat across producers here, so the final table
    is independent of the interleaving and must equal the one built item by item; every item is accounted for."""
    import threading

    pub = K.ec_mul_g(0x1234567)
    P, per_batch, batches = 8, 12000, 3
    data = []
    for p in range(P):
        mine = []
        for b in range(batches):
            it = _items(per_batch, 100 * p + b)
            it["x"][:, 3] = (it["x"][:, 3] & np.uint64(0x00FFFFFFFFFFFFFF)) | np.uint64((p * batches + b) << 56)  # disjoint x
            it["x"][:, 0] ^= np.uint64(p * batches + b)  # ... also in the 128 bits the table compares
            mine.append(it)
        data.append(mine)
    a = host.Solver(0, (1 << 64) - 1, pub)
    a.set_threads(4)
    deads = [[] for _ in range(P)]

    def feed(p):
        for it in data[p]:
            rc, dead = a.add_items(it)
            assert rc == 0
            deads[p].append(dead.copy())

    th = [threading.Thread(target=feed, args=(p,)) for p in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    c = host.Solver(0, (1 << 64) - 1, pub)
    c.set_threads(1)
    ref_dead = [[c.add_items(it)[1].copy() for it in data[p]] for p in range(P)]
    assert a.table_size == c.table_size and a.dead == c.dead and a.unstored == c.unstored == 0
    assert a.table_size + a.dead == P * per_batch * batches
    assert all(np.array_equal(x, y) for p in range(P) for x, y in zip(deads[p], ref_dead[p]))
    assert sorted(a.dump()) == sorted(c.dump())


def test_every_item_is_accounted_for_when_the_key_is_found():
    """table + dead + unstored == items handed in, also for the batch that solves the key and for what arrives
    after it (threaded path and one-by-one path)."""
    priv = 0xA5B3C1D2E7
    pub = K.ec_mul_g(priv)
    td = 0x3141592653
    wd = (td - priv) % K.N
    x = K.ec_mul_g(td)[0]
    for n, threads in ((20000, 4), (3000, 4), (20000, 1)):
        sv = host.Solver(0, (1 << 40) - 1, pub)
        sv.set_threads(threads)
        items = _items(n, 9)
        for pos, d, kidx in ((n // 4, td, 100), (3 * n // 4, wd, 101)):
            items["x"][pos] = [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
            items["d"][pos] = [(d >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
            items["kIdx"][pos] = kidx
        rc, _ = sv.add_items(items)
        assert rc == 2 and sv.found == priv
        assert sv.unstored >= 1 and sv.table_size + sv.dead + sv.unstored == n
        sv.add_items(_items(500, 10))       # after the key: nothing is stored any more
        assert sv.add(1, 2, 0) == 2
        assert sv.table_size + sv.dead + sv.unstored == n + 501


def test_table_keeps_every_dp_through_its_growth_steps():
    """5.6 M distinct DPs take every shard from its first 64 KiB array to one huge page and on to the first doubling of a
    mapped array (host_secp.cpp CellArray / Shard::grow); handed in a second time, every one of them must be found again
    (an exact duplicate: a dead kangaroo, nothing stored)."""
    pub = K.ec_mul_g(0x1234567)
    sv = host.Solver(0, (1 << 64) - 1, pub)
    sv.set_threads(8)
    per, batches = 1_400_000, 4
    data = []
    for b in range(batches):
        rnd = np.random.default_rng(1000 + b)
        it = np.zeros(per, dtype=ITEM_DTYPE)
        it["x"] = rnd.integers(0, 1 << 63, size=(per, 4), dtype=np.uint64)
        it["x"][:, 1] = (it["x"][:, 1] & np.uint64(0x0FFFFFFFFFFFFFFF)) | np.uint64(b << 60)  # distinct across batches
        it["d"][:, :2] = rnd.integers(0, 1 << 62, size=(per, 2), dtype=np.uint64)
        it["kIdx"] = rnd.integers(0, 1 << 20, size=per, dtype=np.uint64)
        data.append(it)
        rc, dead = sv.add_items(it)
        assert rc == 0 and len(dead) == 0
    assert sv.table_size == per * batches and sv.dead == 0
    for it in data:
        rc, dead = sv.add_items(it)
        assert rc == 0 and len(dead) == per and dead[0] == 0 and dead[-1] == per - 1
    assert sv.table_size == per * batches and sv.dead == per * batches and sv.unstored == 0

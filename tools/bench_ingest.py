#!/usr/bin/env python3
"""DP ingest rate of the host table: random DPs in batches of one launch's worth.
  python tools/bench_ingest.py [items_per_batch] [batches]                  one producer, 1..16 threads per batch
  python tools/bench_ingest.py [items_per_batch] [batches] --producers 1,2,4,8
      P producer threads (one per GPU in a solve, core/kangaroo.cpp:780-786) feed ONE table concurrently, each its own
      batches; the C call releases the GIL.  Reports the aggregate rate and checks table + dead == items."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
per = int(argv[0]) if len(argv) > 0 else 2_000_000
batches = int(argv[1]) if len(argv) > 1 else 8
producers = None
if "--producers" in sys.argv:
    producers = [int(v) for v in sys.argv[sys.argv.index("--producers") + 1].split(",")]
    argv = [a for a in argv if a != sys.argv[sys.argv.index("--producers") + 1]]
    per = int(argv[0]) if len(argv) > 0 else 1_000_000
    batches = int(argv[1]) if len(argv) > 1 else 4


def make(seed, count):
    rnd = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        it = np.zeros(per, dtype=ITEM_DTYPE)
        it["x"] = rnd.integers(0, 1 << 63, size=(per, 4), dtype=np.uint64)
        it["d"][:, :2] = rnd.integers(0, 1 << 62, size=(per, 2), dtype=np.uint64)
        it["kIdx"] = rnd.integers(0, 1 << 23, size=per, dtype=np.uint64)
        out.append(it)
    return out


pub = host.ec_mul_g(0x1234567)
cpus = os.cpu_count() or 1
if producers is None:
    data = make(1, batches)
    for threads in (1, 2, 4, 8, 16):
        if threads > cpus:
            break
        sv = host.Solver(0, (1 << 64) - 1, pub)
        sv.set_threads(threads)
        t0 = time.perf_counter()
        for it in data:
            sv.add_items(it)
        dt = time.perf_counter() - t0
        print(f"threads={threads:2d}: {per * batches / dt / 1e6:6.2f} M DPs/s  ({per} per batch, {batches} batches, table {sv.table_size})", flush=True)
else:
    threads = min(16, cpus)
    for P in producers:
        data = [make(100 + p, batches) for p in range(P)]
        sv = host.Solver(0, (1 << 64) - 1, pub)
        sv.set_threads(threads)
        warm = make(999, 1)[0][:20000]
        sv.add_items(warm)  # starts the pool outside the timed region
        go = threading.Barrier(P + 1)

        def feed(mine):
            go.wait()
            for it in mine:
                sv.add_items(it)

        th = [threading.Thread(target=feed, args=(data[p],)) for p in range(P)]
        for t in th:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        total = P * per * batches + 20000
        ok = sv.table_size + sv.dead + sv.unstored == total
        print(f"producers={P}: {P * per * batches / dt / 1e6:6.2f} M DPs/s aggregate  ({per} per batch, {batches} batches each, "
              f"{threads} threads per batch incl. the producer, table {sv.table_size}, dead {sv.dead}, accounted {ok})", flush=True)
        if not ok:
            sys.exit(1)

#!/usr/bin/env python3
"""DP ingest rate of the host table: random DPs in batches of one launch's worth, inserted with
1..T worker threads.  Usage: python tools/bench_ingest.py [items_per_batch] [batches]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE

per = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rnd = np.random.default_rng(1)
data = []
for b in range(batches):
    it = np.zeros(per, dtype=ITEM_DTYPE)
    it["x"] = rnd.integers(0, 1 << 63, size=(per, 4), dtype=np.uint64)
    it["d"][:, :2] = rnd.integers(0, 1 << 62, size=(per, 2), dtype=np.uint64)
    it["kIdx"] = rnd.integers(0, 1 << 23, size=per, dtype=np.uint64)
    data.append(it)
pub = host.ec_mul_g(0x1234567)
for threads in (1, 2, 4, 8, 16):
    if threads > (os.cpu_count() or 1):
        break
    sv = host.Solver(0, (1 << 64) - 1, pub)
    sv.set_threads(threads)
    t0 = time.perf_counter()
    for it in data:
        sv.add_items(it)
    dt = time.perf_counter() - t0
    print(f"threads={threads:2d}: {per * batches / dt / 1e6:6.2f} M DPs/s  ({per} per batch, {batches} batches, table {sv.table_size})", flush=True)

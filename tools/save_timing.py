#!/usr/bin/env python3
"""Time of Kangaroo::SaveWork's counterpart (kgh_work_save) and of loading the file back, for a table of N x 8.4 M DPs.
  python tools/save_timing.py [batches] [dir ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dirs = sys.argv[2:] or ["/dev/shm", "/tmp"]
per = 8388608
pub = host.ec_mul_g(0x1234567)
sv = host.Solver(0, (1 << 64) - 1, pub)
rnd = np.random.default_rng(1)
for _ in range(batches):
    it = np.zeros(per, dtype=ITEM_DTYPE)
    it["x"] = rnd.integers(0, 1 << 63, size=(per, 4), dtype=np.uint64)
    it["d"][:, :2] = rnd.integers(0, 1 << 62, size=(per, 2), dtype=np.uint64)
    it["kIdx"] = rnd.integers(0, 1 << 23, size=per, dtype=np.uint64)
    sv.add_items(it)
print(f"table {sv.table_size}", flush=True)
for d in dirs:
    path = os.path.join(d, "kg_save_timing.bin")
    t0 = time.perf_counter()
    sv.save_work(path, 8, total_count=1, total_time=1.0)
    t1 = time.perf_counter()
    size = os.path.getsize(path)
    info = host.work_info(path)
    t2 = time.perf_counter()
    sv2 = host.Solver(0, (1 << 64) - 1, pub)
    sv2.load_work(path)
    t3 = time.perf_counter()
    os.unlink(path)
    print(f"{d}: save {t1 - t0:.2f} s ({size / 1e6:.0f} MB, {size / 1e6 / (t1 - t0):.0f} MB/s), info {t2 - t1:.2f} s, load {t3 - t2:.2f} s, "
          f"items {info['items']}, loaded table {sv2.table_size}", flush=True)
    del sv2

#!/usr/bin/env python3
"""Per-batch ingest rate of the host DP table while it grows: batches of one launch's worth of DPs at
BASELINE config 5's DP size on the default grid (8 388 608 items), one producer.  Shows what table growth costs.
  python tools/ingest_batches.py [batches] [threads]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE
per=8388608; batches=int(sys.argv[1]) if len(sys.argv)>1 else 6
rnd=np.random.default_rng(1)
data=[]
for _ in range(batches):
    it=np.zeros(per,dtype=ITEM_DTYPE)
    it["x"]=rnd.integers(0,1<<63,size=(per,4),dtype=np.uint64)
    it["d"][:,:2]=rnd.integers(0,1<<62,size=(per,2),dtype=np.uint64)
    it["kIdx"]=rnd.integers(0,1<<23,size=per,dtype=np.uint64)
    data.append(it)
pub=host.ec_mul_g(0x1234567)
sv=host.Solver(0,(1<<64)-1,pub); sv.set_threads(int(sys.argv[2]) if len(sys.argv) > 2 else min(16, os.cpu_count() or 1))
tot = 0.0
for it in data:
    t0=time.perf_counter(); sv.add_items(it); dt=time.perf_counter()-t0
    tot += dt
    print(f"batch {per/dt/1e6:.1f} M/s  {dt*1e3:.0f} ms table {sv.table_size}",flush=True)
print(f"all {per*batches/tot/1e6:.1f} M/s")

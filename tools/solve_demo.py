#!/usr/bin/env python3
"""Known-key solve (BASELINE configs 2/3).  One GPU: python tools/solve_demo.py <range_bits> [dp_bits] [gx gy]
GPU_IDS=0,1,...: one host thread + engine per listed device on ONE shared table (the reference's shape; an id may repeat).
N GPUs of one node, one process per GPU, DP stream all-gathered and the DP table sharded over the ranks:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/solve_demo.py <range_bits> ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kangaroo_amd import host, solve

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dp = int(sys.argv[2]) if len(sys.argv) > 2 else None
grid = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
priv = (0xD1B54A32D192ED03C0FFEE1234567890ABCDEF % (1 << (bits - 1))) | (1 << (bits - 1))
pub = host.ec_mul_g(priv)
last = [time.time()]


def progress(launches, n_dp, table):
    if time.time() - last[0] > 20:
        last[0] = time.time()
        print(f"  launch {launches}: {n_dp} DPs, table {table}", flush=True)


t0 = time.time()
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    found, st = solve.solve_key_multi(0, (1 << bits) - 1, pub, grid=grid, dp_bits=dp, seed=11, progress=progress)
    if int(os.environ.get("RANK", "0")) != 0:
        sys.exit(0 if found == priv else 1)
    st["table"], st["dead"] = st["table_this_rank"], st["dead_this_rank"]
elif os.environ.get("GPU_IDS"):
    ids = tuple(int(v) for v in os.environ["GPU_IDS"].split(","))
    found, st = solve.solve_key_threads(0, (1 << bits) - 1, pub, gpu_ids=ids, grid=grid, dp_bits=dp, seed=11, progress=progress)
else:
    found, st = solve.solve_key(0, (1 << bits) - 1, pub, grid=grid, dp_bits=dp, seed=11, progress=progress)
dt = time.time() - t0
print(f"range 2^{bits}: key {'FOUND' if found == priv else 'NOT FOUND'} = {found:#x}" if found else "not found")
print(f"  {st['device']}: kangaroos {st['kangaroos']}, DP bits {st['dp_bits']}, launches {st['launches']}, "
      f"jumps 2^{__import__('math').log2(st['jumps']):.2f}, DPs {st['dps']}, table {st['table']}, dead {st['dead']}")
print(f"  walk {st['seconds']:.1f} s ({st['jumps'] / st['seconds'] / 1e9:.2f} G jumps/s incl. host ingest), total {dt:.1f} s")
sys.exit(0 if found == priv else 1)

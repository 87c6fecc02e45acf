#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: low DP bits (default 8) so that about one DP per kangaroo arrives per launch;
reports jumps/s, DPs/s drained from the pinned ring, table insert rate, table size and lost items (must be 0).
Usage: python tools/stress_dp.py [dp_bits] [launches] [gx] [gy]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kangaroo_amd import GPUEngine, host, solve

dp_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gx = int(sys.argv[3]) if len(sys.argv) > 3 else 128
gy = int(sys.argv[4]) if len(sys.argv) > 4 else 128
lo, hi = 0, (1 << 110) - 1                       # far too wide to be solved: the table only grows
pub = host.ec_mul_g(0x2545F4914F6CDD1D0123456789ABCDEF % hi)
sv = host.Solver(lo, hi, pub)
eng = GPUEngine(gx, gy, 0, 65536 * 2)
n = eng.nbKangaroo
d = host.herd_distances(host.Rng(7), sv.range_power, sv.range_width, n, 0)
jd, jx, jy, _ = host.create_jump_table(sv.range_power)
eng.SetWildOffset(sv.wild_offset)
eng.SetParams(solve.dp_mask_of(dp_bits), jd, jx, jy)
eng.HerdFromDistances(d, sv.key)
rng = host.Rng(8)
eng.callKernel()
eng.Launch()  # warm-up launch, its DPs are dropped
t0 = time.perf_counter()
n_dp = 0
t_ins = t_launch = 0.0
dead_total = 0
kernel_ms, sum_ms = [], []
for i in range(launches):
    t1 = time.perf_counter()
    items = eng.Launch(copy=False)
    t_launch += time.perf_counter() - t1
    kernel_ms.append(eng.LastKernelMs())
    sum_ms.append(eng.LastSumKernelMs())
    n_dp += len(items)
    t1 = time.perf_counter()
    rc, dead = sv.add_items(items)
    t_ins += time.perf_counter() - t1
    dead_total += len(dead)
    for di in dead[:64]:  # re-seed (bounded: the point here is the drain rate)
        k_idx = int(items[di]["kIdx"])
        nx, ny, nd = host.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, k_idx % 2)
        eng.SetKangaroo(k_idx, nx[0], ny[0], nd[0])
dt = time.perf_counter() - t0  # `launches` launches were awaited and ingested in the loop; the one still in flight is not timed
eng.Collect()
print(f"{eng.deviceName} {eng.KernelName()} kangaroos={n} dp_bits={dp_bits} launches={launches}")
print(f"jumps/s (wall, host ingest in the loop) : {n * 256 * launches / dt / 1e9:.2f} G")
print(f"walk kernel ms per launch               : median {sorted(kernel_ms)[len(kernel_ms) // 2]:.2f}, min {min(kernel_ms):.2f}, max {max(kernel_ms):.2f}")
print(f"history kernel ms (2nd stream, overlaps) : median {sorted(sum_ms)[len(sum_ms) // 2]:.2f}")
print(f"wall ms per launch                       : {1e3 * dt / launches:.2f}")
print(f"DPs drained                             : {n_dp} = {n_dp / dt / 1e6:.2f} M/s, lost {eng.lost}")
print(f"Launch call (wait + item unpack)        : {1e3 * t_launch / launches:.2f} ms per launch; add_items {1e3 * t_ins / launches:.2f} ms per launch")
print(f"table insert rate (add_items only)      : {n_dp / t_ins / 1e6:.2f} M/s, table {sv.table_size}, dead {dead_total}")
eng.close()

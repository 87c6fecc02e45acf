#!/usr/bin/env python3
"""BASELINE config 5 rehearsed on ONE GPU: a multi-key input (three keys in a row), DP = 8, EIGHT GPU threads — eight
engines, all on device 0 as `-gpuId 0,0,0,0,0,0,0,0` would make them (core/kangaroo.cpp:780-786) — feeding ONE shared
table, and -w saves with the herds every few seconds while the search runs.  The eight engines share one GPU, so the
jump rate is one GPU's; what the run shows is the host path of an eight-GPU node (eight producers, the save handshake
with eight parked threads, the key loop) with real engines.  Usage: python tools/config5_demo.py [bits] [threads] [gx gy]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kangaroo_amd import host, solve

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 66
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
grid = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16, 128)
lo = 1 << 90
privs = [lo + v % (1 << bits) for v in (0x0C0FFEE1234567890ABCDEF1, 0x0DEADBEEF9876543210FEDCB, 0x0FEEDFACECAFEBEEF0123457)]
pubs = [host.ec_mul_g(p) for p in privs]
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "c5.kg")
    t0 = time.time()
    res = solve.solve_keys(lo, lo + (1 << bits) - 1, pubs, gpu_ids=(0,) * threads, grid=grid, dp_bits=8, seed=31,
                           work_file=path, save_period=float(os.environ.get("SAVE_PERIOD", "0.3")), max_launches=200000)
    dt = time.time() - t0
    info = host.work_info(path)
    size = os.path.getsize(path)
ok = [r[0] for r in res] == privs
print(f"config 5 on one GPU: {len(pubs)} keys of 2^{bits}, DP = 8, {threads} GPU threads x {128 * grid[0] * grid[1]} kangaroos on one table: "
      f"{'ALL KEYS FOUND' if ok else 'FAILED'} in {dt:.1f} s")
for (found, st), priv in zip(res, privs):
    acc = st["table"] + st["dead"] + st["unstored"] == st["dps"]
    print(f"  key {found:#x} ({'ok' if found == priv else 'WRONG'}): {st['seconds']:.1f} s, launches {st['launches']} "
          f"({'/'.join(str(t['launches']) for t in st['per_thread'])}), jumps 2^{__import__('math').log2(st['jumps']):.2f} "
          f"= {st['jumps'] / st['seconds'] / 1e9:.2f} G/s, DPs {st['dps']} = {st['dps'] / st['seconds'] / 1e6:.1f} M/s, "
          f"table {st['table']}, dead {st['dead']}, unstored {st['unstored']}, accounted {acc}, saves {st['saves']}")
    ok = ok and acc
print(f"  last work file: {size / 1e6:.1f} MB, DP bits {info['dp_bits']}, {info['items']} table entries, {info['kangaroos']} kangaroos")
sys.exit(0 if ok else 1)

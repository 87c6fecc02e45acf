#!/usr/bin/env python3
"""Full-size invariant stress: after `launches` launches every sampled kangaroo must satisfy P = d*G (synthetic
herd: tame-style points), and so must every DP of the last launch.  Host EC code of the product library.
Usage: python tools/stress_invariant.py <gx> <gy> <launches> <kangaroos to check> [stream|reg]   (env RANGE = range bits)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kangaroo_amd import GPUEngine, host
gx_, gy_, launches, nchk = (int(v) for v in sys.argv[1:5])
e = GPUEngine(gx_, gy_, 0, 1 << 17, mode=sys.argv[5] if len(sys.argv) > 5 else None)
n = e.nbKangaroo
print(e.KernelName(), e.KernelInfo(), flush=True)
px, py, d = host.synthetic_herd(n)
rp = int(os.environ.get("RANGE", "125")); jd, jx, jy, _ = host.create_jump_table(rp)
e.SetWildOffset(((1 << rp) - 1) >> 1)
e.SetParams(0xFFFF000000000000, jd, jx, jy)
e.SetKangaroos(px, py, d)
e.callKernel()
nit = 0
for i in range(launches - 1):
    it = e.Launch(); nit += len(it)
it = e.Collect(); nit += len(it)
gx, gy, gd = e.GetKangaroos()
e.close()
rng = np.random.default_rng(7)
chk = rng.integers(0, n, nchk)
t = time.time()
hx, hy = host.ec_mul_g_batch(np.ascontiguousarray(gd[chk]))
bad = np.any(hx != gx[chk], axis=1) | np.any(hy != gy[chk], axis=1)
print(f"launches={launches} DPs={nit} checked={nchk} kangaroos ({time.time()-t:.1f}s host)  P != d*G: {int(bad.sum())}", flush=True)
hx, _ = host.ec_mul_g_batch(np.ascontiguousarray(it["d"]))
print("last launch DPs with x != (d*G).x:", int(np.any(hx != it["x"], axis=1).sum()), "of", len(it))
sys.exit(1 if bad.any() else 0)

#!/usr/bin/env python3
"""SURVEY §8d throughput sweep on one GPU: kangaroos per GPU x NB_RUN, BASELINE config 4's workload (125-bit range,
DP mask of 32 bits).  Per point: `--seconds` of back-to-back launches, the median of the per-launch kernel rates (HIP
events on the engine's stream) and the wall rate.  Optional --k pins kangaroos per lane, --mode the kernel form.
Usage: python tools/sweep.py --log2n 20,21,22,23,24,25 --nb-run 64,256,1024 --seconds 30 --out gpurun_out/sweep.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kangaroo_amd import GPUEngine, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", default="20,21,22,23,24,25")
ap.add_argument("--nb-run", default="64,256,1024")
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--k", default="0", help="comma list of kangaroos per lane to try at every point (0 = engine's choice)")
ap.add_argument("--mode", default=None, choices=[None, "stream", "reg"])
ap.add_argument("--out", default="gpurun_out/sweep.json")
args = ap.parse_args()

jd, jx, jy, _ = host.create_jump_table(125)
width = (1 << 125) - 1
key = host.ec_mul_g(0x0A5B3C1D2E7F9081726354453627180F % width)
points = []
for lg in (int(v) for v in args.log2n.split(",")):
    n = 1 << lg
    gy = 128
    gx = n // (128 * gy)
    d = host.herd_distances(host.Rng(0x5EED0000 + lg), 125, width, n, 0)
    for k in (int(v) for v in args.k.split(",")):
        for nb_run in (int(v) for v in args.nb_run.split(",")):
            eng = GPUEngine(gx, gy, 0, 65536 * 2, mode=args.mode, k_per_lane=k)
            assert eng.nbKangaroo == n
            eng.SetWildOffset(width >> 1)
            eng.SetParams(0xFFFFFFFF00000000, jd, jx, jy)
            eng.SetNbRun(nb_run)
            eng.HerdFromDistances(d, key)
            eng.callKernel()
            eng.Launch()  # warm-up
            ms = []
            t0 = time.perf_counter()
            launches = 0
            while time.perf_counter() - t0 < args.seconds or launches < 3:
                eng.Launch()
                ms.append(eng.LastKernelMs())
                launches += 1
            eng.Collect()
            ms.append(eng.LastKernelMs())
            launches += 1
            wall = time.perf_counter() - t0
            rates = n * nb_run / (np.array(ms) * 1e-3)
            pt = {"log2_kangaroos": lg, "kangaroos": n, "nb_run": nb_run, "kernel": eng.KernelName(), **eng.KernelInfo(),
                  "launches": launches, "seconds": wall, "median_kernel_jumps_per_s": float(np.median(rates)),
                  "min_kernel_jumps_per_s": float(rates.min()), "max_kernel_jumps_per_s": float(rates.max()),
                  "wall_jumps_per_s": n * nb_run * launches / wall, "memory_bytes": eng.GetMemory()}
            points.append(pt)
            print(json.dumps(pt), flush=True)
            eng.close()
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            with open(args.out, "w") as f:
                json.dump({"device": pt["kernel"], "points": points}, f, indent=1)
    del d

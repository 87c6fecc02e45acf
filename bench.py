#!/usr/bin/env python3
"""bench.py — kangaroo jumps/s on N MI355X GPUs (BASELINE.json metric, config 4).

Workload (one GPU): range [0, 2^125-1] (rangePower 125 -> 63-bit jumps), jump table from the fixed seed
0x600DCAFE, DP mask of 32 bits (no solve expected, DP traffic negligible), default grid 2*CUs x 128 ->
8.4 M kangaroos per GPU, NB_RUN = 256.  A "step" is one GPUEngine::Launch = nbKangaroo * NB_RUN jumps.
The herd is a seeded CreateHerd herd for a synthetic key (distances on the host, points on the GPU) and is
resident in HBM before the timed region starts.  N > 1: one process per GPU, disjoint herds, no
collective in the jump path (weak scaling); torch.distributed is used for barrier / max only.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

MAD32_PER_JUMP = 416          # 5 modmul * 74 + 1 modsqr * 46, inversion excluded
# Peak of record (the measured micro-benchmark is the denominator): the best SUSTAINED v_mad_u64_u32 rate
# measured on this chip, 540.8 G wave-instructions/s over 0.7 s at two waves per SIMD and a measured 2382 MHz
# (profiles/r02_ubench_power.txt:4, tools/ubench_power.hip) = 34.6 T MAD32/s = 4.51 cycles per wave64 and SIMD.
PEAK_WAVE_MADS_PER_S = 540.8e9
PEAK_MAD32_PER_S = PEAK_WAVE_MADS_PER_S * 64
MAD_CYCLES_PER_WAVE = 1024 * 2.382e9 / PEAK_WAVE_MADS_PER_S   # 4.51
# Round 1's short run (1.1 ms, 4 waves per SIMD, assumed 2.4 GHz): 5.08 cycles -> 30.96 T MAD32/s; kept beside it, labelled
PEAK_MAD32_R01 = 1024 * 64 / 5.08 * 2.4e9
# v_mad_u64_u32 the kernel really issues per jump and lane with 9x29-bit limbs: 5.5 multiplications x 101 + 1 squaring x
# 65 + the shared inversion's share (one inversion + 9 multiplications per 4 waves x K jumps) ~ 5  (DESIGN.md section 4;
# static counts from tools/loop_mix.py, total VALU per jump from the PMC pass in profiles/)
MADS_ISSUED_PER_JUMP = 626
STATE_BYTES_PER_KANGAROO = 192  # 96 B loaded + 96 B stored per launch


def cpu_model():
    """Model string of the host CPU: "model name" of /proc/cpuinfo plus family / model / stepping (some hosts of the
    pool report a bare vendor string as the name) and the logical CPUs this process may use."""
    want = ("model name", "vendor_id", "cpu family", "model", "stepping")
    got = {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k = k.strip()
                if k in want and k not in got:
                    got[k] = v.strip()
                if len(got) == len(want):
                    break
    except OSError:
        pass
    if not got:
        return "unknown"
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 0
    return (f"{got.get('model name', '?')} ({got.get('vendor_id', '?')} family {got.get('cpu family', '?')} model "
            f"{got.get('model', '?')} stepping {got.get('stepping', '?')}; {usable} of {os.cpu_count()} logical CPUs usable)")


def cpu_baseline(seconds_all=30.0, seconds_one=5.0):
    """Oracle (CPU port of SolveKeyCPU's loop) on the host cores: 1024 kangaroos per thread as the
    reference's CPU_GRP_SIZE, same 125-bit jump table.  Bounded sample; reported, never the target.
    Two figures as SURVEY §8d asks: all the threads the box gives one GPU (16), >= 30 s, and one thread, >= 5 s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kgoracle as K

    cores = min(16, os.cpu_count() or 1)
    jd, jx, jy, _ = K.create_jump_table(125)
    rng = K.RNG(1)
    key = K.ec_mul_g(0xA5B3C1D2E7)
    width = (1 << 125) - 1
    herds = [K.create_herd(rng, 125, width, key, 1024, 0) for _ in range(cores)]
    mask = 0xFFFFFFFF00000000
    # calibrate (short), then the one-thread figure over >= seconds_one
    t0 = time.perf_counter()
    K.walk_raw(*herds[0], 60, mask, jd, jx, jy, 16)
    per_step = (time.perf_counter() - t0) / 60
    steps_one = max(200, int(seconds_one / per_step) + 1)
    t0 = time.perf_counter()
    K.walk_raw(*herds[0], steps_one, mask, jd, jx, jy, 16)
    dt_one = time.perf_counter() - t0
    one_thread = 1024 * steps_one / dt_one
    steps = max(200, int(seconds_all / (dt_one / steps_one) * 1.05) + 1)  # threads share cores' turbo: a little longer

    def work(h):
        K.walk_raw(h[0], h[1], h[2], steps, mask, jd, jx, jy, 16)

    th = [threading.Thread(target=work, args=(h,)) for h in herds]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    jumps = cores * 1024 * steps
    return {"value": jumps / dt, "unit": "jumps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "value_1thread": one_thread, "seconds": dt, "seconds_1thread": dt_one,
            "sample": f"{cores} threads x 1024 kangaroos x {steps} steps, 125-bit range, oracle/kg_oracle.c ({dt:.1f} s); "
                      f"1 thread: 1024 kangaroos x {steps_one} steps ({dt_one:.1f} s)"}


class PowerSampler:
    """Board power of the GPU this rank drives, sampled every 50 ms in a thread (read-only sysfs).  The hwmon entry is
    the one whose PCI device is the engine's (kg_device_pci_bus_id): DRM card order need not be HIP device order.  The
    jump kernel runs at the board's power cap; the line reports what was drawn."""

    def __init__(self, pci_bus_id=None):
        import glob

        self.path = self.cap = self.card = None
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            dev = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))  # .../0000:c1:00.0
            if pci_bus_id and dev.lower() != pci_bus_id.lower():
                continue
            for leaf in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(d, leaf)):
                    self.path = os.path.join(d, leaf)
                    self.card = dev
                    try:
                        self.cap = int(open(os.path.join(d, "power1_cap")).read()) * 1e-6
                    except (OSError, ValueError):
                        pass
                    break
            if self.path:
                break
        self.samples = []
        self._stop = threading.Event()
        self._th = None

    def start(self):
        if not self.path:
            return

        def run():
            while not self._stop.is_set():
                try:
                    self.samples.append(int(open(self.path).read()) * 1e-6)
                except (OSError, ValueError):
                    pass
                time.sleep(0.05)

        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        if self._th:
            self._stop.set()
            self._th.join()
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {"median_W": v[len(v) // 2], "max_W": v[-1], "cap_W": self.cap, "samples": len(v), "pci": self.card,
                "sysfs": self.path}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, before
    anything in this process has touched the GPU.  Rank 0's stdout (the JSON line) is relayed; the exit code is
    non-zero if any rank failed."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while ALL children are polled: a rank that dies early (GPU init failure,
    # import error) must end the run with a clear message, not leave the others waiting in a rendezvous or a barrier
    # until the distributed timeout
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    failed = False
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if not failed and any(rc not in (None, 0) for rc in rcs):
            failed = True
            for r, p in enumerate(procs):  # the survivors would wait for the dead rank: stop them (our own children)
                if rcs[r] is None:
                    p.terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    out = b"".join(c for c in out_chunks if c)
    for line in out.decode().splitlines():  # the JSON line to stdout, anything else a library printed to stderr
        print(line, file=sys.stdout if line.startswith("{") and not failed else sys.stderr, flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code; negative = stopped because another rank failed): {bad}", file=sys.stderr)
        return 1
    return 0


def plumbing_only(args, rank, world):
    """The N-rank flow without an engine: barrier, timed region, max over ranks, sum over ranks, one line."""
    from kangaroo_amd import dist

    gx, gy = (int(v) for v in args.grid.split(",")) if args.grid else (512, 128)
    n = 128 * gx * gy
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))  # ranks take different times: the slowest one must set the clock
    dist.barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0)
    total = dist.sum_over_ranks(n)
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no GPU work, not a measurement)", "valid": False, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "value": total * args.nb_run * args.steps / elapsed,
                          "jumps": total * args.nb_run * args.steps, "kangaroos_per_gpu": n, "elapsed_s": elapsed,
                          "scaling": "weak"}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=str, default="", help="x,y (kangaroos = 128*x*y per GPU); default engine grid")
    ap.add_argument("--nb-run", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default=None, choices=["stream", "reg"],
                    help="pin the kernel form (kg_create_tuned); default: the engine's choice (streaming)")
    ap.add_argument("--check", type=int, default=2048, help="kangaroos re-derived from their distance after the run")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 flow on a one-GPU box: gloo instead of RCCL, every rank on GPU 0")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work at all: every rank reports the jumps its herd WOULD make (a CPU test of the rank "
                         "launcher and the reductions; the line is marked invalid)")
    args = ap.parse_args()

    # N > 1 without a launcher (no WORLD_SIZE in the environment): become the launcher
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from kangaroo_amd import dist

    rank, local_rank, world = dist.init(backend="gloo" if (args.rehearse_one_gpu or args.plumbing_only) else None)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but {world} rank(s) were started", file=sys.stderr)
        sys.exit(2)
    gpu_id = 0 if args.rehearse_one_gpu else local_rank
    if args.plumbing_only:
        return plumbing_only(args, rank, world)

    from kangaroo_amd import GPUEngine, host

    gx, gy = (int(v) for v in args.grid.split(",")) if args.grid else (0, 0)
    gx, gy = GPUEngine.GetGridSize(gpu_id, gx, gy)
    eng = GPUEngine(gx, gy, gpu_id, 65536 * 2, mode=args.mode)
    n = eng.nbKangaroo

    # configuration: built on rank 0, broadcast once (6 KiB jump table + dpMask + wild offset)
    if rank == 0:
        jd, jx, jy, _ = host.create_jump_table(125)
        cfg = np.concatenate([jd.reshape(-1), jx.reshape(-1), jy.reshape(-1),
                              np.array([0xFFFFFFFF00000000], dtype=np.uint64), host.limbs(((1 << 125) - 1) >> 1)])
    else:
        cfg = np.zeros(64 * 4 * 3 + 1 + 4, dtype=np.uint64)
    cfg = dist.broadcast_config(cfg)
    jd, jx, jy = (cfg[i * 256:(i + 1) * 256].reshape(64, 4) for i in range(3))
    dp_mask = int(cfg[768])
    wild = cfg[769:773]

    # herd: CreateHerd's distances (core/kangaroo.cpp:531-548) from a per-rank seed, so the ranks' herds are
    # disjoint random samples; the points d*G (+ key for the wild half) are computed on the GPU
    width = (1 << 125) - 1
    key = host.ec_mul_g(0x0A5B3C1D2E7F9081726354453627180F % width)
    d = host.herd_distances(host.Rng(0x5EED0000 + rank), 125, width, n, 0)
    eng.SetWildOffset(wild)
    eng.SetParams(dp_mask, jd, jx, jy)
    eng.SetNbRun(args.nb_run)
    eng.HerdFromDistances(d, key)
    del d

    # warmup
    eng.callKernel()
    for _ in range(max(args.warmup - 1, 0)):
        eng.Launch()
    eng.Collect()

    import torch

    dist.barrier()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    kernel_ms, clock_mhz = [], []
    n_dp = 0
    try:
        pci = GPUEngine.PciBusId(gpu_id)
    except Exception:  # noqa: BLE001 — power evidence is optional
        pci = None
    power = PowerSampler(pci)  # every rank samples its own board
    power.start()
    t0 = time.perf_counter()
    eng.callKernel()
    for _ in range(args.steps - 1):
        n_dp += len(eng.Launch())
        kernel_ms.append(eng.LastKernelMs())
        clock_mhz.append(eng.LastKernelClockMHz())
    n_dp += len(eng.Collect())
    kernel_ms.append(eng.LastKernelMs())
    clock_mhz.append(eng.LastKernelClockMHz())
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    dist.barrier()
    elapsed = time.perf_counter() - t0
    board_power = power.stop()
    elapsed = dist.max_over_ranks(elapsed)
    total_kangaroos = dist.sum_over_ranks(n)
    # median board power of every rank's own GPU (0 where sysfs gave nothing)
    power_per_rank = [float(v[0]) * 1e-3 for v in
                      dist.all_gather_array(np.array([int(1e3 * board_power["median_W"]) if board_power else 0], dtype=np.uint64))]

    # self-check, outside the timed region: a sample of the herd must still satisfy P_i = d_i*G (+ key for the
    # wild, odd-indexed half) after all the jumps; host EC code of the product library, not the oracle
    gxs, gys, gds = eng.GetKangaroos()
    sample = np.random.default_rng(1234 + rank).integers(0, n, args.check)
    hx, hy = host.ec_mul_g_batch(np.ascontiguousarray(gds[sample]))
    bad = 0
    for j, i in enumerate(sample):
        pt = (host.to_int(hx[j]), host.to_int(hy[j]))
        if i & 1:
            pt = host.ec_add(pt, key)
        bad += pt != (host.to_int(gxs[i]), host.to_int(gys[i]))
    del gxs, gys, gds
    bad = dist.sum_over_ranks(int(bad))
    if bad:
        raise RuntimeError(f"bench self-check failed: {bad} of {args.check * world} sampled kangaroos are off their track")

    jumps = total_kangaroos * args.nb_run * args.steps
    value = jumps / elapsed
    info = eng.KernelInfo()
    avg_kernel_ms = float(np.mean(kernel_ms))
    kernel_jumps_per_s = n * args.nb_run / (avg_kernel_ms * 1e-3)
    achieved = kernel_jumps_per_s * MAD32_PER_JUMP
    # the shader clock the kernel really ran at (s_memtime against the 100 MHz reference, workgroup 0): under this
    # all-VALU load the chip settles well below its 2.4 GHz peak, so the issue-rate peak AT THAT CLOCK is quoted too
    clk = float(np.median([c for c in clock_mhz if c > 0])) if any(c > 0 for c in clock_mhz) else None
    peak_at_clk = 1024 * 64 / MAD_CYCLES_PER_WAVE * clk * 1e6 if clk else None
    # HBM bytes per launch from the PMC passes of the same command (FETCH_SIZE + WRITE_SIZE, see
    # profiles/r01_pmc_hbm.json and DESIGN.md "Measurement"); only quoted when the launch shape is the one profiled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc_hbm.json")) as f:
            pmc = json.load(f)
        if pmc["kangaroos_per_gpu"] == n and pmc["nb_run"] == args.nb_run and pmc["kernel"] == eng.KernelName():
            traffic = pmc["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    # HBM side of the same launch.  The streaming kernel keeps the herd in HBM during the launch: per jump it
    # reads and writes one workspace slot of G = ceil((27 + DW) / 3) 12-byte groups; the
    # register-resident kernels only load/store the herd once per launch.
    name = eng.KernelName()
    if "stream" in name:
        # per jump: x, y read and written as eight words each (64 + 64 B), one stored partial product per pair of
        # kangaroos read and written (16 + 16 B), 1 B of jump-index history written and read back once by
        # kg_hist_sum_kernel; per launch the herd planes (192 B per kangaroo)
        bytes_per_jump = 64 + 64 + 16 + 16 + 1 + 1 + STATE_BYTES_PER_KANGAROO / args.nb_run
    else:
        bytes_per_jump = STATE_BYTES_PER_KANGAROO / args.nb_run
    hbm_gbps = kernel_jumps_per_s * bytes_per_jump / 1e9
    out = {
        "metric": "kangaroo jumps/sec (secp256k1 affine jump incl. grouped inversion, DP test and append)",
        "value": value, "unit": "jumps/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 limbs (9x29-bit), 64-bit accumulate", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: 125-bit range throughput, jump table seed 0x600DCAFE, DP mask 32 bits",
                   "kangaroos_per_gpu": n, "grid": [gx, gy], "nb_run": args.nb_run,
                   "kangaroos_per_lane": info["k_per_lane"], "delta_words": info["delta_words"],
                   "lds_bytes_per_workgroup": info["lds_bytes"], "workgroups": info["grid"], "dps_seen": n_dp,
                   "self_check": {"invariant": "P_i == d_i*G (+ key for wild) after the timed launches",
                                  "sampled_per_gpu": args.check, "ranks_checked": int(dist.sum_over_ranks(1)), "bad": 0}},
        "roofline": {"bound": "valu-int32-mul", "achieved": achieved / 1e12, "peak": PEAK_MAD32_PER_S / 1e12,
                     "unit": "TMAD32/s", "frac": achieved / PEAK_MAD32_PER_S, "traffic": traffic,
                     "traffic_source": "profiles/r03_pmc_hbm.json (rocprofv3 --pmc passes over this command, not measured in this run)"
                     if traffic else None,
                     "kernel": eng.KernelName(), "kernel_ms_avg": avg_kernel_ms,
                     "kernel_ms_median": float(np.median(kernel_ms)),
                     "mad_issue_frac": kernel_jumps_per_s * MADS_ISSUED_PER_JUMP / PEAK_MAD32_PER_S,
                     "mads_issued_per_jump": MADS_ISSUED_PER_JUMP,
                     "shader_clock_mhz": clk if clk else "n/a (register-resident kernel has no clock stamps)",
                     "peak_at_shader_clock": peak_at_clk / 1e12 if peak_at_clk else None,
                     "frac_at_shader_clock": achieved / peak_at_clk if peak_at_clk else None,
                     "peak_r01_short_run": PEAK_MAD32_R01 / 1e12, "frac_vs_peak_r01_short_run": achieved / PEAK_MAD32_R01,
                     "board_power": board_power, "board_power_median_W_per_rank": power_per_rank,
                     "mad32_per_jump": MAD32_PER_JUMP, "kernel_jumps_per_s": kernel_jumps_per_s,
                     "hbm": {"bytes_per_jump": bytes_per_jump, "achieved_GBps": hbm_gbps, "peak_GBps": 8000.0,
                             "frac": hbm_gbps / 8000.0},
                     "note": "bound per SURVEY §8d is VALU integer multiply issue.  peak = best sustained measured v_mad_u64_u32 "
                             "rate (540.8 G wave-instr/s x 64 lanes at a measured 2382 MHz, profiles/r02_ubench_power.txt); frac = "
                             "algorithmic 416 MAD32/jump over that peak; mad_issue_frac = the 626 multiply-adds the 9x29-limb "
                             "kernel really issues per jump over the same peak; frac_at_shader_clock scales the peak to the "
                             "clock the kernel ran at (informative); peak_r01_short_run is round 1's 1 ms measurement.  The "
                             "streaming kernel also moves bytes_per_jump through HBM, see hbm.frac; traffic = PMC HBM bytes "
                             "per launch (profiles/)"},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    eng.close()


if __name__ == "__main__":
    main()

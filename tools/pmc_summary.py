#!/usr/bin/env python3
"""Per-launch averages of every counter collected by tools/profile_bench.sh for the jump kernel, plus the derived
per-jump figures.  Usage: python3 tools/pmc_summary.py gpurun_out/prof_<tag> [kangaroos nb_run] > profiles/<tag>_pmc_summary.txt"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8388608
nb_run = int(sys.argv[3]) if len(sys.argv) > 3 else 256
acc = collections.defaultdict(list)
kernel = None
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kg_walk_kernel" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"].split("(")[0].replace("void kg::", "").replace(" ", "")
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {c: sum(v) / len(v) for c, v in acc.items()}
launches = max(len(v) for v in acc.values())
jumps = n * nb_run
print(f"# rocprofv3 --pmc passes over `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline` (tools/profile_bench.sh), MI355X,")
print(f"# kernel {kernel}, {n} kangaroos x {nb_run} steps per launch; per-launch averages over {launches} launches.")
print("# FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them.")
for c in sorted(avg):
    print(f"{c:28s} {avg[c]:.6g}")
print()
if "SQ_INSTS_VALU" in avg:
    print(f"VALU instructions per jump (per lane)      : {avg['SQ_INSTS_VALU'] * 64 / jumps:.0f}")
    print(f"SALU instructions per jump                 : {avg['SQ_INSTS_SALU'] * 64 / jumps:.0f}")
if "SQ_WAIT_ANY" in avg and "SQ_WAVE_CYCLES" in avg:
    print(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES               : {avg['SQ_WAIT_ANY'] / avg['SQ_WAVE_CYCLES']:.3f}")
if "SQ_LDS_BANK_CONFLICT" in avg:
    print(f"LDS bank-conflict cycles / LDS active      : {avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_LDS_IDX_ACTIVE']:.3f}")
if "FETCH_SIZE" in avg:
    rd, wr = 2 * avg["FETCH_SIZE"] * 1024, avg["WRITE_SIZE"] * 1024
    print(f"HBM read bytes per launch (2 x FETCH_SIZE) : {rd:.4g}   = {rd / jumps:.1f} B/jump (algorithmic 80 B/jump + 96 B/kangaroo)")
    print(f"HBM write bytes per launch (WRITE_SIZE)    : {wr:.4g}   = {wr / jumps:.1f} B/jump (algorithmic 81 B/jump + 96 B/kangaroo)")

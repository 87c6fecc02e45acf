#!/usr/bin/env python3
"""HBM bytes per launch of the jump kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh.

Usage: python3 tools/pmc_hbm_summary.py gpurun_out/prof_<tag> [kangaroos nb_run] > profiles/<round>_pmc_hbm.json
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) counts the 128-byte
requests of a coalesced streaming read at 64 B, so it is doubled; WRITE_SIZE (KB) is taken as it is.
"""
import csv
import glob
import json
import sys

root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8388608
nb_run = int(sys.argv[3]) if len(sys.argv) > 3 else 256
vals = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c = r["Counter_Name"]
        if c in vals and "kg_walk_kernel" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void kg::", "").replace("kg::", "").replace(" ", "")
            vals[c].setdefault(name, []).append(float(r["Counter_Value"]))
kernel = max(vals["FETCH_SIZE"], key=lambda k: len(vals["FETCH_SIZE"][k]))
fetch = sum(vals["FETCH_SIZE"][kernel]) / len(vals["FETCH_SIZE"][kernel])
write = sum(vals["WRITE_SIZE"][kernel]) / len(vals["WRITE_SIZE"][kernel])
per_jump = 64 + 64 + 16 + 16 + 1 + 1 + 192 / nb_run if "stream" in kernel else 192 / nb_run
print(json.dumps({
    "kangaroos_per_gpu": n, "nb_run": nb_run, "kernel": kernel, "launches_averaged": len(vals["FETCH_SIZE"][kernel]),
    "fetch_size_raw_kb": fetch, "write_size_raw_kb": write,
    "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
    "algorithmic_bytes_per_launch": per_jump * n * nb_run,
    "note": "read bytes = 2 x FETCH_SIZE (gfx950 correction for coalesced streaming reads, 16 B per lane), write bytes = "
            "WRITE_SIZE (exact for 16-byte-per-lane streaming stores); kg_hist_sum_kernel's 1 B/jump read is a separate "
            "kernel and not in these counters; algorithmic = what the streaming design moves per jump (x, y as 32 B each "
            "read and written, one 32 B partial product per pair of kangaroos read and written, 1 B of jump history) "
            "plus the herd planes once per launch",
}, indent=1))

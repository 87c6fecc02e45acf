#!/usr/bin/env python3
"""Samples the GPU's power, shader clock and temperature from sysfs (hwmon) while a command runs, to tell a
power- or thermally-limited kernel from an issue-limited one.  Read-only; needs no privileges.
Usage: python tools/power_trace.py out.json -- <command ...>"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

out = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]


def find_hwmon():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        names = os.listdir(d)
        if any(n.startswith("power1") for n in names) or "freq1_input" in names:
            return d
    return None


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


hw = find_hwmon()
files = {}
if hw:
    for key, cands in {"power_uW": ["power1_average", "power1_input"], "sclk_Hz": ["freq1_input"], "mclk_Hz": ["freq2_input"],
                       "temp_mC": ["temp1_input"], "power_cap_uW": ["power1_cap"]}.items():
        for c in cands:
            if os.path.exists(os.path.join(hw, c)):
                files[key] = os.path.join(hw, c)
                break
dev = os.path.dirname(os.path.dirname(hw)) if hw else None
samples = []
stop = threading.Event()


def sampler():
    t0 = time.time()
    while not stop.is_set():
        s = {"t": time.time() - t0}
        for k, p in files.items():
            v = read(p)
            if v is not None and v.lstrip("-").isdigit():
                s[k] = int(v)
        if dev:
            cur = read(os.path.join(dev, "pp_dpm_sclk"))
            if cur:
                for line in cur.splitlines():
                    if line.rstrip().endswith("*"):
                        s["dpm_sclk"] = line.strip()
        samples.append(s)
        time.sleep(0.05)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(0.5)  # idle baseline
t_start = time.time()
rc = subprocess.call(cmd)
t_end = time.time()
time.sleep(0.3)
stop.set()
th.join()
with open(out, "w") as f:
    json.dump({"hwmon": hw, "files": files, "cmd": cmd, "rc": rc, "seconds": t_end - t_start, "samples": samples}, f)


def stat(key, scale):
    v = sorted(s[key] * scale for s in samples if key in s)
    return (v[len(v) // 2], v[-1], v[0]) if v else None


print("hwmon:", hw, "files:", list(files))
for key, scale, unit in (("power_uW", 1e-6, "W"), ("sclk_Hz", 1e-6, "MHz"), ("mclk_Hz", 1e-6, "MHz"), ("temp_mC", 1e-3, "C"),
                         ("power_cap_uW", 1e-6, "W")):
    st = stat(key, scale)
    if st:
        print(f"{key}: median {st[0]:.0f} max {st[1]:.0f} min {st[2]:.0f} {unit}")
sys.exit(rc)

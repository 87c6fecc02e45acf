#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/power_trace.py gpurun_out/r02_power_packed.json -- python -u bench.py --steps 100 --no-cpu-baseline > gpurun_out/r02_power_packed.txt 2>&1
rc=$?; tail -8 gpurun_out/r02_power_packed.txt; [ $rc -eq 0 ] || exit $rc
bash tools/ab.sh r02a packed singles base || exit 1

#!/bin/bash
# Cost of reporting distinguished points: per-kernel times (rocprofv3 kernel trace) of tools/stress_dp.py at several DP
# sizes on one grid.  Usage (inside a gpurun call): bash tools/dp_cost.sh <gx> <gy> <dp_bits> [<dp_bits> ...]
export TMPDIR=/tmp
GX=$1; GY=$2; shift 2
mkdir -p gpurun_out
for dp in "$@"; do
  rm -rf gpurun_out/dpcost_$dp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dpcost_$dp -- python3 tools/stress_dp.py $dp 6 $GX $GY > gpurun_out/dpcost_$dp.txt 2> gpurun_out/dpcost_$dp.err || { echo "dp $dp failed"; tail -5 gpurun_out/dpcost_$dp.err; exit 1; }
  echo "== dp_bits $dp grid ${GX}x${GY}"
  grep -E "kernel ms|DPs drained|insert rate|wall" gpurun_out/dpcost_$dp.txt
  python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/dpcost_$dp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kg_walk" in r["Name"] or "hist_sum" in r["Name"]:
            print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms  min {float(r['MinNs'])/1e6:8.3f}  max {float(r['MaxNs'])/1e6:8.3f}")
PY
done

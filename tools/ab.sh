#!/bin/bash
# A/B of library builds on ONE box: for every abv/libkg_<name>.so (tools/build_variant.sh) given, the smoke parity check and a short bench.
# Usage (inside a gpurun call): bash tools/ab.sh <tag> <name> [<name> ...]      results: gpurun_out/ab_<tag>.txt
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.txt
mkdir -p gpurun_out; : > $OUT
for round in 1 2; do
for v in "$@"; do
  export KANGAROO_AMD_LIB=$PWD/abv/libkg_$v.so
  if [ $round -eq 1 ]; then
    timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1 || { echo "$v: SMOKE FAILED" | tee -a $OUT; exit 1; }
  fi
  timeout -k 10 200 python -u bench.py --steps 30 --no-cpu-baseline > gpurun_out/ab_${TAG}_$v.json 2> gpurun_out/ab_${TAG}_$v.err || { echo "$v: bench failed"; tail -3 gpurun_out/ab_${TAG}_$v.err; exit 1; }
  python - <<PY | tee -a $OUT
import json
d=json.load(open("gpurun_out/ab_${TAG}_$v.json")); r=d["roofline"]
print(f"$v: {d['value']/1e9:.3f} G jumps/s wall, kernel {r['kernel_ms_avg']:.2f} ms avg {r.get('kernel_ms_median',0):.2f} median, clock {r.get('shader_clock_mhz')} MHz, frac {r['frac']:.3f} / at clock {r.get('frac_at_shader_clock')}")
PY
done
done

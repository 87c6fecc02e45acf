#!/bin/bash
# Timing experiments with builds that compute WRONG results (abv/libkg_<name>.so from tools/build_variant.sh with a
# -DKG_TIMING_* switch): bench without self-check, twice per variant.  Usage (inside a gpurun call):
#   bash tools/timing_variants.sh <name> [<name> ...]
for r in 1 2; do for v in "$@"; do
  KANGAROO_AMD_LIB=$PWD/abv/libkg_$v.so timeout -k 10 200 python -u bench.py --steps 30 --no-cpu-baseline --check 0 > gpurun_out/tv_$v.json 2> gpurun_out/tv_$v.err || { echo "$v failed"; tail -3 gpurun_out/tv_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/tv_$v.json'));r=d['roofline'];print('$v: %.3f G jumps/s, kernel %.2f ms, clock %s MHz, %s W' % (d['value']/1e9, r['kernel_ms_avg'], r['shader_clock_mhz'], (r['board_power'] or {}).get('median_W')))"
done; done

#!/bin/bash
# Herd depth scan with clock and power: bash tools/kscan.sh [variant] [extra bench flags]   (inside a gpurun call)
V=${1:-}; shift
[ -n "$V" ] && export KANGAROO_AMD_LIB=$PWD/abv/libkg_$V.so
for g in "64,128" "128,128" "256,128" "512,128"; do
  timeout -k 10 200 python -u bench.py --steps 40 --no-cpu-baseline --grid $g "$@" > gpurun_out/kscan.json 2> gpurun_out/kscan.err || { echo "grid $g failed"; tail -3 gpurun_out/kscan.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/kscan.json'));r=d['roofline'];c=d['config']
print('${V:-lib} grid $g: K=%d  %.3f G jumps/s, kernel %.2f ms, clock %.0f MHz, %s W' % (c['kangaroos_per_lane'], d['value']/1e9, r['kernel_ms_avg'], r['shader_clock_mhz'], (r['board_power'] or {}).get('median_W')))"
done

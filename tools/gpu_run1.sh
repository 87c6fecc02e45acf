#!/bin/bash
# round-2 GPU call: the whole GPU suite + the not-yet-collected no-alloc test + A/B of the packed workspace + short sweeps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_dropin.py -x -v --timeout 300 --timeout-method thread > gpurun_out/r02_gputest1b.txt 2>&1
rc=$?; tail -5 gpurun_out/r02_gputest1b.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r02_noalloc1.txt 2>&1
rc=$?; tail -5 gpurun_out/r02_noalloc1.txt; [ $rc -eq 0 ] || exit $rc
for v in base packed base packed; do
  KANGAROO_AMD_LIB=$PWD/ab/libkg_$v.so timeout -k 10 200 python -u bench.py --steps 20 --no-cpu-baseline > gpurun_out/r02_ab_$v.json 2> gpurun_out/r02_ab_$v.err
  rc=$?; echo "$v rc=$rc $(python -c "import json;d=json.load(open('gpurun_out/r02_ab_$v.json'));print(d['value']/1e9, d['roofline']['kernel_ms_avg'])" 2>&1)"; [ $rc -eq 0 ] || exit $rc
done
timeout -k 10 400 python -u tools/sweep.py --log2n 20,21,22,23,24 --nb-run 256 --seconds 3 --out gpurun_out/r02_sweep_short.json > gpurun_out/r02_sweep_short.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_sweep_short.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u tools/sweep.py --log2n 20,21 --nb-run 64 --k 8,16,64 --seconds 3 --out gpurun_out/r02_sweep_k.json > gpurun_out/r02_sweep_k.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_sweep_k.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u tools/sweep.py --log2n 23 --nb-run 64 --k 8,16,32,64 --seconds 3 --out gpurun_out/r02_sweep_k23.json > gpurun_out/r02_sweep_k23.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_sweep_k23.txt; exit $rc

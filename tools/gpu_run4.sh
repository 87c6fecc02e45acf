#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/r02_gputest2.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_gputest2.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 ./ab/ubench_power 1.2 > gpurun_out/r02_ubench_power.txt 2>&1
rc=$?; cat gpurun_out/r02_ubench_power.txt; exit $rc

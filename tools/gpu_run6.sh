#!/bin/bash
# round 2: full GPU suite on the final kernel, then the committed profile (kernel trace + PMC passes), then the §8d sweep
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02_gputest3.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_gputest3.txt; [ $rc -eq 0 ] || exit $rc
bash tools/profile_bench.sh r02 || exit 1
echo profile-done

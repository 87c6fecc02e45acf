#!/bin/bash
set -o pipefail
bash tools/count_insts.sh packed p2 > gpurun_out/r02_counts1.txt 2>&1; rc=$?; cat gpurun_out/r02_counts1.txt; [ $rc -eq 0 ] || exit $rc
bash tools/ab.sh r02b packed p2

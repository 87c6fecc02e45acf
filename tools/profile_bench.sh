#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel-trace stats, then the PMC passes, each in a run of its own.
# Usage (from the repo root, inside a gpurun call):  bash tools/profile_bench.sh <tag>
# Results land in gpurun_out/prof_<tag>/; copy the summaries worth keeping into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || exit 1
done
echo "profile done: $OUT"

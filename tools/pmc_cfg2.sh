#!/bin/bash
# Memory-path PMC passes over one configuration run (scratch/one_cfg2.py <gx> <gy> <nb_run>).
# Usage: bash tools/pmc_cfg2.sh <tag> <gx> <gy> <nb_run>
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "TA_BUSY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum" "TCC_TAG_STALL_sum TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$OUT/$name" -- python3 scratch/one_cfg2.py "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; continue; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "kg_walk" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(acc):
    v = acc[c]
    print(f"{c:32s} n={len(v)} avg={sum(v)/len(v):.6g}")
PY

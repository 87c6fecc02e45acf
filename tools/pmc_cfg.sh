#!/bin/bash
# PMC passes over one configuration run (scratch/one_cfg2.py <gx> <gy> <nb_run>); KG_* env selects the kernel.
# Usage: bash tools/pmc_cfg.sh <tag> <gx> <gy> <nb_run>
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$OUT/$name" -- python3 scratch/one_cfg2.py "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; exit 1; }
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
    print(f"{c:28s} n={len(v)} avg={sum(v)/len(v):.6g}")
PY

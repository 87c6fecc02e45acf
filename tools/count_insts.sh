#!/bin/bash
# Dynamic instruction counts per jump of the walk kernel for library builds abv/libkg_<name>.so (tools/build_variant.sh) (rocprofv3 PMC pass over
# a 3-launch bench).  Usage (inside a gpurun call): bash tools/count_insts.sh <name> [<name> ...]
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "$@"; do
  export KANGAROO_AMD_LIB=$PWD/abv/libkg_$v.so
  rm -rf gpurun_out/cnt_$v
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d gpurun_out/cnt_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/cnt_$v.json 2> gpurun_out/cnt_$v.err || { echo "$v: rocprofv3 failed"; tail -3 gpurun_out/cnt_$v.err; exit 1; }
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/cnt_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kg_walk_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
jumps = 8388608 * 256
print("$v:", ", ".join(f"{k} {sum(x)/len(x)*64/jumps:.1f}/jump" for k, x in sorted(acc.items()) if k != "SQ_WAVES"), f"({len(acc.get('SQ_INSTS_VALU', []))} launches)")
PY
done

#!/usr/bin/env python3
"""Registers, scratch, LDS and spill counts of every kernel in the built library, read from the gfx950 code object's
metadata notes (no GPU needed).  Usage: python tools/kernel_resources.py [lib.so] [--co out.co] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def extract_code_object(lib, out):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "x")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"])


def resources(co):
    txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    rows = []
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
        rows.append(dict(name=name, vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"),
                         spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                         lds=g("group_segment_fixed_size"), wave=g("wavefront_size")))
    return rows


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(root, "kangaroo_amd", "libkangaroo_amd.so")
    co = sys.argv[sys.argv.index("--co") + 1] if "--co" in sys.argv else "/tmp/kg_gfx950.co"
    extract_code_object(lib, co)
    print(f"# {os.path.relpath(lib, root)}: per kernel  VGPR (arch) / AGPR / SGPR, spilled VGPR / SGPR, scratch bytes per lane,")
    print("# static LDS bytes (the walk kernels add dynamic LDS, kg_kernel_info), wavefront size; waves per SIMD = 512 / (VGPR + AGPR) rounded down to 8")
    for r in resources(co):
        tot = int(r["vgpr"]) + int(r["agpr"]) if r["vgpr"].isdigit() and r["agpr"].isdigit() else 0
        # unified register file: vgpr_count already includes the AGPRs on gfx90a+ (the note reports the total)
        occ = min(8, 512 // max(int(r["vgpr"]), 1)) if r["vgpr"].isdigit() else "?"
        print(f"{r['name']}: vgpr {r['vgpr']} agpr {r['agpr']} sgpr {r['sgpr']} | spilled vgpr {r['spill_v']} sgpr {r['spill_s']} | "
              f"scratch {r['scratch']} B | lds {r['lds']} B | wave{r['wave']} | waves/SIMD by registers {occ}")

#!/usr/bin/env python3
"""Instruction mix of the hot loop of a kernel, from the disassembly of the built code object: finds the innermost
backward branch that spans the most instructions below a size cap (the pass loop of the walk kernels) and counts its
instructions by class.  Usage: python tools/loop_mix.py [lib.so] [kernel-substring]"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, extract_code_object  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 else "kangaroo_amd/libkangaroo_amd.so"
want = sys.argv[2] if len(sys.argv) > 2 else "kg_walk_kernel_stream<2"
co = "/tmp/kg_loop_mix.co"
extract_code_object(lib, co)
dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "-C", co], text=True)
cur, ins = None, []
for line in dis.splitlines():
    m = re.match(r"^([0-9a-f]+) <(.+)>:$", line.strip())
    if m:
        cur = m.group(2)
        continue
    if cur and want in cur and "//" in line:
        text, addr = line.split("//", 1)
        a = int(addr.split(":")[0].strip(), 16)
        parts = text.strip().split(None, 1)
        m = re.search(r"<.*\+0x([0-9a-f]+)>", addr)
        ins.append((a, parts[0], (parts[1] if len(parts) > 1 else "") + (f" <+0x{m.group(1)}>" if m else "")))
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, mn, ops) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        # target: objdump prints the offset operand; compute from encoding is fiddly, use the symbolic comment if present
        m = re.search(r"<.*\+0x([0-9a-f]+)>", ops)
        if m:
            tgt = ins[0][0] + int(m.group(1), 16)
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], i))
if not loops:
    print("no backward branches found")
    sys.exit(1)


def classify(mn):
    if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32"):
        return "valu_mad64"
    if mn.startswith("v_"):
        return "valu_other"
    if mn.startswith("s_nop"):
        return "s_nop"
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("buffer_load", "global_load", "flat_load")):
        return "vmem_load"
    if mn.startswith(("buffer_store", "global_store", "flat_store")):
        return "vmem_store"
    return "other"


print(f"{lib}: {want}: {len(ins)} instructions, {len(loops)} loops")
for lo, hi in sorted(loops, key=lambda t: t[1] - t[0], reverse=True):
    if hi - lo < 300:
        continue
    body = ins[lo:hi + 1]
    c = collections.Counter(classify(mn) for _, mn, _ in body)
    top = collections.Counter(mn for _, mn, _ in body if mn.startswith("v_")).most_common(14)
    print(f"loop {lo}..{hi}: {len(body)} instructions: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))
    print("   top VALU: " + ", ".join(f"{k} {v}" for k, v in top))

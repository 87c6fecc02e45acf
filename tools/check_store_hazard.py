#!/usr/bin/env python3
"""Build-time check of the emitted gfx950 code for the wide-store data hazard (walk_kernel_stream.h):
a VMEM store of more than 8 bytes (dwordx3 / dwordx4) reads its data VGPRs late; a VALU instruction that overwrites
one of them too soon after the store can make the store write the NEW value.  The compiler pads two wait states for
the forms its hazard model knows; round 1 found on MI355X that a buffer store with an SGPR offset needs more, and
the kernels now keep four wait states (ws_hold = s_nop 3) behind those.  This script disassembles every kernel of
the library and fails if a VALU write to a wide store's data register follows the store within
  4 wait states for buffer stores with an SGPR soffset,
  2 wait states for every other wide store (the documented rule),
so that a compiler update cannot silently re-open the hole.  One instruction = one wait state, s_nop N = N + 1.
Branches inside the window are followed: both sides of a conditional one, the target of an unconditional one; a jump
through a register (s_setpc) inside the window counts as a violation, the end of the program ends the scan.
Usage: python tools/check_store_hazard.py [lib.so]   (exit code 1 and a report on violations)"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, extract_code_object  # noqa: E402

WIDE_STORE = re.compile(r"^(buffer|global|flat|scratch)_store_dwordx([34])$")
REG = re.compile(r"^([va])(?:\[(\d+):(\d+)\]|(\d+))$")


def reg_range(op):
    m = REG.match(op.strip())
    if not m:
        return None
    if m.group(4) is not None:
        return m.group(1), int(m.group(4)), int(m.group(4))
    return m.group(1), int(m.group(2)), int(m.group(3))


def parse(disasm):
    """-> {kernel: [(mnemonic, [operands], text, address, branch target address or None)]}"""
    kernels, cur, start = {}, None, 0
    for line in disasm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line.strip())
        if m:
            cur, start = m.group(2), int(m.group(1), 16)
            kernels[cur] = []
            continue
        if cur is None or "//" not in line:
            continue
        text, comment = line.split("//", 1)
        text = text.strip()
        if not text:
            continue
        parts = text.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        addr = int(comment.split(":")[0].strip(), 16)
        t = re.search(r"<.*\+0x([0-9a-f]+)>", comment)
        kernels[cur].append((parts[0], ops, text, addr, start + int(t.group(1), 16) if t else None))
    return kernels


def valu_dest_ranges(mn, ops):
    """VGPR/AGPR ranges a VALU instruction writes (first operand; v_swap writes both)."""
    if not mn.startswith("v_") or not ops:
        return []
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and not mn.startswith("v_cmpx"):
        return []
    out = [reg_range(ops[0])]
    if mn.startswith("v_swap") and len(ops) > 1:
        out.append(reg_range(ops[1]))
    return [r for r in out if r]


def check(kernels):
    violations, n_wide, n_sgpr = [], 0, 0
    for name, ins in kernels.items():
        index_of = {addr: i for i, (_, _, _, addr, _) in enumerate(ins)}

        def scan(j, ws, need, data, store_text, depth=0):
            """follow the code from instruction j with ws wait states already behind the store"""
            while j < len(ins) and ws < need:
                mn2, ops2, text2, _, target = ins[j]
                if mn2 == "s_endpgm":
                    return
                if mn2.startswith("s_setpc") or mn2.startswith("s_swappc"):
                    violations.append((name, store_text, text2 + "  (indirect jump inside the window)", ws, need))
                    return
                if mn2 == "s_branch" or mn2.startswith("s_cbranch"):
                    if target is None or target not in index_of or depth > 4:
                        violations.append((name, store_text, text2 + "  (branch target not found)", ws, need))
                        return
                    scan(index_of[target], ws + 1, need, data, store_text, depth + 1)
                    if mn2 == "s_branch":
                        return
                    ws += 1
                    j += 1
                    continue
                for kind, lo, hi in valu_dest_ranges(mn2, ops2):
                    if kind == data[0] and lo <= data[2] and hi >= data[1]:
                        violations.append((name, store_text, text2, ws, need))
                ws += int(ops2[0], 0) + 1 if mn2 == "s_nop" and ops2 else 1
                j += 1

        for i, (mn, ops, text, _, _) in enumerate(ins):
            m = WIDE_STORE.match(mn)
            if not m:
                continue
            n_wide += 1
            if m.group(1) == "buffer":
                data = reg_range(ops[0])
                sgpr_off = len(ops) > 3 and re.match(r"^s\d+", ops[3].split()[0]) is not None
            else:  # global / flat / scratch: vaddr, vdata, ...
                data = reg_range(ops[1]) if len(ops) > 1 else None
                sgpr_off = False
            if data is None:
                continue
            n_sgpr += sgpr_off
            scan(i + 1, 0, 4 if sgpr_off else 2, data, text)
    return violations, n_wide, n_sgpr


def main(lib=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = lib or os.path.join(root, "kangaroo_amd", "libkangaroo_amd.so")
    with tempfile.TemporaryDirectory() as tmp:  # builds and A/B variants may run this side by side
        co = os.path.join(tmp, "kernels.co")
        extract_code_object(lib, co)
        dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], text=True)
    kernels = parse(dis)
    violations, n_wide, n_sgpr = check(kernels)
    print(f"{os.path.relpath(lib, root)}: {len(kernels)} kernels, {n_wide} wide stores ({n_sgpr} with an SGPR offset), "
          f"{len(violations)} hazard violations")
    for name, st, wr, ws, need in violations:
        print(f"  {name}: `{wr}` overwrites the data of `{st}` after {ws} wait state(s), {need} required")
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))

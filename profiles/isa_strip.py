#!/usr/bin/env python3
"""One character per instruction of a kernel in hipcc's -save-temps assembly, in program order: the interleaving of the matrix
pipe with the vector ALU at a glance.  M matrix instruction, v vector ALU, a accvgpr move, D LDS, G global memory, s scalar,
w s_waitcnt, n s_nop, B barrier.
    python profiles/isa_strip.py FILE.s KERNEL_SUBSTRING [first [count]]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
out, on = [], False
for line in open(path):
    if not on:
        on = bool(re.match(r"^_Z\S*%s\S*:" % re.escape(key), line))
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    out.append("M" if op.startswith("v_mfma") else "a" if op.startswith("v_accvgpr") else "v" if op.startswith("v_") else
               "D" if op.startswith("ds_") else "G" if op.startswith(("global_", "buffer_", "flat_")) else
               "B" if op.startswith("s_barrier") else "w" if op.startswith("s_waitcnt") else "n" if op.startswith("s_nop") else
               "s" if op.startswith("s_") else "?")
txt = "".join(out[first:first + count])
for i in range(0, len(txt), 160):
    print("%6d %s" % (first + i, txt[i:i + 160]))

#!/usr/bin/env python3
"""Static instruction count of a kernel by class, from the assembly hipcc emits (-save-temps) or from the built library.

    python profiles/isa_count.py <file.s | libpinn_hip.so> <kernel-name substring, e.g. k_fused20dILi0ELi8ELb1> [...]

Classes: mfma | valu_f64 | dpp | bpermute | lds | accvgpr | waitcnt | nop | scalar | valu_other | vmem.  With a lone wave per
SIMD every instruction costs an issue slot of ~5.3 cycles, so the count IS the cost model of the one-tile
launches.  Also prints the register / scratch figures of the kernel descriptor (.s input only)."""
import collections
import os
import re
import sys


def classify(m):
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("v_accvgpr"): return "accvgpr"
    if m.startswith("ds_bpermute") or m.startswith("ds_swizzle"): return "bpermute"
    if m.startswith("ds_"): return "lds"
    if m == "s_waitcnt": return "waitcnt"
    if m == "s_nop": return "nop"
    if m == "s_barrier": return "barrier"
    if m.startswith("s_"): return "scalar"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if "dpp" in m: return "dpp"
    if re.search(r"_f64", m): return "valu_f64"
    if m.startswith("v_"): return "valu_other"
    return "other"


def from_s(path):
    funcs, cur, meta = {}, None, {}
    for line in open(path):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            cur = m.group(1); funcs[cur] = []; continue
        if t.startswith(".Lfunc_end"):
            cur = None; continue
        if t.startswith(".amdhsa_kernel "):
            meta[t.split()[1]] = {}
            continue
        m = re.match(r"^\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size|group_segment_fixed_size|next_free_sgpr)\s+(\S+)", t)
        if m and meta:
            meta[list(meta)[-1]][m.group(1)] = m.group(2)
        if cur is None or not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        mn = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", mn):
            funcs[cur].append(mn + ("_dpp" if (" row_" in t or "quad_perm" in t) and "dpp" not in mn else ""))
    return funcs, meta


def from_so(path):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "helpers"))
    import isa_lint
    funcs = {}
    for _, obj in isa_lint.code_objects(path):
        for name, ins in isa_lint.disassemble(obj).items():
            funcs[name] = [i[1] for i in ins]
    return funcs, {}


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    funcs, meta = from_s(path) if path.endswith(".s") else from_so(path)
    for name, ins in funcs.items():
        if pats and not any(p in name for p in pats):
            continue
        c = collections.Counter(classify(m) for m in ins)
        print("%s\n  total %d  %s" % (name, len(ins), "  ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))
        if name in meta:
            print("  descriptor:", meta[name])


if __name__ == "__main__":
    main()

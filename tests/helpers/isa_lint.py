"""Hazard lint over the gfx950 code objects inside libpinn_hip.so (test infrastructure, CPU only).

hipcc's hazard recogniser does not look inside inline asm.  csrc/ has two kinds of inline asm whose operands take part in
a hazard the recogniser would otherwise have padded with s_nop (csrc/kernels_fused20d.h `agd_put*`,
csrc/kernels_fused20m.h `agpr_put`, `lds_dma_b128`):

  R1  a matrix instruction writes VGPRs / AGPRs, a following `v_accvgpr_write_b32 aX, vS` / `v_accvgpr_read_b32 vD, aS`
      reads one of them before the result has landed ("XDL / DGEMM write VGPR -> VALU read", LLVM
      GCNHazardRecognizer::checkMAIVALUHazards; observed in round 2 as stale low words, 1e-8 relative errors)
  R2  `s_mov_b32 m0, ...` directly followed by an LDS-DMA load that reads m0 (GCNHazardRecognizer::checkReadM0Hazards:
      1 wait state on gfx9)

The sources keep clear of both by construction (`agd_put_after`'s dependency operand; an s_nop inside the DMA asm); this
lint checks the RESULT: it extracts every gfx950 code object from the library's .hip_fatbin section, disassembles it with
llvm-objdump and walks each kernel's control-flow graph forward from every matrix instruction / m0 write, counting wait
states the way the recogniser does (one per instruction issued in between, N+1 for `s_nop N`).  A violation names kernel,
offset and the two instructions.  `python tests/helpers/isa_lint.py [lib]` prints the report."""
import os
import re
import struct
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
OBJCOPY = "objcopy"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# Wait states a VALU read of a matrix-instruction result needs.  LLVM's tables (DGEMM 4x4: 6; SMFMA: passes + 2, i.e.
# 4 for f32 4x4x1, 10 for f32 16x16x4; f64 16x16x4 on gfx950: 19) and, measured, the SMALLEST distance hipcc itself leaves
# between such an instruction and a compiler-visible VALU read of its result anywhere in this library (it pads with s_nop
# up to exactly these): 6, 4, 10, 19.  Kinds not used today take the 16-pass bound.
MFMA_WAIT = [
    (re.compile(r"v_mfma_f64_4x4x4"), 6),
    (re.compile(r"v_mfma_f64_16x16x4"), 19),
    (re.compile(r"v_mfma_f32_4x4x1"), 4),
    (re.compile(r"v_mfma_f32_16x16x4"), 10),
    (re.compile(r"v_mfma_"), 19),
]
M0_WAIT = 1

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def code_objects(lib):
    """[(index, bytes)] of the gfx950 device code objects bundled into `lib`."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fatbin")
        subprocess.run([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", blob, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append((len(out), blob[i + off:i + off + size]))
        pos = i + len(MAGIC)
    return out


def disassemble(obj_bytes):
    """{kernel: [(offset, mnemonic, operand string, branch target offset or None)]}"""
    with tempfile.NamedTemporaryFile(suffix=".o") as fh:
        fh.write(obj_bytes)
        fh.flush()
        text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", fh.name], check=True, capture_output=True,
                              text=True).stdout
    funcs, cur, start = {}, None, 0
    head = re.compile(r"^([0-9a-f]+) <([^>]+)>:$")
    ins = re.compile(r"^\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$")
    tgt = re.compile(r"<[^>]*?\+0x([0-9a-fA-F]+)>\s*$")
    for line in text.splitlines():
        m = head.match(line)
        if m:
            cur, start = funcs.setdefault(m.group(2), []), int(m.group(1), 16)
            continue
        m = ins.match(line)
        if m and cur is not None:
            t = tgt.search(m.group(4)) if m.group(1).startswith(("s_branch", "s_cbranch")) else None
            if t is None and m.group(1).startswith(("s_branch", "s_cbranch")) and re.search(r"<[^>+]*>\s*$", m.group(4)):
                target = start                  # a branch to the first instruction prints without +0x
            else:
                target = start + int(t.group(1), 16) if t else None
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), target))
    return funcs


def regs(operand):
    """set of ('v'|'a', index) named by one operand string"""
    s = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return s


def split_operands(ops):
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def mfma_wait(mn):
    for rx, w in MFMA_WAIT:
        if rx.match(mn):
            return w
    return None


_WRITES_FIRST = re.compile(r"^(v_(?!cmp_)|ds_read|ds_bpermute|ds_permute|global_load|buffer_load|scratch_load|flat_load)")
_IS_DMA = re.compile(r"_lds_|^(global|buffer)_load.*\blds\b")


def lint_function(name, body):
    """Violations of R1 / R2 in one kernel: from every matrix instruction (every write of m0) walk the control-flow
    graph forward for as many wait states as its result needs and look for an accvgpr move (LDS-DMA) that reads it;
    a path ends where the registers are overwritten, at s_endpgm, or when the wait states are used up."""
    index = {off: i for i, (off, _, _, _) in enumerate(body)}

    def successors(i):
        _, mn, _, target = body[i]
        if mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            return []
        nxt = [i + 1] if i + 1 < len(body) else []
        if mn == "s_branch":
            return [index[target]] if target in index else []
        if mn.startswith("s_cbranch") and target in index:
            return nxt + [index[target]]
        return nxt

    def states(i):
        _, mn, ops, _ = body[i]
        return int(ops.split()[0], 0) + 1 if mn == "s_nop" else 1

    bad = []
    for i, (off, mn, ops, _) in enumerate(body):
        need = mfma_wait(mn)
        if need is not None:
            text = mn + " " + ops
            seen = set()
            stack = [(j, need, frozenset(regs(split_operands(ops)[0]))) for j in successors(i)]
            while stack:
                j, owed, live = stack.pop()
                if owed <= 0 or not live or (j, owed, live) in seen:
                    continue
                seen.add((j, owed, live))
                off2, mn2, ops2, _ = body[j]
                o = split_operands(ops2)
                if mn2 in ("v_accvgpr_write_b32", "v_accvgpr_read_b32") and len(o) > 1 and regs(o[1]) & live:
                    bad.append("%s +0x%x: `%s %s` reads the result of `%s` (+0x%x) %d wait state(s) early"
                               % (name, off2, mn2, ops2, text, off, owed))
                    continue
                if _WRITES_FIRST.match(mn2) and o:
                    live = live - regs(o[0])
                for k in successors(j):
                    stack.append((k, owed - states(j), live))
        if mn.startswith("s_") and re.match(r"m0\b", ops):
            for j in successors(i):
                off2, mn2, ops2, _ = body[j]
                if _IS_DMA.search(mn2 + " " + ops2):
                    bad.append("%s +0x%x: `%s %s` reads m0 written at +0x%x with no wait state in between"
                               % (name, off2, mn2, ops2, off))
    return bad


def lint_library(lib):
    """(number of kernels walked, counts of the instructions of interest, violations)"""
    n, seen, bad = 0, {"v_accvgpr_write_b32": 0, "v_accvgpr_read_b32": 0, "lds_dma": 0, "mfma": 0}, []
    for _, blob in code_objects(lib):
        for name, body in disassemble(blob).items():
            n += 1
            for _, mn, ops, _ in body:
                if mn in seen:
                    seen[mn] += 1
                elif "_lds_" in mn:
                    seen["lds_dma"] += 1
                elif mn.startswith("v_mfma_"):
                    seen["mfma"] += 1
            bad += lint_function(name, body)
    return n, seen, bad


if __name__ == "__main__":
    import sys
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "pinns-tf2.0_amd", "pinn_native",
        "libpinn_hip.so")
    n, seen, bad = lint_library(lib)
    print("%d kernels, %s" % (n, seen))
    for b in bad:
        print("VIOLATION", b)
    sys.exit(1 if bad else 0)

"""The built library's device code holds none of the two hazards inline asm can hide from hipcc's hazard recogniser
(tests/helpers/isa_lint.py: matrix-instruction result read by an accvgpr move too early; m0 written right in front of an
LDS-DMA load).  CPU only: reads the gfx950 code objects out of libpinn_hip.so with objcopy + llvm-objdump."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import isa_lint  # noqa: E402

needs_tools = pytest.mark.skipif(not (os.path.exists(isa_lint.OBJDUMP) and shutil.which(isa_lint.OBJCOPY)),
                                 reason="llvm-objdump / objcopy not here")


def _body(lines):
    return [(4 * i, mn, ops, None) for i, (mn, ops) in enumerate(lines)]


def test_the_lint_sees_the_hazards_it_is_there_for():
    put = ("v_accvgpr_write_b32", "a5, v10")
    dgemm = ("v_mfma_f64_4x4x4_4b_f64", "v[10:11], v[2:3], v[4:5], v[10:11]")
    filler = ("v_add_f64", "v[20:21], v[22:23], v[24:25]")
    # the round-2 bug: the parked word is read straight out of the matrix instruction
    assert len(isa_lint.lint_function("k", _body([dgemm, put]))) == 1
    assert len(isa_lint.lint_function("k", _body([dgemm] + [filler] * 5 + [put]))) == 1
    assert isa_lint.lint_function("k", _body([dgemm] + [filler] * 6 + [put])) == []
    assert isa_lint.lint_function("k", _body([dgemm, ("s_nop", "5"), put])) == []
    assert len(isa_lint.lint_function("k", _body([dgemm, ("s_nop", "4"), put]))) == 1
    # a compiler-visible instruction that overwrites the register ends the hazard; an unrelated register never had one
    assert isa_lint.lint_function("k", _body([dgemm, ("v_mov_b32_e32", "v10, v3"), put])) == []
    assert isa_lint.lint_function("k", _body([dgemm, ("v_accvgpr_write_b32", "a5, v12")])) == []
    # ... and it follows branches: the read sits at the branch target
    body = [(0, "v_mfma_f32_16x16x4_f32", "a[0:3], v1, v2, a[0:3]", None), (8, "s_branch", "2", 20),
            (12, "v_accvgpr_write_b32", "a0, 0", None), (20, "v_accvgpr_read_b32", "v9, a0", None)]
    assert len(isa_lint.lint_function("k", body)) == 1
    body[1] = (8, "s_branch", "0", 12)
    assert isa_lint.lint_function("k", body[:3]) == []
    # m0
    dma = ("global_load_lds_dwordx4", "v[44:45], off")
    assert len(isa_lint.lint_function("k", _body([("s_mov_b32", "m0, s5"), dma]))) == 1
    assert isa_lint.lint_function("k", _body([("s_mov_b32", "m0, s5"), ("s_nop", "0"), dma])) == []


@needs_tools
def test_no_hidden_hazard_in_the_built_library():
    import pinn_native
    pinn_native.load()                      # builds it if it is not there (hipcc cross-compiles without a GPU)
    n, seen, bad = isa_lint.lint_library(pinn_native.LIB_PATH)
    # the walk saw what it is meant to judge: the AGPR stash of the width-20 kernels and the LDS-DMA of the float32 ones
    assert n > 50 and seen["mfma"] > 10000 and seen["v_accvgpr_write_b32"] > 1000 and seen["lds_dma"] > 20, (n, seen)
    assert bad == [], "\n".join(bad[:20])


@needs_tools
def test_no_scratch_in_the_kernels_that_serve_the_baseline_configs_or_the_exchange():
    """Code-object metadata of the built library (llvm-readelf --notes): the loss+gradient kernels of BASELINE configs[1..4],
    the reduction / optimiser kernels behind them and -- since round 6 -- every variant of k_reduce_xgmi (the fallback exchange
    when RCCL fails everywhere; it carried 18 spilled registers until round 5) use no private segment and spill nothing."""
    import re
    import subprocess
    import tempfile
    import pinn_native
    readelf = os.path.join(os.path.dirname(isa_lint.OBJDUMP), "llvm-readelf")
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not here")
    pinn_native.build()
    wanted = ("k_reduce_xgmi", "k_fused20dILi0ELi8", "k_fused20dILi1ELi8", "k_fused20mILi0ELi8", "k_fused20mILi1ELi8",
              "k_t16_fusedILi2ELi4", "k_reduce_adam", "k_reduce_rows", "k_lbc_dots", "k_lbc_coef_apply")
    seen = {}
    for _, obj in isa_lint.code_objects(pinn_native.LIB_PATH):
        with tempfile.NamedTemporaryFile(suffix=".o") as fh:
            fh.write(obj)
            fh.flush()
            notes = subprocess.run([readelf, "--notes", fh.name], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if any(w in name for w in wanted):
                seen[name] = (int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)),
                              int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)))
    assert sum("k_reduce_xgmi" in n for n in seen) == 4 and any("k_t16_fusedILi2ELi4" in n for n in seen), sorted(seen)
    bad = {n: v for n, v in seen.items() if v != (0, 0)}
    assert not bad, bad

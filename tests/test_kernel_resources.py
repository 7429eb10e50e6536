"""Build-time guard of the hand-scheduled kernels (tools/check_resources.py): the device code is compiled with
-Rpass-analysis=kernel-resource-usage and every product kernel must stay inside its register budget -- K1's two primary-only
forms and the megakernel without its bounce loop at <= 64 VGPRs / <= 80 SGPRs / no scratch (eight waves per SIMD: past 80
scalar registers a SIMD holds seven, which the compiler's occupancy remark does not show).  Needs hipcc, no GPU."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_product_kernels_stay_inside_their_register_budgets():
    import check_resources
    kernels = check_resources.report()
    bad, rows = check_resources.check(kernels)
    assert not bad, "\n".join(bad)
    assert len(rows) >= len(check_resources.BUDGET)
    # the checker itself: one more scalar register in the bench line's kernel is reported
    frag = "k_primaryILi7ELb0ELi1ELb1ELi0EE"
    name = next(n for n in kernels if frag in n)
    worse = {n: dict(k) for n, k in kernels.items()}
    worse[name]["TotalSGPRs"] = 81
    bad2, _ = check_resources.check(worse)
    assert any(frag in b and "SGPRs" in b for b in bad2)


def test_assembly_blocks_write_only_what_they_declare():
    """tools/check_asm_clobbers.py on the preprocessed device source: literal registers of every hand-written block are in its
    clobber list, written operands are output operands -- and the lint does find a block that breaks either rule"""
    import check_asm_clobbers as lint
    stmts = lint.asm_statements(lint.preprocessed())
    bad, blocks = lint.lint(stmts)
    assert blocks >= 10, blocks                      # the look-up loops and their counting twins, the DPP reductions, the runs
    assert not bad, bad[:5]
    good = ('asm volatile("s_mov_b64 s[68:69], exec\\n\\t" "v_mov_b32 v53, %[a]\\n\\t" "v_add_u32 %[o], v53, %[a]\\n\\t" "s_nop 0\\n\\t" "s_nop 0\\n\\t" "s_nop 0\\n\\t"'
            ' "s_nop 0\\n\\t" "s_nop 0\\n\\t" "s_mov_b64 exec, s[68:69]\\n\\t" : [o] "=v"(o) : [a] "v"(a) : "v53", "s68", "s69");')
    assert lint.lint(lint.asm_statements(good)) == ([], 1)
    undeclared = good.replace('"v53", "s68", "s69"', '"s68", "s69"')
    assert any("v53" in b for b in lint.lint(lint.asm_statements(undeclared))[0])
    writes_input = good.replace("v_add_u32 %[o], v53, %[a]", "v_add_u32 %[a], v53, %[a]")
    assert any("INPUT operand" in b for b in lint.lint(lint.asm_statements(writes_input))[0])

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

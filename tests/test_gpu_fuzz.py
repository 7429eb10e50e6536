"""Randomised parity sweep (tests/fuzz_parity.py): random scenes (cubic and non-cubic, empty to dense), cameras (outside,
inside, on lattice points, missing the volume), every traversal mode, split / megakernel, budgets from 0 to 2000 steps,
jitter, random light, denoiser passes / step widths / UBO modes -- all planes bit-exact against the oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_parity(engine):
    import fuzz_parity
    assert fuzz_parity.run(300, 20261003, engine, verbose=False) == 0

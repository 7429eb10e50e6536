import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def vrt():
    import voxel_raytracing_amd as v
    return v


@pytest.fixture(scope="session")
def engine(vrt):
    """One context on cuda:0 for the whole GPU session (fails loudly if the HIP library is missing)."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    e = vrt.Engine(0)
    yield e
    e.destroy()

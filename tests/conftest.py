import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ops_golden():
    with open(os.path.join(GOLDEN, "ops.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ops_small():
    return dict(np.load(os.path.join(GOLDEN, "ops_small.npz")))


@pytest.fixture(scope="session")
def modules_golden():
    return dict(np.load(os.path.join(GOLDEN, "modules.npz")))


@pytest.fixture(scope="session")
def fakequant_golden():
    return dict(np.load(os.path.join(GOLDEN, "fakequant.npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def C():
    """The HIP operator module (C-ABI through ctypes).  GPU tests only."""
    import torch
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    import mixdq_amd._C as C_
    return C_

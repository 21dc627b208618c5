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


# Collection order of the GPU suite: the bit-exact operator parity tests first (seconds, the heart
# of the path: a1-a4), then the module layer, the fused producers, attention, and the UNet-level
# tolerance tests last -- under the driver's `-x` a failure late in the list cannot hide the
# operator evidence.  Within a file the definition order is kept.
_FILE_ORDER = ["test_ops_gpu.py", "test_f16in_gpu.py", "test_large_gpu.py", "test_modules_gpu.py", "test_fused_gpu.py",
               "test_f16_gpu.py", "test_attention_gpu.py", "test_unet_gpu.py", "test_unet_full_gpu.py",
               "test_unet_path_a_gpu.py", "test_dist_gpu.py"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER)
    items.sort(key=rank)          # stable: keeps the order inside each file


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

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds on the CPU")


@pytest.fixture(scope="session")
def built():
    """Product library + checker + test-only simulator, all built in-tree (hipcc cross-compiles without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def anchors():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "anchors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def c1_data():
    import numpy as np
    return np.fromfile(os.path.join(ROOT, "tests", "golden", "testfloat_8_8_128.dat"), dtype=np.float32).reshape(128, 8, 8)


_torch_gpu_ready = []


def pytest_runtest_setup(item):
    """The first GPU test brings up torch's HIP context before the product library has touched the device: some GPU tests hand torch
    tensors to the C ABI, and torch initialising AFTER hundreds of library calls in the same process was seen to fail once
    ("No HIP GPUs are available") when the files ran in another order."""
    if item.get_closest_marker("gpu") is not None and not _torch_gpu_ready:
        _torch_gpu_ready.append(True)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()

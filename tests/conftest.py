import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import ast

    import numpy as np
    import torch

    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = ast.literal_eval(str(z["meta"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


@pytest.fixture(scope="session")
def golden_swiglu():
    return load_golden("ref_tiny_swiglu")


@pytest.fixture(scope="session")
def golden_gelu():
    return load_golden("ref_tiny_gelu")


@pytest.fixture(scope="session")
def golden_radius():
    return load_golden("ref_tiny_radius")


@pytest.fixture(scope="session")
def golden_forward():
    return load_golden("ref_tiny_forward_eval")

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
    arrays = {k: (z[k] if z[k].dtype.kind in "US" else torch.from_numpy(z[k])) for k in z.files if k != "meta"}
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
def golden_central():
    return load_golden("ref_tiny_central")


@pytest.fixture(scope="session")
def golden_forward():
    return load_golden("ref_tiny_forward_eval")


@pytest.fixture(scope="session")
def golden_voronoi():
    return load_golden("ref_tiny_voronoi")


@pytest.fixture(scope="session")
def golden_hier():
    return load_golden("ref_tiny_hier")


@pytest.fixture(scope="session")
def golden_ply():
    return load_golden("ref_demo_ply")


def ply_cases(golden):
    """(name, xyz [1,N,3] f32, rgb [1,N,3] f32, arrays-of-that-file) for every demo PLY of the fixture (aliases resolved)."""
    import numpy as np
    import torch

    meta, a = golden
    for fname in meta["files"]:
        key = fname.replace(".ply", "")
        src = key
        if f"{key}__same_as" in a:
            src = str(a[f"{key}__same_as"])
        sub = {k.split("__", 1)[1]: v for k, v in a.items() if k.startswith(src + "__")}
        xyz = sub["xyz"][None]
        rgb = torch.from_numpy(sub["rgb_u8"].numpy().astype(np.float64) / 255).float()[None]   # demo/app.py:117
        yield key, xyz, rgb, sub

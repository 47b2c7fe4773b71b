import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pkg(sub=None):
    """import 3pu_pytorch_amd[.sub] (the directory name is not a Python identifier)."""
    return importlib.import_module("3pu_pytorch_amd" + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device in this container")
    # the HIP extension must be the thing that runs: fail loudly if it is missing
    pkg("_lib").lib()
    return torch.device("cuda", 0)


def sphere(seed, n, b=1):
    """(b, n, 3) f32 points uniform on S^2 (normalised Gaussians), seeded."""
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((b, n, 3)).astype(np.float32)
    p /= np.linalg.norm(p, axis=2, keepdims=True).astype(np.float32)
    return p.astype(np.float32)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>.npz as a dict of torch tensors / python scalars."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if a.dtype.kind in "US":
                out[k] = [str(v) for v in a.tolist()]
                continue
            out[k] = a.item() if a.ndim == 0 and a.dtype.kind in "iuf" and k not in _TENSOR_SCALARS else torch.from_numpy(np.array(a))
    return out


_TENSOR_SCALARS = set()


def rel_err(a, b):
    a, b = a.double(), b.double()
    denom = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / denom)


@pytest.fixture(scope="session")
def golden():
    return load_golden

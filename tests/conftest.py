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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    """npz -> dict of torch tensors (0-d arrays become python scalars)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        for k in z.files:
            a = z[k]
            if a.dtype.kind in "US":
                out[k] = a
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


def state_dict_from(g, prefix="sd."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def sd_v7():
    return load_golden("sd_v7")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with `-m gpu`; without a device they are skipped rather than failed.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

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
    config.addinivalue_line("markers", "slow: full-size CPU oracle cases (tens of seconds)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" and z[k].ndim > 0 else z[k].item()
                    if z[k].ndim == 0 else z[k]) for k in z.files}


def rel_err(a, b):
    a = a.detach().double().flatten()
    b = b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ellipse_mask(B, H, W):
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    m = ((xx / 0.62) ** 2 + (yy / 0.72) ** 2 <= 1.0).float()
    return m[None, None].repeat(B, 1, 1, 1)


def border_mask(B, H, W, border):
    m = torch.zeros(B, 1, H, W)
    m[:, :, border:H - border, border:W - border] = 1
    return m


def subsample_act(key, ten):
    """same slices as tests/golden/make_golden.py::subsample_act"""
    if key == "outfeat":
        return ten[:, ::8, ::4, ::4] if ten.shape[-1] > 8 else ten[:, ::8]
    n = ten.shape[2]
    return ten[:, ::4, ::max(1, n // 32)]


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()

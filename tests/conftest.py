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
    # the CPU oracle runs on torch's intra-op pool: keep it at this process's CPU share (a GPU box shows 256 logical CPUs
    # and allows 16; at torch's default of 128 threads the full-size oracle cases ran 2.7x slower, hostinfo.py)
    from adaprompt_amd import hostinfo
    hostinfo.limit_torch_threads()
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


# ---- Prodigy fixtures (tests/golden/make_golden.py: run_prodigy) -- shapes, configs and the seeded gradients by name
PRODIGY_SHAPES = [(37, 19), (129,), (5, 3, 3, 3), (1,)]
PRODIGY_CASES = {
    "zs": dict(betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, safeguard_warmup=False, weight_decay=0.0),
    "fast_wd": dict(betas=(0.985, 0.993), d_coef=5.0, use_bias_correction=True, safeguard_warmup=True,
                    weight_decay=0.01),
    "plain_growth": dict(betas=(0.9, 0.999), d_coef=1.0, use_bias_correction=False, safeguard_warmup=False,
                         weight_decay=0.0, growth_rate=1.5),
    "coupled_wd": dict(betas=(0.9, 0.99), d_coef=1.0, use_bias_correction=True, safeguard_warmup=False,
                       weight_decay=0.02, decouple=False),
}


def prodigy_params(case):
    from adaprompt_amd import synth
    return [synth.synthetic_input(f"prodigy.{case}.p{i}", sh, 0, 0.3).clone() for i, sh in enumerate(PRODIGY_SHAPES)]


def prodigy_grads(case, step):
    from adaprompt_amd import synth
    if step == 5:
        return [torch.zeros(sh) for sh in PRODIGY_SHAPES]
    return [synth.synthetic_input(f"prodigy.{case}.g{i}.s{step}", sh, 0, 1.0) * (0.01 if step != 3 else 3.0)
            for i, sh in enumerate(PRODIGY_SHAPES)]

"""The DDIM loop against the reference's OWN DDIMSampler (tests/golden/make_golden_ddim.py: ddim.py:12-292 run on CPU with its
``register_buffer`` replaced on the instance and a closed-form eps model): the oracle's restatement AND the product's sampler
(adaprompt_amd/ldm/models/diffusion/ddim.py, here driven by the same closed-form model on CPU tensors) are held to the same
latents, per-step x0 predictions, timesteps, sigmas and the sequence of UNet calls (doubled batch under guidance)."""
import types

import numpy as np
import pytest
import torch

from oracle import ddim_oracle as DO
from tests.conftest import load_golden as _load


def load_golden(name):
    """numpy view of the fixture (conftest hands out torch tensors)"""
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in _load(name).items()}

CASES = {"guided4": (4, 2, [10.0, 4.0], 0.0, "uniform", 11, True), "guided5_eta": (5, 3, [7.5, 2.0], 0.6, "uniform", 12, True),
         "plain8": (8, 2, [1.0, 1.0], 0.0, "uniform", 13, False), "quad6": (6, 1, [3.0, 2.0], 0.0, "quad", 14, True)}


def eps_closed_form(x, t, c):
    ctx = c[0] if isinstance(c, tuple) else c
    cm = ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1)
    return torch.tanh(0.8 * x + 0.1 * cm) * (0.5 + t.view(-1, 1, 1, 1).float() / 2000.0) + 0.05 * torch.sin(3.0 * x)


def _inputs(g, name, as_tuple):
    xT, ctx, uctx = (torch.from_numpy(g[f"{name}.{k}"]) for k in ("x_T", "ctx", "uctx"))
    B = xT.shape[0]
    c = (ctx, ["a"] * B, {"k": 1}) if as_tuple else ctx
    uc = (uctx, [""] * B, {"k": 1}) if as_tuple else uctx
    return xT, c, uc


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_ddim_loop_vs_reference_sampler(name):
    g = load_golden("ddim_loop")
    S, B, scale, eta, disc, seed, as_tuple = CASES[name]
    xT, c, uc = _inputs(g, name, as_tuple)
    ac = torch.tensor(g["alphas_cumprod"], dtype=torch.float32)
    noises = [torch.from_numpy(n) for n in g[f"{name}.noises"]] if eta > 0 else None
    calls = []

    def eps_fn(x, t, cond):
        calls.append((x.shape[0], int(t[0])))
        return eps_closed_form(x, t, cond)
    z, preds = DO.ddim_sampling(eps_fn, ac, c, xT, S, tuple(scale), uc, eta=eta, ddim_discretize=disc, noises=noises)
    assert [b for b, _ in calls] == g[f"{name}.call_batch"].tolist()
    assert [t for _, t in calls] == g[f"{name}.call_t"].tolist()
    np.testing.assert_allclose(z.numpy(), g[f"{name}.z"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(torch.stack(preds).numpy(), g[f"{name}.pred_x0"], rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(DO.make_ddim_timesteps(disc, S, 1000), g[f"{name}.ddim_timesteps"])


@pytest.mark.parametrize("name", list(CASES))
def test_product_ddim_sampler_vs_reference_sampler(name):
    from adaprompt_amd.ldm.models.diffusion.ddim import DDIMSampler
    g = load_golden("ddim_loop")
    S, B, scale, eta, disc, seed, as_tuple = CASES[name]
    xT, c, uc = _inputs(g, name, as_tuple)
    ac = torch.tensor(g["alphas_cumprod"], dtype=torch.float32)
    calls = []

    def apply_model(x, t, cond):
        calls.append((x.shape[0], int(t[0])))
        return eps_closed_form(x, t, cond)
    model = types.SimpleNamespace(num_timesteps=1000, betas=torch.zeros(1000), alphas_cumprod=ac, apply_model=apply_model)
    sampler = DDIMSampler(model)
    torch.manual_seed(seed + 100)                 # eta > 0: the same global draws as the reference run (util.py noise_like)
    if disc == "uniform":
        z, inter = sampler.sample(S=S, batch_size=B, shape=[4, 8, 8], conditioning=c, verbose=False, guidance_scale=scale,
                                  unconditional_conditioning=uc, eta=eta, x_T=xT, log_every_t=1)
    else:
        sampler.make_schedule(ddim_num_steps=S, ddim_discretize=disc, ddim_eta=eta, verbose=False)
        z, inter = sampler.ddim_sampling(c, (B, 4, 8, 8), x_T=xT, guidance_scale=scale, unconditional_conditioning=uc,
                                         log_every_t=1)
    assert [b for b, _ in calls] == g[f"{name}.call_batch"].tolist()
    assert [t for _, t in calls] == g[f"{name}.call_t"].tolist()
    np.testing.assert_array_equal(np.asarray(sampler.ddim_timesteps), g[f"{name}.ddim_timesteps"])
    np.testing.assert_allclose(np.asarray(sampler.ddim_sigmas, dtype=np.float64), g[f"{name}.ddim_sigmas"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(z.numpy(), g[f"{name}.z"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(torch.stack(inter["pred_x0"][1:]).numpy(), g[f"{name}.pred_x0"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(torch.stack(inter["x_inter"][1:]).numpy(), g[f"{name}.x_inter"], rtol=2e-5, atol=2e-6)


def test_scalar_guidance_scale_is_rejected_like_the_reference():
    """ddim.py:172-176 reads ``max_guide_scale`` before assigning it when ``guidance_scale`` is not a pair (UnboundLocalError in
    the reference's own loop): the product raises instead of inventing a meaning for it."""
    from adaprompt_amd.ldm.models.diffusion.ddim import DDIMSampler
    model = types.SimpleNamespace(num_timesteps=1000, betas=torch.zeros(1000),
                                  alphas_cumprod=torch.tensor(load_golden("ddim_loop")["alphas_cumprod"], dtype=torch.float32),
                                  apply_model=eps_closed_form)
    x = torch.zeros(1, 4, 8, 8)
    with pytest.raises(ValueError):
        DDIMSampler(model).sample(S=4, batch_size=1, shape=[4, 8, 8], conditioning=torch.zeros(1, 5, 6), verbose=False,
                                  guidance_scale=3.0, unconditional_conditioning=torch.zeros(1, 5, 6), x_T=x)

"""Inference-path oracle: VAE decoder restatement against vectors captured from the reference's Decoder
(tests/golden/vae_decode_*.npz), the DDIM schedule helpers against the reference's util functions
(ddim_params.npz), and the sampler loop (not constructible on CPU in the reference) against known answers."""
import numpy as np
import pytest
import torch

from adaprompt_amd import synth
from oracle import ddim_oracle as DO
from oracle import ldm_oracle as O
from conftest import load_golden, rel_err


@pytest.mark.parametrize("tag,dd", [("narrow", dict(synth.SD15_VAE_DD, ch=32, resolution=64))])
def test_vae_decoder_vs_reference(tag, dd):
    g = load_golden(f"vae_decode_{tag}")
    sd = synth.synthetic_vae_state_dict(dd, decoder=True)
    z = synth.synthetic_input(f"dec.{tag}.z", (int(g["B"]), 4, int(g["res"]) // 8, int(g["res"]) // 8), 0, 1.0)
    with torch.no_grad():
        img = O.autoencoder_decode(sd, dd, z)
    sub = int(g["sub"])
    assert rel_err(img[:, :, ::sub, ::sub], g["image"]) < 2e-5


@pytest.mark.slow
def test_vae_decoder_sd15_full_size_vs_reference():
    g = load_golden("vae_decode_sd15")
    dd = dict(synth.SD15_VAE_DD)
    sd = synth.synthetic_vae_state_dict(dd, decoder=True)
    z = synth.synthetic_input("dec.sd15.z", (1, 4, 64, 64), 0, 1.0)
    with torch.no_grad():
        img = O.autoencoder_decode(sd, dd, z)
    assert img.shape == (1, 3, 512, 512)
    assert rel_err(img[:, :, ::4, ::4], g["image"]) < 2e-5


def test_decoder_param_count():
    import math
    shapes = synth.vae_decoder_param_shapes(**synth.SD15_VAE_DD)
    assert sum(math.prod(s) for _, s in shapes) == 49_490_199          # 49.5 M incl. post_quant_conv


def test_ddim_parameters_vs_reference():
    g = load_golden("ddim_params")
    sched = O.make_schedule()
    ac = sched["alphas_cumprod"].double() if isinstance(sched, dict) else None
    assert ac is not None
    for S, eta, method in ((50, 0.0, "uniform"), (20, 0.5, "uniform"), (10, 0.0, "quad")):
        ts = DO.make_ddim_timesteps(method, S, 1000)
        tag = f"S{S}_{method}"
        assert ts.tolist() == g[tag + "_ts"].tolist()
        sig, al, alp = DO.make_ddim_sampling_parameters(ac, ts, eta)
        np.testing.assert_allclose(np.asarray(al, dtype=np.float64), g[tag + "_alphas"].numpy(), rtol=1e-6)
        np.testing.assert_allclose(np.asarray(alp, dtype=np.float64), g[tag + "_alphas_prev"].numpy(), rtol=1e-6)
        np.testing.assert_allclose(np.asarray(sig, dtype=np.float64), g[tag + "_sigmas"].numpy(), rtol=1e-5, atol=1e-12)
    assert DO.make_ddim_timesteps("uniform", 50, 1000)[:3].tolist() == [1, 21, 41]       # ddim.py:30-35 docstring


def test_p_sample_with_the_true_noise_walks_back_along_the_forward_process():
    """known answer: if eps_fn returns the noise that produced x_t from x0, pred_x0 == x0 and (eta = 0) x_{t-1} is
    exactly q_sample(x0, t_prev, same noise)."""
    sched = O.make_schedule()
    ac = sched["alphas_cumprod"].double()
    ts = DO.make_ddim_timesteps("uniform", 50, 1000)
    params = DO.make_ddim_sampling_parameters(ac, ts, 0.0)
    g = torch.Generator().manual_seed(0)
    x0, eps = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    index = 30
    a_t, a_prev = float(params[1][index]), float(params[2][index])
    x_t = a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * eps
    x_prev, pred = DO.p_sample_ddim(lambda x, t, c: eps, x_t, None, torch.full((2,), int(ts[index])), index, params)
    assert torch.allclose(pred, x0, atol=1e-5)
    assert torch.allclose(x_prev, a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps, atol=1e-5)


def test_guidance_combination_and_annealing():
    sched = O.make_schedule()
    ac = sched["alphas_cumprod"].double()
    ts = DO.make_ddim_timesteps("uniform", 10, 1000)
    params = DO.make_ddim_sampling_parameters(ac, ts, 0.0)
    x = torch.ones(2, 4, 4, 4)
    calls = []

    def eps_fn(xin, tin, c):
        calls.append((xin.shape[0], c))
        return torch.cat([torch.full((2, 4, 4, 4), 3.0), torch.full((2, 4, 4, 4), 1.0)]) if xin.shape[0] == 4 \
            else torch.full((2, 4, 4, 4), 3.0)

    c, uc = torch.zeros(2, 5, 8), torch.ones(2, 5, 8)
    index = 4
    _, pred = DO.p_sample_ddim(eps_fn, x, c, torch.full((2,), int(ts[index])), index, params, 2.5, uc)
    assert calls[-1][0] == 4 and torch.equal(calls[-1][1][:2], c) and torch.equal(calls[-1][1][2:], uc)   # (cond, uncond)
    e = 1.0 + 2.5 * (3.0 - 1.0)
    a_t = float(params[1][index])
    assert torch.allclose(pred, (x - (1 - a_t) ** 0.5 * e) / a_t ** 0.5, atol=1e-5)
    _, pred1 = DO.p_sample_ddim(eps_fn, x, c, torch.full((2,), int(ts[index])), index, params, 1.0, uc)
    assert calls[-1][0] == 2                                     # scale 1: no doubled batch (ddim.py:226-227)
    sc = DO.guidance_schedule((10, 4), 50)
    assert sc[0] == 10 and abs(sc[-1] - 4) < 1e-9 and all(a > b for a, b in zip(sc, sc[1:]))

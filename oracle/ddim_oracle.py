"""CPU restatement of the reference's DDIM sampler (ldm/models/diffusion/ddim.py) for the inference path
(SURVEY 8f-2 / config 5).

TEST INFRASTRUCTURE ONLY.  ``make_ddim_timesteps`` / ``make_ddim_sampling_parameters``
(ldm/modules/diffusionmodules/util.py:46-77) are pinned by tests/golden/ddim_params.npz; the sampler loop --
``ddim_sampling`` / ``p_sample_ddim`` (ddim.py:135-292) -- by tests/golden/ddim_loop.npz, captured from the reference's
OWN ``DDIMSampler`` run on CPU (tests/golden/make_golden_ddim.py: its ``register_buffer``, the only place that names
``torch.device("cuda")``, replaced on the instance; a closed-form eps model): 4 cases incl. guidance annealing on the doubled
batch, eta > 0 and the "quad" discretisation (tests/test_ddim_golden.py holds this file AND the product's sampler to them).
"""
import numpy as np
import torch


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps):
    """util.py:46-60: 'uniform' = range(0, T, T // S) + 1; 'quad' = (linspace(0, sqrt(.8 T), S) ** 2).astype(int) + 1."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ts = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ts = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(ddim_discr_method)
    return ts + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """util.py:63-77 (alphacums a torch tensor, as the sampler passes it): -> sigmas, alphas, alphas_prev."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def p_sample_ddim(eps_fn, x, c, t, index, params, guidance_scale=1.0, unconditional_conditioning=None,
                  temperature=1.0, noise=None):
    """ddim.py:220-292.  ``eps_fn(x, t, cond) -> eps``; cond is the reference's triple (emb, c_in, extra_info) or a
    tensor.  With an unconditional conditioning and scale != 1 the batch is doubled in the order (cond, uncond) and
    eps = eps_u + s * (eps_c - eps_u).  ``noise``: the unscaled N(0,1) draw (only matters when sigma > 0)."""
    sigmas, alphas, alphas_prev = params
    b = x.shape[0]
    if unconditional_conditioning is None or guidance_scale == 1.0:
        e_t = eps_fn(x, t, c)
    else:
        x_in, t_in = torch.cat([x] * 2), torch.cat([t] * 2)
        if isinstance(c, tuple):
            c_c, c_in_c, extra_info = c
            c_u, c_in_u, _ = unconditional_conditioning
            c2 = (torch.cat([c_c, c_u]), sum([c_in_c, c_in_u], []), extra_info)
        else:
            c2 = torch.cat([c, unconditional_conditioning])
        e_t, e_t_uncond = eps_fn(x_in, t_in, c2).chunk(2)
        e_t = e_t_uncond + guidance_scale * (e_t - e_t_uncond)
    f = lambda v: torch.full((b, 1, 1, 1), float(v))
    a_t, a_prev, sigma_t = f(alphas[index]), f(alphas_prev[index]), f(sigmas[index])
    sqrt_one_minus_at = f(np.sqrt(1.0 - float(alphas[index])))
    pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    if noise is None:
        noise = torch.zeros_like(x)
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * noise * temperature
    return x_prev, pred_x0


def guidance_schedule(guidance_scale, total_steps):
    """ddim.py:170-183, 213-216: guidance annealing.  ``guidance_scale`` = (max, min) as the script passes it
    (``--scale 10 4``); the scale used at loop iteration i is max - i * (max - min) / (total_steps - 1)."""
    max_s, min_s = guidance_scale
    delta = (max_s - min_s) / (total_steps - 1)
    out, g = [], max_s
    for i in range(total_steps):
        out.append(g)
        g = g - delta if i <= total_steps - 1 else 1
    return out


def ddim_sampling(eps_fn, alphas_cumprod, cond, x_T, S, guidance_scale, unconditional_conditioning=None, eta=0.0,
                  ddim_discretize="uniform", noises=None):
    """ddim.py:135-218 with x_T given, no mask / x0, ddim_use_original_steps=False.  -> (x_0 estimate after the last
    step, list of pred_x0 per step)."""
    T = alphas_cumprod.shape[0]
    ts = make_ddim_timesteps(ddim_discretize, S, T)
    params = make_ddim_sampling_parameters(alphas_cumprod, ts, eta)
    total = ts.shape[0]
    scales = guidance_schedule(guidance_scale, total)
    img, preds = x_T, []
    b = x_T.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        img, pred_x0 = p_sample_ddim(eps_fn, img, cond, t, index, params, scales[i], unconditional_conditioning,
                                     noise=None if noises is None else noises[i])
        preds.append(pred_x0)
    return img, preds

"""Golden vectors from the reference's OWN ``DDIMSampler`` (ldm/models/diffusion/ddim.py:12-292): ``sample`` ->
``make_schedule`` -> ``ddim_sampling`` -> ``p_sample_ddim`` over a few guided steps, incl. the guidance annealing, the doubled
(cond, uncond) batch and eta > 0.

The class is constructible on CPU after all: its ``register_buffer`` (ddim.py:22-26, the only place that names
``torch.device("cuda")``) is an instance method, replaced on the instance by a plain ``setattr``; nothing of the sampler's
arithmetic is touched.  The model is a bare object with what the sampler reads (``num_timesteps``, ``betas``,
``alphas_cumprod``, ``alphas_cumprod_prev`` from the reference's own ``make_beta_schedule``, ``device``) and a CLOSED-FORM
``apply_model`` -- eps = tanh(0.8 x + 0.1 ctx_mean) * (0.5 + t / 2000) + 0.05 sin(3 x) -- that tests/test_ddim_golden.py
restates, so the fixture holds only inputs and outputs.

    python tests/golden/make_golden_ddim.py        # writes tests/golden/ddim_loop.npz (own process: sys.modules stand-ins)
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "ddim_loop.npz")


def import_reference_ddim():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    # import-time names of ldm/util.py that the sampler never touches
    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils", make_grid=None, draw_bounding_boxes=None)
    oc = stub("omegaconf")
    oc.listconfig = stub("omegaconf.listconfig", ListConfig=list)
    sys.path.insert(0, REF)
    import ldm.models.diffusion.ddim as DD
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    assert DD.__file__.startswith(REF)
    return DD, make_beta_schedule


def eps_closed_form(x, t, ctx):
    """the stand-in UNet: any deterministic function of (x, t, context) will do; this one is non-linear in x and differs
    between the conditional and the unconditional half."""
    cm = ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1)
    return torch.tanh(0.8 * x + 0.1 * cm) * (0.5 + t.view(-1, 1, 1, 1).float() / 2000.0) + 0.05 * torch.sin(3.0 * x)


# (a SCALAR guidance_scale is not a case: the reference's own loop raises UnboundLocalError on it, ddim.py:172-176)
CASES = {   # name: (S, batch, guidance_scale, eta, ddim_discretize, seed, tuple-conditioning?)
    "guided4": (4, 2, [10.0, 4.0], 0.0, "uniform", 11, True),
    "guided5_eta": (5, 3, [7.5, 2.0], 0.6, "uniform", 12, True),
    "plain8": (8, 2, [1.0, 1.0], 0.0, "uniform", 13, False),         # scale 1: no doubled batch (ddim.py:226-227)
    "quad6": (6, 1, [3.0, 2.0], 0.0, "quad", 14, True),
}


def main():
    DD, make_beta_schedule = import_reference_ddim()
    out = {}
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012, cosine_s=8e-3)       # yaml:9-10
    ac = np.cumprod(1.0 - betas, axis=0)
    acp = np.append(1.0, ac[:-1])
    out["alphas_cumprod"] = ac.astype(np.float64)
    for name, (S, B, scale, eta, disc, seed, as_tuple) in CASES.items():
        g = torch.Generator().manual_seed(seed)
        model = types.SimpleNamespace(num_timesteps=1000, device=torch.device("cpu"), betas=torch.tensor(betas, dtype=torch.float32),
                                      alphas_cumprod=torch.tensor(ac, dtype=torch.float32),
                                      alphas_cumprod_prev=torch.tensor(acp, dtype=torch.float32))
        calls = []

        def apply_model(x, t, c, calls=calls):
            ctx = c[0] if isinstance(c, tuple) else c
            calls.append((tuple(x.shape), int(t[0])))
            return eps_closed_form(x, t, ctx)
        model.apply_model = apply_model
        sampler = DD.DDIMSampler(model)
        sampler.register_buffer = lambda n, a, s=sampler: setattr(s, n, a)         # ddim.py:22-26 without the "cuda" move
        xT = torch.randn(B, 4, 8, 8, generator=g)
        ctx = torch.randn(B, 5, 6, generator=g)
        uctx = torch.randn(B, 5, 6, generator=g) * 0.3
        c = (ctx, ["a"] * B, {"k": 1}) if as_tuple else ctx
        uc = (uctx, [""] * B, {"k": 1}) if as_tuple else uctx
        torch.manual_seed(seed + 100)            # eta > 0: noise_like draws from the global generator (util.py noise_like)
        if disc == "uniform":
            z, inter = sampler.sample(S=S, batch_size=B, shape=[4, 8, 8], conditioning=c, verbose=False, guidance_scale=scale,
                                      unconditional_conditioning=uc, eta=eta, x_T=xT, log_every_t=1)
        else:                                    # sample() always discretises uniformly: its two steps, spelled out
            sampler.make_schedule(ddim_num_steps=S, ddim_discretize=disc, ddim_eta=eta, verbose=False)
            z, inter = sampler.ddim_sampling(c, (B, 4, 8, 8), x_T=xT, guidance_scale=scale, unconditional_conditioning=uc,
                                             log_every_t=1)
        out[f"{name}.x_T"], out[f"{name}.ctx"], out[f"{name}.uctx"] = xT.numpy(), ctx.numpy(), uctx.numpy()
        out[f"{name}.z"] = z.numpy()
        out[f"{name}.pred_x0"] = torch.stack(inter["pred_x0"][1:]).numpy()
        out[f"{name}.x_inter"] = torch.stack(inter["x_inter"][1:]).numpy()
        out[f"{name}.call_batch"] = np.asarray([s[0] for s, _ in calls])
        out[f"{name}.call_t"] = np.asarray([t for _, t in calls])
        out[f"{name}.ddim_timesteps"] = np.asarray(sampler.ddim_timesteps)
        out[f"{name}.ddim_sigmas"] = np.asarray(sampler.ddim_sigmas, dtype=np.float64)
        # the noise the reference drew (eta > 0), replayed from the same global seed in the draw order of the loop
        if eta > 0:
            torch.manual_seed(seed + 100)
            out[f"{name}.noises"] = torch.stack([torch.randn(B, 4, 8, 8) for _ in range(S)]).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith(".z")})


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Capture golden vectors from the reference's own modules (runs ONLY in the build
container, where /root/reference is mounted; the GPU box never sees the reference).

    python tests/golden/make_golden.py [--full]      # writes tests/golden/*.npz

The reference modules are imported unmodified from /root/reference on CPU (recipe:
SURVEY.md Appendix D -- two in-memory no-op stubs for ``torchvision.utils`` and
``omegaconf.listconfig`` that the arithmetic never touches).  They are filled with the
deterministic synthetic state-dict of ``adaprompt_amd.synth`` (a function of tensor name,
shape and seed), fed seeded inputs, and their outputs are stored.  A fixture holds only
data: config, seeds, inputs too small to regenerate from a seed, and expected outputs.

``--full`` additionally captures the full-size SD-1.5 UNet (859.5 M parameters) and VAE
encoder at 512x512; takes ~1 min and ~10 GB of RAM.
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

import importlib.util

import numpy as np
import torch

# load adaprompt_amd/synth.py by path: putting the repo root on sys.path would let this repo's own
# ``ldm`` alias package shadow the reference's (namespace) ``ldm`` package.
_spec = importlib.util.spec_from_file_location("adaprompt_synth", os.path.join(ROOT, "adaprompt_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
sys.path = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]

REF = "/root/reference"


def import_reference():
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tvu.draw_bounding_boxes = lambda *a, **k: None
    tv.utils = tvu
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu
    oc = types.ModuleType("omegaconf")
    ocl = types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass
    ocl.ListConfig = ListConfig
    oc.listconfig = ocl
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = ocl
    sys.path.insert(0, REF)
    from ldm.modules.diffusionmodules import openaimodel, model, util
    from ldm.modules import attention
    from ldm.modules.distributions import distributions
    return openaimodel, model, util, attention, distributions


def fill(module, prefix, seed):
    """load the synthetic state dict (by name) into a reference module."""
    sd = {k: synth.synthetic_tensor(prefix + k, v.shape, seed) for k, v in module.state_dict().items()}
    module.load_state_dict(sd, strict=True)
    return module.eval()


ONLY = ""


def save(name, **arrs):
    if ONLY and ONLY not in name:
        return
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KB")


def ellipse_mask(B, H, W):
    """centred ellipse ~35 % area as float [B,1,H,W]."""
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    m = ((xx / 0.62) ** 2 + (yy / 0.72) ** 2 <= 1.0).float()
    return m[None, None].repeat(B, 1, 1, 1)


def border_mask(B, H, W, border):
    m = torch.zeros(B, 1, H, W)
    m[:, :, border:H - border, border:W - border] = 1
    return m


def subsample_act(key, ten):
    """strided subsample of a captured activation (shared with tests/test_oracle_golden.py)."""
    if key == "outfeat":                       # [B, C, H, W]
        return ten[:, ::8, ::4, ::4] if ten.shape[-1] > 8 else ten[:, ::8]
    n = ten.shape[2]                           # attn/attnscore [B,h,N,M], q [B,h,N,d]
    return ten[:, ::4, ::max(1, n // 32)]


PRODIGY_SHAPES = [(37, 19), (129,), (5, 3, 3, 3), (1,)]
PRODIGY_CASES = {
    # the shipped zero-shot config: betas (0.9, 0.999), bias correction on, d_coef from the yaml (ddpm.py:5207-5217)
    "zs": dict(betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, safeguard_warmup=False, weight_decay=0.0),
    "fast_wd": dict(betas=(0.985, 0.993), d_coef=5.0, use_bias_correction=True, safeguard_warmup=True,
                    weight_decay=0.01),
    "plain_growth": dict(betas=(0.9, 0.999), d_coef=1.0, use_bias_correction=False, safeguard_warmup=False,
                         weight_decay=0.0, growth_rate=1.5),
    "coupled_wd": dict(betas=(0.9, 0.99), d_coef=1.0, use_bias_correction=True, safeguard_warmup=False,
                       weight_decay=0.02, decouple=False),
}


def prodigy_grad(case, step, i, shape):
    """seeded gradient of parameter i at optimiser step ``step`` (shared with the tests by name)."""
    g = synth.synthetic_input(f"prodigy.{case}.g{i}.s{step}", shape, 0, 1.0)
    return g * (0.01 if step != 3 else 3.0)       # step 3: a large gradient so that the 0.5 clip engages


def run_prodigy():
    from ldm.prodigy import Prodigy
    from ldm.util import SequentialLR2
    from torch.optim.lr_scheduler import ConstantLR, PolynomialLR
    nsteps = 8
    for case, kw in PRODIGY_CASES.items():
        for clip in (0.0, 0.5):
            ps = [torch.nn.Parameter(synth.synthetic_input(f"prodigy.{case}.p{i}", sh, 0, 0.3).clone())
                  for i, sh in enumerate(PRODIGY_SHAPES)]
            opt = Prodigy(ps, lr=1.0, **kw)
            # ddpm.py:5219-5247 with max_steps=8, warm_up_steps=2, one Linear cycle
            warm = ConstantLR(opt, factor=1.0, total_iters=2)
            lin = PolynomialLR(opt, power=1, total_iters=(nsteps - 2) * 1.1)
            sched = SequentialLR2(opt, schedulers=[warm, lin], milestones=[2])
            traj, ds, lrs, norms = [], [], [], []
            for step in range(nsteps):
                for i, p in enumerate(ps):
                    p.grad = prodigy_grad(case, step, i, p.shape).clone()
                if step == 5:                           # an all-zero gradient step: d_denom == 0 only if s == 0,
                    for p in ps:                        # so here it just decays the moments
                        p.grad.zero_()
                lrs.append(opt.param_groups[0]["lr"])
                if clip > 0:
                    norms.append(float(torch.nn.utils.clip_grad_norm_(ps, clip)))
                opt.step()
                sched.step()
                g0 = opt.param_groups[0]
                traj.append(torch.cat([p.detach().flatten() for p in ps]).clone())
                ds.append([g0["d"], g0["d_max"], g0["d_numerator"], g0.get("d_denom", 0.0), g0.get("d_hat", 0.0),
                           float(g0["k"])])
            st = [opt.state[p] for p in ps]
            save(f"prodigy_{case}_clip{int(clip * 10)}", case=case, clip=clip, nsteps=nsteps,
                 params=torch.stack(traj), dstate=np.array(ds, dtype=np.float64), lrs=np.array(lrs, dtype=np.float64),
                 grad_norms=np.array(norms, dtype=np.float64),
                 exp_avg=torch.cat([s_["exp_avg"].flatten() for s_ in st]),
                 exp_avg_sq=torch.cat([s_["exp_avg_sq"].flatten() for s_ in st]),
                 s=torch.cat([s_["s"].flatten() for s_ in st]),
                 p0=torch.cat([s_["p0"].flatten() for s_ in st]))
    # a first step whose gradients are all zero returns before touching anything (prodigy.py:200-201)
    ps = [torch.nn.Parameter(synth.synthetic_input("prodigy.zero.p0", (11,), 0, 0.3).clone())]
    opt = Prodigy(ps, lr=1.0)
    ps[0].grad = torch.zeros(11)
    opt.step()
    save("prodigy_zero_first_step", params=ps[0].detach(), k=opt.param_groups[0]["k"], d=opt.param_groups[0]["d"])


def run_anneal():
    """ldm/util.py:1468-1530 with seeded ``random`` / ``np.random`` (the restatement must consume them identically)."""
    import random
    from ldm.util import probably_anneal_t, anneal_value, anneal_array
    cases, outs = [], []
    for seed in range(12):
        random.seed(100 + seed)
        np.random.seed(200 + seed)
        tp = [0.0, 0.13, 0.5, 0.97][seed % 4]
        t = torch.tensor([[3, 250, 640, 999], [0, 998, 500, 77], [850, 851, 10, 420]][seed % 3])
        rr, kp = [((1, 1.3), (0.4, 0.2)), ((0.8, 1.0), (0.5, 0.3))][seed % 2]
        out = probably_anneal_t(t, tp, 1000, ratio_range=rr, keep_prob_range=kp)
        cases.append([seed, tp, rr[0], rr[1], kp[0], kp[1]] + t.tolist())
        outs.append(out.tolist())
    av = [anneal_value(tp, fp, (0.2, 0.9)) for tp in (0.0, 0.3, 0.5, 1.0) for fp in (0.5, 1.0)]
    aa = anneal_array(0.25, 0.5, [0.4, 0.3, 0.2, 0.1], [0.1, 0.2, 0.3, 0.4])
    save("anneal_t", cases=np.array(cases, dtype=np.float64), outs=np.array(outs, dtype=np.int64),
         anneal_values=np.array(av, dtype=np.float64), anneal_array=np.array(aa, dtype=np.float64))


def run_regs():
    """the recon iteration's regulariser helpers, ldm/util.py: ortho_subtract :280, demean :425, calc_ref_cosine_loss :437,
    normalized_sum :2110, normalize_dict_values :1423, calc_prompt_emb_delta_loss :2037 -- values AND gradients."""
    from ldm.util import (calc_prompt_emb_delta_loss, calc_ref_cosine_loss, normalize_dict_values, normalized_sum,
                          ortho_subtract)
    with torch.enable_grad():
        _run_regs(calc_prompt_emb_delta_loss, calc_ref_cosine_loss, normalize_dict_values, normalized_sum, ortho_subtract)


def _run_regs(calc_prompt_emb_delta_loss, calc_ref_cosine_loss, normalize_dict_values, normalized_sum, ortho_subtract):
    out = {}
    a = synth.synthetic_input("regs.a", (3, 5, 7, 24))
    b = synth.synthetic_input("regs.b", (3, 5, 7, 24)) + 0.3 * a
    out["ortho"] = ortho_subtract(a, b)
    # calc_ref_cosine_loss over its switches; gradients w.r.t. both arguments
    cases = [dict(exponent=2, do_demean_first=True, first_n_dims_to_flatten=3, ref_grad_scale=0.05, aim_to_align=True),
             dict(exponent=2, do_demean_first=False, first_n_dims_to_flatten=3, ref_grad_scale=0, aim_to_align=True),
             dict(exponent=3, do_demean_first=True, first_n_dims_to_flatten=2, ref_grad_scale=1, aim_to_align=False),
             dict(exponent=2, do_demean_first=True, first_n_dims_to_flatten=1, ref_grad_scale=1, aim_to_align=True),
             dict(exponent=2, do_demean_first=True, first_n_dims_to_flatten=3, ref_grad_scale=0.1, aim_to_align=True,
                  margin=0.2)]
    emb_mask = (synth.synthetic_input("regs.m", (3, 1, 7, 1)) > -0.3).float() * 0.5 + \
               (synth.synthetic_input("regs.m2", (3, 1, 7, 1)) > 0.2).float() * 0.5
    batch_mask = torch.tensor([1.0, 0.0, 1.0])
    for ci, kw in enumerate(cases):
        for mi, (em, bm) in enumerate(((None, None), (emb_mask, None), (emb_mask, batch_mask))):
            if kw["first_n_dims_to_flatten"] != 3 and em is not None:
                continue
            d = a.clone().requires_grad_(True)
            r = b.clone().requires_grad_(True)
            loss = calc_ref_cosine_loss(d, r, batch_mask=bm, emb_mask=em, **kw)
            loss.backward()
            out[f"cos{ci}_{mi}_loss"] = loss.detach()
            out[f"cos{ci}_{mi}_gd"] = d.grad
            out[f"cos{ci}_{mi}_gr"] = r.grad if r.grad is not None else torch.zeros_like(r)
    # calc_prompt_emb_delta_loss: [4*BS, 16, 77, 16] embeddings, mask 1 / 0.5 (padding) as the embedder produces
    BS = 2
    emb = synth.synthetic_input("regs.emb", (4 * BS, 16, 77, 16)).requires_grad_(True)
    pm = torch.ones(4 * BS, 77, 1)
    for i, n in enumerate((20, 31, 26, 31, 20, 31, 26, 31)):        # single / comp prompt lengths; the rest is padding
        pm[i, n:] = 0.5
    loss = calc_prompt_emb_delta_loss(emb, pm.clone())
    loss.backward()
    out["pdelta_loss"], out["pdelta_grad"], out["pdelta_mask"] = loss.detach(), emb.grad, pm
    loss0 = calc_prompt_emb_delta_loss(emb.detach(), None)
    out["pdelta_loss_nomask"] = loss0
    nd = normalize_dict_values({8: 0.5, 12: 1.0, 16: 1.0, 19: 0.5, 22: 0.25})
    out["ndict_keys"], out["ndict_vals"] = np.array(list(nd.keys())), np.array(list(nd.values()), dtype=np.float64)
    ls = [torch.tensor(0.3), torch.tensor(1.7), torch.tensor(0.02)]
    out["nsum0"], out["nsum05"] = normalized_sum(ls), normalized_sum(ls, norm_pow=0.5)
    save("regs_util", **out)


def run_regs_masks():
    """ldm/util.py: masked_mean :1450, resize_mask_for_feat_or_attn :1570, sel_emb_attns_by_indices :1945 -- the helpers
    of the fg/bg complementary loss (ddpm.py:3932-4258)."""
    from ldm.util import masked_mean, resize_mask_for_feat_or_attn, sel_emb_attns_by_indices
    out = {}
    ts = synth.synthetic_input("regm.ts", (3, 8, 64))
    mask = (synth.synthetic_input("regm.mask", (3, 1, 64)) > 0.2).float()
    iw = torch.tensor([1.0, 0.0, 1.0])
    out["mm_all"] = masked_mean(ts, mask)
    out["mm_dim"] = masked_mean(ts, mask, dim=(1, 2), keepdim=True)
    out["mm_iw"] = masked_mean(ts, ts > 0.1, instance_weights=iw)
    out["mm_none"] = masked_mean(ts, None, instance_weights=iw)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
    m64 = torch.stack([((xx / 0.5) ** 2 + (yy / 0.7) ** 2 <= 1).float(), ((xx - 0.3).abs() + (yy + 0.2).abs() <= 0.12).float()])[:, None]
    for h in (8, 16, 32):
        out[f"rm_{h}"] = resize_mask_for_feat_or_attn(torch.zeros(2, 8, h * h), m64, "fg_mask", num_spatial_dims=1,
                                                       mode="nearest|bilinear")
    out["rm_near_16"] = resize_mask_for_feat_or_attn(torch.zeros(2, 8, 256), m64, "fg_mask", num_spatial_dims=1, mode="nearest")
    attn = synth.synthetic_input("regm.attn", (3, 77, 8, 64))
    subj = (torch.arange(3).repeat_interleave(4), torch.tensor([5, 6, 7, 8, 6, 7, 8, 9, 5, 6, 7, 8]))
    bg = (torch.arange(3), torch.tensor([11, 12, 34]))
    out["sel_sum"] = sel_emb_attns_by_indices(attn, subj, do_sum=True, do_mean=False, do_sqrt_norm=False)
    out["sel_sqrt"] = sel_emb_attns_by_indices(attn, subj, do_sum=True, do_mean=False, do_sqrt_norm=True)
    out["sel_bg"] = sel_emb_attns_by_indices(attn, bg, do_sum=True, do_mean=False, do_sqrt_norm=False)
    save("regs_masks", **out)


def run_cond_helpers():
    """the host-side helpers of the conditioning assembly (LatentDiffusion.forward, ddpm.py:1710-2042): ldm/util.py
    repeat_selected_instances :1410, add_noise_to_tensor :2123, anneal_add_noise_to_embedding :2144,
    distribute_embedding_to_M_tokens(_by_dict) :882-932 -- with seeded ``random`` / ``np.random`` / torch RNG."""
    import random
    from ldm.util import (add_noise_to_tensor, anneal_add_noise_to_embedding, distribute_embedding_to_M_tokens,
                          distribute_embedding_to_M_tokens_by_dict, repeat_selected_instances)
    out = {}
    a = synth.synthetic_input("ch.a", (4, 3, 5))
    b = synth.synthetic_input("ch.b", (4, 7))
    r = repeat_selected_instances(slice(0, 2), 3, a, None, b)
    out["rep_a"], out["rep_b"] = r[0], r[2]
    emb = synth.synthetic_input("ch.emb", (6, 32))
    torch.manual_seed(11)
    out["noise_rel"] = add_noise_to_tensor(emb, 0.1, noise_std_is_relative=True, keep_norm=False)
    torch.manual_seed(12)
    out["noise_keepnorm"] = add_noise_to_tensor(emb, 0.05, noise_std_is_relative=False, keep_norm=True)
    for i, (tp, prob) in enumerate(((0.0, 1.0), (0.6, 0.5), (0.3, 0.0))):
        random.seed(30 + i)
        np.random.seed(40 + i)
        torch.manual_seed(50 + i)
        out[f"anneal_noise_{i}"] = anneal_add_noise_to_embedding(emb, tp, begin_noise_std_range=[0.02, 0.06],
                                                                 end_noise_std_range=[0.01, 0.03], add_noise_prob=prob)
    random.seed(33)
    np.random.seed(43)
    torch.manual_seed(53)
    out["anneal_noise_noend"] = anneal_add_noise_to_embedding(emb, 0.5, begin_noise_std_range=[0.02, 0.06],
                                                              end_noise_std_range=None, add_noise_prob=1.0)
    te = synth.synthetic_input("ch.te", (16, 77, 24))
    idx = torch.tensor([5, 6, 7, 8, 5, 6, 7, 8])
    out["dist_sqrt"] = distribute_embedding_to_M_tokens(te, idx)
    out["dist_M"] = distribute_embedding_to_M_tokens(te, idx, divide_scheme="M")
    out["dist_single"] = distribute_embedding_to_M_tokens(te, torch.tensor([9]))
    d = {"z": (torch.zeros(4, dtype=torch.long), torch.tensor([5, 6, 7, 8])), "y": None,
         "w": (torch.zeros(1, dtype=torch.long), torch.tensor([20]))}
    out["dist_dict"] = distribute_embedding_to_M_tokens_by_dict(te, d)
    save("cond_helpers", **out)


def run_decoder_and_ddim(model, util, full):
    """VAE Decoder + post_quant_conv (model.py:502-608, autoencoder.py:330-333) and the DDIM schedule helpers
    (util.py:46-77)."""
    def run_dec(dd, B, res, tag, sub):
        dec = model.Decoder(**dd)
        fill(dec, "first_stage_model.decoder.", 0)
        pqc = torch.nn.Conv2d(4, dd["z_channels"], 1)
        fill(pqc, "first_stage_model.post_quant_conv.", 0)
        z = synth.synthetic_input(f"dec.{tag}.z", (B, 4, res // 8, res // 8), 0, 1.0)
        img = dec(pqc(z))
        save(f"vae_decode_{tag}", B=B, res=res, sub=sub, image=img[:, :, ::sub, ::sub])

    run_dec(dict(synth.SD15_VAE_DD, ch=32, resolution=64), 2, 64, "narrow", 1)
    if full:
        run_dec(dict(synth.SD15_VAE_DD), 1, 512, "sd15", 4)
    betas = util.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = torch.tensor(np.cumprod(1.0 - betas, axis=0))
    out = {}
    for S, eta, method in ((50, 0.0, "uniform"), (20, 0.5, "uniform"), (10, 0.0, "quad")):
        ts = util.make_ddim_timesteps(method, S, 1000, verbose=False)
        sig, al, alp = util.make_ddim_sampling_parameters(ac, ts, eta, verbose=False)
        tag = f"S{S}_{method}"
        out[tag + "_ts"] = np.asarray(ts)
        out[tag + "_sigmas"] = np.asarray(sig, dtype=np.float64)
        out[tag + "_alphas"] = np.asarray(al, dtype=np.float64)
        out[tag + "_alphas_prev"] = np.asarray(alp, dtype=np.float64)
    save("ddim_params", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="", help="only (re)write fixtures whose name contains this substring")
    args = ap.parse_args()
    global ONLY
    ONLY = args.only
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    openaimodel, model, util, attention, distributions = import_reference()
    seed = 0

    # ---------------- per-op: timestep embedding, GroupNorm32+SiLU ----------------------
    t = torch.tensor([0, 1, 500, 999])
    save("op_timestep_embedding", t=t, out=util.timestep_embedding(t, 320))

    for tag, (C, H, eps) in {"c320": (320, 16, 1e-5), "c1920": (1920, 8, 1e-5),
                             "c640e6": (640, 8, 1e-6)}.items():
        gn = util.GroupNorm32(32, C, eps=eps)
        fill(gn, f"gn.{tag}.", seed)
        x = synth.synthetic_input(f"gn.{tag}", (2, C, H, H), seed) * 1.5 + 0.3
        y = gn(x)
        y = y * torch.sigmoid(y)
        save(f"op_groupnorm_silu_{tag}", C=C, H=H, eps=eps, out=y)

    # ---------------- per-op: CrossAttention ---------------------------------------------
    # (C, N, M ctx tokens or 0 for self, split K/V, mask)
    cases = {
        "self_c320_n256": (320, 256, 0, False, False),
        "self_c320_n256_mask": (320, 256, 0, False, True),
        "self_c640_n64": (640, 64, 0, False, False),
        "self_c1280_n64": (1280, 64, 0, False, False),
        "cross_c320_n256_m77": (320, 256, 77, False, False),
        "cross_c1280_n64_m77_split": (1280, 64, 77, True, False),
    }
    for tag, (C, N, M, split, use_mask) in cases.items():
        ca = attention.CrossAttention(query_dim=C, context_dim=768 if M else None, heads=8,
                                      dim_head=C // 8)
        fill(ca, f"ca.{tag}.", seed)
        ca.save_attn_vars = True
        x = synth.synthetic_input(f"ca.{tag}.x", (2, N, C), seed)
        ctx = None
        if M:
            if split:
                c = synth.synthetic_input(f"ca.{tag}.ctx", (2, 2 * M, 768), seed)
                v, k = c.chunk(2, dim=1)
                ctx = (v, k)
            else:
                ctx = synth.synthetic_input(f"ca.{tag}.ctx", (2, M, 768), seed)
        mask = None
        if use_mask:
            hw = int(N ** 0.5)
            mask = border_mask(2, hw, hw, 2)
            mask[1] = ellipse_mask(1, hw, hw)[0]
        out = ca(x, context=ctx, mask=mask)
        acts = ca.cached_activations
        save(f"op_cross_attention_{tag}", C=C, N=N, M=M, split=split, use_mask=use_mask, out=out,
             q=acts["q"][:, :, ::4], attn=acts["attn"][:, :, ::16],
             attnscore=acts["attnscore"][:, :, ::16],
             mask=mask if mask is not None else np.zeros(0))

    # ---------------- per-op: ResBlock / Down / Up / SpatialTransformer -------------------
    for tag, (ci, co, H) in {"c320_320": (320, 320, 16), "c640_320_skip": (640, 320, 8),
                             "c1920_640_skip": (1920, 640, 4)}.items():
        rb = openaimodel.ResBlock(ci, 1280, 0, out_channels=co, dims=2, use_checkpoint=True)
        fill(rb, f"res.{tag}.", seed)
        x = synth.synthetic_input(f"res.{tag}.x", (2, ci, H, H), seed)
        emb = synth.synthetic_input(f"res.{tag}.emb", (2, 1280), seed)
        save(f"op_resblock_{tag}", ci=ci, co=co, H=H, out=rb(x, emb))

    dn = openaimodel.Downsample(320, True, dims=2, out_channels=320)
    fill(dn, "down.", seed)
    x = synth.synthetic_input("down.x", (2, 320, 16, 16), seed)
    save("op_downsample_c320", out=dn(x))
    up = openaimodel.Upsample(320, True, dims=2, out_channels=320)
    fill(up, "up.", seed)
    x = synth.synthetic_input("up.x", (2, 320, 8, 8), seed)
    save("op_upsample_c320", out=up(x))

    for tag, (C, H, use_mask) in {"c320_h16": (320, 16, False), "c320_h16_mask": (320, 16, True),
                                  "c640_h8": (640, 8, False)}.items():
        st = attention.SpatialTransformer(C, 8, C // 8, depth=1, context_dim=768)
        fill(st, f"st.{tag}.", seed)
        x = synth.synthetic_input(f"st.{tag}.x", (2, C, H, H), seed)
        ctx = synth.synthetic_input(f"st.{tag}.ctx", (2, 77, 768), seed)
        mask = border_mask(2, 64, 64, 9) if use_mask else None
        out = st(x, context=lambda: ((ctx, ctx), None), mask=mask)
        save(f"op_spatial_transformer_{tag}", C=C, H=H, use_mask=use_mask, out=out)

    # ---------------- per-op: VAE blocks ---------------------------------------------------
    for tag, (ci, co, H) in {"c128_128": (128, 128, 16), "c128_256_nin": (128, 256, 16)}.items():
        rb = model.ResnetBlock(in_channels=ci, out_channels=co, temb_channels=0, dropout=0.0)
        fill(rb, f"vres.{tag}.", seed)
        x = synth.synthetic_input(f"vres.{tag}.x", (2, ci, H, H), seed)
        save(f"op_vae_resnet_{tag}", ci=ci, co=co, H=H, out=rb(x, None))
    dn = model.Downsample(128, True)
    fill(dn, "vdown.", seed)
    x = synth.synthetic_input("vdown.x", (2, 128, 16, 16), seed)
    save("op_vae_downsample_c128", out=dn(x))
    for tag, use_mask in {"nomask": False, "mask": True}.items():
        ab = model.AttnBlock(128)
        fill(ab, "vattn.", seed)
        x = synth.synthetic_input("vattn.x", (2, 128, 16, 16), seed)
        mask = None
        if use_mask:
            mask = {"fg_mask": ellipse_mask(2, 128, 128), "aug_mask": border_mask(2, 128, 128, 17)}
        save(f"op_vae_attnblock_{tag}", out=ab(x, mask))

    # ---------------- whole model: narrow UNet (mc=32, ctx 64), bs=2 -----------------------
    def run_unet(cfg, B, M, tag, iter_type="normal_recon", capture=True, use_mask=False,
                 with_grad=False, subsample=False):
        kw = dict(cfg)
        kw["attention_resolutions"] = list(kw["attention_resolutions"])
        kw["channel_mult"] = list(kw["channel_mult"])
        unet = openaimodel.UNetModel(**kw)
        fill(unet, "model.diffusion_model.", seed)
        x = synth.synthetic_input(f"unet.{tag}.x", (B, 4, 64, 64), seed)
        tt = torch.tensor([500, 37, 999, 3][:B])
        ntok = 2 * M if iter_type == "mix_hijk" else M
        ctx = synth.synthetic_input(f"unet.{tag}.ctx", (16 * B, ntok, cfg["context_dim"]), seed)
        img_mask = border_mask(B, 64, 64, 6) if use_mask else None
        extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1,
                 "iter_type": iter_type, "is_training": True, "capture_distill_attn": capture,
                 "placeholder2indices": None, "img_mask": img_mask}
        out = {}
        if with_grad:
            with torch.enable_grad():
                ctx_g = ctx.clone().requires_grad_(True)
                eps = unet(x, tt, context=ctx_g, context_in=None, extra_info=extra)
                w = synth.synthetic_input(f"unet.{tag}.gw", eps.shape, seed)
                (eps * w).sum().backward()
                g = ctx_g.grad
                out["grad_context_norm"] = g.norm()
                out["grad_context"] = g[:, ::4, ::8] if subsample else g
                eps = eps.detach()
        else:
            eps = unet(x, tt, context=ctx, context_in=None, extra_info=extra)
        out["eps"] = eps
        acts = extra["ca_layers_activations"]
        for key in ("outfeat", "attn", "attnscore", "q"):
            for li, ten in acts[key].items():
                ten = ten.detach()
                # strided subsamples keep the fixtures small; the test applies the same slices
                ten = subsample_act(key, ten)
                out[f"{key}_{li}"] = ten
        save(f"unet_{tag}", B=B, M=M, iter_type=iter_type, capture=capture, use_mask=use_mask,
             t=tt, **out)
        del unet

    narrow = dict(synth.SD15_UNET, model_channels=64, context_dim=128)
    run_unet(narrow, 2, 77, "narrow_recon", with_grad=True)
    run_unet(narrow, 2, 77, "narrow_mask", use_mask=True, capture=False)
    run_unet(narrow, 2, 77, "narrow_mixhijk", iter_type="mix_hijk")

    # ---------------- whole model: narrow VAE encoder (ch=32, 64x64 input) -----------------
    def run_vae(dd, B, res, tag, use_mask):
        enc = model.Encoder(**{**dd, "ch_mult": list(dd["ch_mult"]), "attn_resolutions": []})
        fill(enc, "first_stage_model.encoder.", seed)
        qc = torch.nn.Conv2d(2 * dd["z_channels"], 2 * 4, 1)
        fill(qc, "first_stage_model.quant_conv.", seed)
        x = synth.synthetic_input(f"vae.{tag}.x", (B, 3, res, res), seed, 0.5).clamp(-1, 1)
        mask = None
        if use_mask:
            mask = {"fg_mask": ellipse_mask(B, res, res), "aug_mask": border_mask(B, res, res, res // 16)}
        h = enc(x, mask)
        moments = qc(h)
        post = distributions.DiagonalGaussianDistribution(moments)
        save(f"vae_{tag}", B=B, res=res, use_mask=use_mask, moments=moments,
             mean=post.mean, std=post.std)

    vnarrow = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    run_vae(vnarrow, 2, 64, "narrow_nomask", False)
    run_vae(vnarrow, 2, 64, "narrow_mask", True)

    # ---------------- optimiser: Prodigy.step + grad-norm clip + LR schedule (ldm/prodigy.py, ldm/util.py:26-41) ------
    run_prodigy()
    run_anneal()
    run_regs()
    run_regs_masks()
    run_cond_helpers()
    run_decoder_and_ddim(model, util, args.full)

    if args.full:
        run_unet(dict(synth.SD15_UNET), 1, 77, "sd15_recon", with_grad=True, subsample=True)
        run_vae(dict(synth.SD15_VAE_DD), 1, 512, "sd15_mask", True)


if __name__ == "__main__":
    main()

"""Whole-model parity on the MI355X: the HIP UNet / VAE encoder / training step against
(a) the golden vectors captured from the reference's own modules and (b) the oracle run on the
same inputs.  Tolerances: matrix-core operands are bf16 (fp32 accumulate, fp32 residual stream,
fp32 norm / softmax statistics); the north-star bar is <= 1e-3 relative on the eps-prediction
MSE loss, per-tensor errors are reported and bounded at the bf16 level."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from adaprompt_amd import ops, synth
from conftest import load_golden, rel_err, ellipse_mask, border_mask, subsample_act

pytestmark = pytest.mark.gpu

EPS_TOL = 1.5e-2       # relative L2 of eps-hat (bf16 operands through ~60 contractions): measured 8.7e-3 full size, 9.2e-3 narrow
GRAD_CTX_TOL = 2.5e-2  # d loss / d context: measured 1.42e-2 full size, 1.69e-2 narrow (tests/test_precision_gpu.py holds 2e-2)
LOSS_TOL = 1e-3        # north_star: eps-pred MSE, relative


def dev():
    return torch.device("cuda:0")


def build_unet(cfg, train=False):
    """``train=False``: the shipped config (`unfreeze_model: False`, yaml:26) -- gradients go to the context only."""
    from adaprompt_amd.ldm.util import instantiate_from_config
    m = instantiate_from_config({"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(cfg)})
    P = "model.diffusion_model."          # tensors are seeded by their full checkpoint name
    sd = {k[len(P):]: v for k, v in synth.synthetic_unet_state_dict(cfg, prefix=P).items()}
    m.load_state_dict(sd, strict=True)
    m.requires_grad_(train)
    return m.to(dev()).eval()


def mse_loss_vs_noise(eps, seed_tag):
    """the parity scalar: masked MSE of eps-hat against a fixed synthetic noise target (calc_recon_loss)."""
    from oracle import ldm_oracle as O
    noise = synth.synthetic_input(seed_tag + ".noise", eps.shape)
    B = eps.shape[0]
    fg = ellipse_mask(B, 64, 64)
    im = border_mask(B, 64, 64, 4)
    loss, _ = O.calc_recon_loss(eps.detach().float().cpu(), noise, im, fg, 1.0, 0.1)
    return float(loss)


def run_unet_case(cfg, tag, g, with_grad, subs_grad=False):
    B, M = g["B"], g["M"]
    unet = build_unet(cfg)
    x = synth.synthetic_input(f"unet.{tag}.x", (B, 4, 64, 64)).to(dev())
    ntok = 2 * M if g["iter_type"] == "mix_hijk" else M
    ctx = synth.synthetic_input(f"unet.{tag}.ctx", (16 * B, ntok, cfg["context_dim"])).to(dev())
    img_mask = border_mask(B, 64, 64, 6).to(dev()) if g["use_mask"] else None
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": g["iter_type"],
             "is_training": True, "capture_distill_attn": bool(g["capture"]), "placeholder2indices": None,
             "img_mask": img_mask}
    if with_grad:
        ctx = ctx.clone().requires_grad_(True)
    eps = unet(x, g["t"].to(dev()), context=ctx, context_in=None, extra_info=extra)
    assert eps.shape == g["eps"].shape
    e = rel_err(eps.cpu(), g["eps"])
    l_hip, l_ref = mse_loss_vs_noise(eps, tag), mse_loss_vs_noise(g["eps"], tag)
    print(f"[{tag}] eps rel L2 {e:.3e}   loss hip {l_hip:.6f} ref {l_ref:.6f} rel {abs(l_hip - l_ref) / l_ref:.2e}")
    assert e < EPS_TOL
    assert abs(l_hip - l_ref) / l_ref < LOSS_TOL
    acts = extra["ca_layers_activations"]
    n = 0
    for key in ("outfeat", "attn", "attnscore", "q"):
        for li, ten in acts[key].items():
            got = subsample_act(key, ten.detach().float().cpu())
            ea = rel_err(got, g[f"{key}_{li}"])
            assert ea < 4e-2, (key, li, ea)
            n += 1
    if g["capture"]:
        assert n == 48
    if with_grad:
        w = synth.synthetic_input(f"unet.{tag}.gw", tuple(eps.shape)).to(dev())
        (eps * w).sum().backward()
        gr = ctx.grad.cpu()
        ref = g["grad_context"]
        got = gr[:, ::4, ::8] if subs_grad else gr
        eg = rel_err(got, ref)
        print(f"[{tag}] grad_context rel L2 {eg:.3e}  norm hip {float(gr.norm()):.5f} ref {float(g['grad_context_norm']):.5f}")
        assert eg < GRAD_CTX_TOL
        assert abs(float(gr.norm()) / float(g["grad_context_norm"]) - 1) < 2e-2


NARROW = dict(synth.SD15_UNET, model_channels=64, context_dim=128)


def test_unet_narrow_recon_and_grad():
    run_unet_case(NARROW, "narrow_recon", load_golden("unet_narrow_recon"), True)


def test_unet_narrow_mask():
    run_unet_case(NARROW, "narrow_mask", load_golden("unet_narrow_mask"), False)


@pytest.mark.parametrize("c_call", [True, False])
def test_context_kv_hoist_equals_the_per_layer_projections(c_call):
    """``functional.ContextKVFn``: the 16 layers' cross-attention K | V projections (attention.py:195-213) as 5 batched launches
    in front of the UNet -- and their context gradients as 5 behind its backward -- against the same projections made inside
    the blocks, one launch per layer each way (``HF.HOIST_KV`` off).  Same operands, same bf16 rounding of the context; the
    batched launch goes to another contraction kernel (bf16 operand ring kernel instead of the f32-gather kernel), so the f32
    accumulation order differs: K | V agree to a bf16 ulp, eps / the token maps / the context gradient to round-off."""
    from adaprompt_amd import functional as HF
    B, M = 2, 77
    unet = build_unet(NARROW)
    x = synth.synthetic_input("kvh.x", (B, 4, 64, 64)).to(dev())
    t = torch.tensor([420, 640]).to(dev())
    ctx0 = synth.synthetic_input("kvh.ctx", (16 * B, M, NARROW["context_dim"])).to(dev())
    w = synth.synthetic_input("kvh.gw", (B, 4, 64, 64)).to(dev())
    subj = (torch.arange(B, device=dev()).repeat_interleave(16), torch.arange(4, 20, device=dev()).repeat(B))

    def run(hoist):
        was, was_c = HF.HOIST_KV, HF.STBLOCK_C
        HF.HOIST_KV, HF.STBLOCK_C = hoist, c_call
        try:
            ctx = ctx0.clone().requires_grad_(True)
            extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon", "is_training": True,
                     "capture_distill_attn": True, "placeholder2indices": None, "img_mask": border_mask(B, 64, 64, 6).to(dev()),
                     "subj_indices": subj, "capture_token_maps_only": True}
            eps = unet(x, t, context=ctx, context_in=None, extra_info=extra)
            tms = extra["ca_layers_activations"]["attnscore_tokmap"]
            roots = [eps] + [tms[li] for li in sorted(tms)]
            grads = [w] + [torch.full_like(tms[li], 1e-3) for li in sorted(tms)]
            torch.autograd.backward(roots, grads)
            torch.cuda.synchronize()
            return eps.detach().clone(), ctx.grad.detach().clone(), {li: v.detach().clone() for li, v in tms.items()}
        finally:
            HF.HOIST_KV, HF.STBLOCK_C = was, was_c

    n0 = list(HF.STB_CALLS)
    e1, g1, tm1 = run(True)
    if c_call:
        assert HF.STB_CALLS[0] == n0[0] + 16 and HF.STB_CALLS[1] == n0[1] + 16, "the C path was not taken"
    e0, g0, tm0 = run(False)
    e1b, g1b, _ = run(True)
    assert torch.equal(e1, e1b) and torch.equal(g1, g1b)                 # reproducible
    assert rel_err(e1, e0) < 2e-3, rel_err(e1, e0)
    assert rel_err(g1, g0) < 5e-3, rel_err(g1, g0)
    for li in tm0:
        assert rel_err(tm1[li], tm0[li]) < 2e-3, (li, rel_err(tm1[li], tm0[li]))
    # the projections themselves: batched launch against one ops.linear per layer
    runs = unet._kv_runs(NARROW["context_dim"]) if HF.MODEL_STAMP is not None else None
    HF.MODEL_STAMP = (id(unet),) + HF.model_stamp(list(unet.parameters()))
    try:
        runs = unet._kv_runs(NARROW["context_dim"])
        assert [(l0, n) for l0, n, *_ in runs] == [(0, 2), (2, 2), (4, 6), (10, 3), (13, 3)]
        ctx_l = ctx0.reshape(B, 16, M, -1).permute(1, 0, 2, 3).contiguous()
        kvs = HF.ContextKVFn.apply(ctx_l, runs)
        for ca, stm in enumerate(unet._ca_stms):
            a2 = stm.transformer_blocks[0].attn2
            pk = stm._wc.get("kv2", [a2.to_k.weight, a2.to_v.weight])
            _, ref = ops.linear(ctx_l[ca], pk.fwd, pk.O, out_f32=False, out_bf16=True)
            d = (kvs[ca].float() - ref.float()).abs()
            assert float(d.max()) <= float(ref.float().abs().max()) * 2 ** -7, (ca, float(d.max()))
            assert float((d > 0).float().mean()) < 0.02, (ca, float((d > 0).float().mean()))
    finally:
        HF.MODEL_STAMP = None


def test_unet_narrow_mixhijk():
    run_unet_case(NARROW, "narrow_mixhijk", load_golden("unet_narrow_mixhijk"), False)


def test_unet_sd15_full_size_vs_reference_golden():
    """the 859.5 M-parameter SD-1.5 UNet on the output of the reference's UNetModel (config 1) + grad wrt context"""
    run_unet_case(dict(synth.SD15_UNET), "sd15_recon", load_golden("unet_sd15_recon"), True, subs_grad=True)


def test_unet_narrow_weight_gradients_vs_oracle():
    """`unfreeze_model: True` (ddpm.py:775-786): every one of the UNet's parameters receives its gradient from the HIP
    backward (conv / Linear dW and bias, GroupNorm / LayerNorm affine, the time-embedding MLP), compared tensor by tensor
    with autograd through the fp32 oracle fed the same upstream gradient; the context gradient stays that of the frozen
    run (the data-gradient path is untouched)."""
    from oracle import ldm_oracle as O
    cfg = dict(NARROW)
    B = 2
    P = "model.diffusion_model."
    usd = synth.synthetic_unet_state_dict(cfg, prefix=P)
    x = synth.synthetic_input("unet.wg.x", (B, 4, 64, 64))
    t = torch.tensor([120, 870])
    ctx = synth.synthetic_input("unet.wg.ctx", (16 * B, 77, cfg["context_dim"]))
    im = border_mask(B, 64, 64, 6)
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": False, "placeholder2indices": None, "img_mask": im}
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in usd.items()}
    ctx_ref = ctx.clone().requires_grad_(True)
    eps_ref = O.unet_forward(sd_ref, cfg, x, t, ctx_ref, dict(extra), prefix=P)
    g_eps = synth.synthetic_input("unet.wg.gw", tuple(eps_ref.shape))
    eps_ref.backward(g_eps)

    def run(train):
        unet = build_unet(cfg, train=train)
        c = ctx.to(dev()).clone().requires_grad_(True)
        e = dict(extra, img_mask=im.to(dev()))
        eps = unet(x.to(dev()), t.to(dev()), context=c, context_in=None, extra_info=e)
        eps.backward(g_eps.to(dev()))
        return unet, c.grad

    unet, gctx = run(True)
    _, gctx_frozen = run(False)
    # Not bit-identical: the training run computes the time-embedding MLP under torch autograd (rocBLAS) instead of the
    # few-row HIP kernel; last-bit differences there flip bf16 roundings of the first activations and the two runs end up
    # with independent rounding noise (measured: eps differs by 8e-3 between them, each is 9e-3 from the oracle).  What
    # must hold is that the training run is as close to the oracle as the frozen one.
    e_tr, e_fr = rel_err(gctx.cpu(), ctx_ref.grad), rel_err(gctx_frozen.cpu(), ctx_ref.grad)
    print(f"[unet weight grads] context gradient vs oracle: training run {e_tr:.2e}, frozen run {e_fr:.2e}, "
          f"between them {rel_err(gctx.cpu(), gctx_frozen.cpu()):.2e}")
    assert e_tr < 1.25 * e_fr + 1e-3
    assert rel_err(gctx.cpu(), ctx_ref.grad) < GRAD_CTX_TOL
    num = den = 0.0
    worst = (0.0, None)
    n = 0
    for name, p in unet.named_parameters():
        ref = sd_ref[P + name].grad
        assert ref is not None, name
        assert p.grad is not None, f"no gradient reached {name}"
        got = p.grad.detach().float().cpu().reshape(ref.shape)
        e = rel_err(got, ref)
        if e > worst[0]:
            worst = (e, name)
        num += float((got.double() - ref.double()).pow(2).sum())
        den += float(ref.double().pow(2).sum())
        assert e < 8e-2, (name, e, float(ref.norm()))
        n += 1
    tot = (num / den) ** 0.5
    print(f"[unet weight grads] {n} tensors, global rel L2 {tot:.3e}, worst {worst[1]} {worst[0]:.3e}")
    assert n == len(sd_ref) and tot < 3e-2


def test_unfrozen_unet_optimizer_step_rebuilds_weight_packs():
    """after Prodigy has moved the UNet's parameters (raw-pointer kernels on its flat buffer) the next forward must use
    the new weights: the bf16 packs are keyed on Tensor._version, which the step bumps.  A fresh module loaded with
    the stepped state dict must give the bit-identical output."""
    from adaprompt_amd.ldm.prodigy import Prodigy
    cfg = dict(NARROW)
    B = 1
    unet = build_unet(cfg, train=True)
    x = synth.synthetic_input("unet.st.x", (B, 4, 64, 64)).to(dev())
    t = torch.tensor([400]).to(dev())
    ctx = synth.synthetic_input("unet.st.ctx", (16 * B, 77, cfg["context_dim"])).to(dev())
    g = synth.synthetic_input("unet.st.g", (B, 4, 64, 64)).to(dev())
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": False, "placeholder2indices": None, "img_mask": None}
    opt = Prodigy(list(unet.parameters()), lr=1.0, d0=1e-3)
    opt.grad_buffer                                   # parameters and gradients move into the flat buffers
    v0 = unet.out[2].weight._version
    eps0 = unet(x, t, context=ctx, context_in=None, extra_info=dict(extra))
    eps0.backward(g)
    assert float(opt.grad_buffer.abs().sum()) > 0     # the block backward wrote into the optimiser's flat buffer
    opt.step()
    opt.zero_grad()
    assert unet.out[2].weight._version > v0
    eps1 = unet(x, t, context=ctx, context_in=None, extra_info=dict(extra))
    assert rel_err(eps1.detach().cpu(), eps0.detach().cpu()) > 1e-3          # the weights moved
    fresh = build_unet(cfg, train=True)
    fresh.load_state_dict(unet.state_dict())
    eps2 = fresh(x, t, context=ctx, context_in=None, extra_info=dict(extra))
    assert torch.equal(eps1, eps2)


def build_vae(dd):
    from adaprompt_amd.ldm.models.autoencoder import AutoencoderKL
    m = AutoencoderKL(dict(dd), None, 4)
    P = "first_stage_model."
    sd = {k[len(P):]: v for k, v in synth.synthetic_vae_state_dict(dd, prefix=P).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("post_quant_conv") for k in missing)
    return m.to(dev()).eval()


def run_vae_case(dd, tag, g, tol):
    B, res = g["B"], g["res"]
    vae = build_vae(dd)
    x = synth.synthetic_input(f"vae.{tag}.x", (B, 3, res, res), 0, 0.5).clamp(-1, 1).to(dev())
    mask = None
    if g["use_mask"]:
        mask = {"fg_mask": ellipse_mask(B, res, res).to(dev()), "aug_mask": border_mask(B, res, res, res // 16).to(dev())}
    post = vae.encode(x, mask)
    e = rel_err(post.parameters.cpu(), g["moments"])
    em = rel_err(post.mean.cpu(), g["mean"])
    print(f"[vae {tag}] moments rel L2 {e:.3e}  mean {em:.3e}")
    assert e < tol and em < tol
    # the pixel-major fast path agrees with the NCHW boundary path
    m2 = vae.encode_moments_nhwc(x.permute(0, 2, 3, 1), mask).permute(0, 3, 1, 2)
    assert torch.equal(m2, post.parameters)


@pytest.mark.parametrize("tag", ["narrow_nomask", "narrow_mask"])
def test_vae_narrow(tag):
    run_vae_case(dict(synth.SD15_VAE_DD, ch=32, resolution=64), tag, load_golden("vae_" + tag), 2e-2)


def test_vae_sd15_full_size_vs_reference_golden():
    run_vae_case(dict(synth.SD15_VAE_DD), "sd15_mask", load_golden("vae_sd15_mask"), 2e-2)


def test_training_step_matches_oracle():
    """one pure-recon micro-batch through LatentDiffusion.training_step vs oracle.recon_step, narrow widths"""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ldm_oracle as O
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    usd = synth.synthetic_unet_state_dict(ucfg)
    vsd = synth.synthetic_vae_state_dict(vdd)
    missing, unexpected = ld.load_state_dict({**usd, **vsd}, strict=False)
    assert not unexpected
    ld = ld.to(dev())
    ld.freeze_unet()
    B, res = 2, 64
    # the narrow VAE maps 64x64 -> 8x8 latents; the UNet needs 64x64, so drive the UNet with its own latent here:
    # parity of the step is checked stage by stage
    img = synth.synthetic_input("step.img", (B, res, res, 3), 0, 0.5).clamp(-1, 1)
    fgm = ellipse_mask(B, res, res)[:, 0]
    augm = border_mask(B, res, res, 4)[:, 0]
    post_noise = synth.synthetic_input("step.pn", (B, 4, 8, 8))
    batch = {"image": img.to(dev()), "fg_mask": fgm.to(dev()), "aug_mask": augm.to(dev())}
    z, _ = ld.get_input(batch, post_noise.to(dev()))
    with torch.no_grad():
        mom = O.autoencoder_encode_moments(vsd, vdd, img.permute(0, 3, 1, 2),
                                           {"fg_mask": fgm[:, None], "aug_mask": augm[:, None]})
        z_ref = O.get_first_stage_encoding(mom, post_noise)
    assert rel_err(z.cpu(), z_ref) < 2e-2
    # UNet part of the step at 64x64 latents
    x0 = synth.synthetic_input("step.x0", (B, 4, 64, 64))
    noise = synth.synthetic_input("step.noise", (B, 4, 64, 64))
    t = torch.tensor([200, 900])
    ctx = synth.synthetic_input("step.ctx", (16 * B, 77, ucfg["context_dim"]))
    fg64, im64 = ellipse_mask(B, 64, 64), border_mask(B, 64, 64, 5)
    sched = O.make_schedule()
    ctx_ref = ctx.clone().requires_grad_(True)
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": True, "img_mask": im64}
    eps_ref = O.unet_forward(usd, ucfg, O.q_sample(sched, x0, t, noise), t, ctx_ref, extra)
    loss_ref, _ = O.calc_recon_loss(eps_ref, noise, im64, fg64, 1.0, 0.1)
    (g_ref,) = torch.autograd.grad(loss_ref, ctx_ref)
    ctx_hip = ctx.to(dev()).clone().requires_grad_(True)
    extra_h = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
               "is_training": True, "capture_distill_attn": True, "img_mask": im64.to(dev())}
    eps_hip, x_noisy = ld.guided_denoise(x0.to(dev()), noise.to(dev()), t.to(dev()), (ctx_hip, None, extra_h))
    loss_hip, grad = ld.calc_recon_loss(eps_hip, noise.to(dev()), im64.to(dev()), fg64.to(dev()), 1.0, 0.1)
    eps_hip.backward(grad)
    print(f"[step] loss hip {float(loss_hip):.6f} ref {float(loss_ref):.6f}  grad rel {rel_err(ctx_hip.grad.cpu(), g_ref):.3e}")
    assert abs(float(loss_hip) - float(loss_ref)) / float(loss_ref) < LOSS_TOL
    assert rel_err(ctx_hip.grad.cpu(), g_ref) < 2e-2                      # measured 9.1e-3


def test_recon_step_with_regularizers_matches_oracle():
    """a9: the do_normal_recon iteration = masked MSE + fg/bg complementary loss with its three mask hinges
    (ddpm.py:3461-3500) + cross-layer consistency of the subject / background attention maps (both with the gradient
    THROUGH the captured attnscore, via the token maps) + prompt-delta loss on the four-way static embeddings
    (ddpm.py:3207-3270).  The regulariser weights are raised from 5e-5 / 2e-4 to O(1) so that their gradients
    are not lost beside the MSE gradient; loss parts and d loss / d context against the oracle."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ldm_oracle as O
    from oracle import regs_oracle as R
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
                         fg_bg_xlayer_consist_loss_weight=2.0, prompt_emb_delta_reg_weight=3.0,
                         fg_bg_complementary_loss_weight=1.5)
    usd = synth.synthetic_unet_state_dict(ucfg)
    missing, unexpected = ld.load_state_dict({**usd, **synth.synthetic_vae_state_dict(vdd)}, strict=False)
    assert not unexpected
    ld = ld.to(dev())
    ld.freeze_unet()
    B, D = 2, ucfg["context_dim"]
    x0 = synth.synthetic_input("reg.x0", (B, 4, 64, 64))
    noise = synth.synthetic_input("reg.noise", (B, 4, 64, 64))
    t = torch.tensor([300, 750])
    emb4 = synth.synthetic_input("reg.emb4", (4 * B, 16, 77, D))            # subj single | subj comp | cls single | cls comp
    pmask = torch.full((4 * B, 77, 1), 0.5)
    for blk, n_tok in enumerate((21, 31, 21, 31)):
        pmask[blk * B:(blk + 1) * B, :n_tok] = 1.0
    inst = torch.arange(B)
    subj = (inst.repeat_interleave(16), torch.arange(4, 20).repeat(B))
    bgi = (inst.repeat_interleave(4), torch.arange(24, 28).repeat(B))
    fg64, im64 = ellipse_mask(B, 64, 64), border_mask(B, 64, 64, 5)
    base = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
            "is_training": True, "capture_distill_attn": True}

    # ---- oracle: the same three terms with the reference's scales (zero-shot + Prodigy: 0.1 | 0.2 / 0.06)
    e_ref = emb4.clone().requires_grad_(True)
    ctx_ref = e_ref[:B].reshape(16 * B, 77, D)                              # the UNet sees the subject-single block
    sched = O.make_schedule()
    ex_ref = dict(base, img_mask=im64)
    eps_ref = O.unet_forward(usd, ucfg, O.q_sample(sched, x0, t, noise), t, ctx_ref, ex_ref)
    mse_ref, _ = O.calc_recon_loss(eps_ref, noise, im64, fg64, 1.0, 0.1)
    fg_ref, bg_ref = R.calc_fg_bg_xlayer_consist_loss(ex_ref["ca_layers_activations"]["attnscore"], subj, bgi, B)
    pd_ref = R.calc_prompt_emb_delta_loss(e_ref, pmask.clone())
    cm_ref = R.calc_fg_bg_complementary_loss(ex_ref["ca_layers_activations"]["attnscore"], subj, bgi, B, fg_grad_scale=0.1,
                                             fg_mask=fg64)
    total_ref = mse_ref + (fg_ref * 0.2 + bg_ref * 0.06) * 2.0 + pd_ref * 3.0 * 0.1 \
        + (cm_ref[0] * 0.2 + cm_ref[1] + cm_ref[2] + cm_ref[3]) * 1.5
    total_ref.backward()

    # ---- HIP
    e_hip = emb4.to(dev()).clone().requires_grad_(True)
    ctx_hip = e_hip[:B].reshape(16 * B, 77, D)
    ex_hip = dict(base, subj_indices=tuple(i.to(dev()) for i in subj), bg_indices=tuple(i.to(dev()) for i in bgi),
                  c_static_emb_4b=e_hip, prompt_emb_mask=pmask.to(dev()).clone())
    batch = {"fg_mask": fg64[:, 0].to(dev()), "aug_mask": im64[:, 0].to(dev())}
    loss, grad, out, aux = ld.shared_step(batch, t=t.to(dev()), noise=noise.to(dev()), cond=(ctx_hip, None, ex_hip),
                                          x_start=x0.to(dev()))
    ld.manual_backward(out, grad, aux)
    parts = aux["reg_parts"]
    print(f"[recon+regs] total hip {float(loss):.6f} ref {float(total_ref.detach()):.6f} | fg {float(parts['fg_xlayer_consist']):.5f} "
          f"/ {float(fg_ref):.5f}  bg {float(parts['bg_xlayer_consist']):.5f} / {float(bg_ref):.5f}  "
          f"delta {float(parts['static_prompt_delta']):.5f} / {float(pd_ref):.5f}")
    assert abs(float(parts["fg_xlayer_consist"]) - float(fg_ref)) < 2e-2 * abs(float(fg_ref))
    assert abs(float(parts["bg_xlayer_consist"]) - float(bg_ref)) < 2e-2 * abs(float(bg_ref))
    assert abs(float(parts["static_prompt_delta"]) - float(pd_ref)) < 1e-4 * abs(float(pd_ref))
    for name, ref in zip(("fg_bg_complem", "subj_mb_suppress", "bg_mf_suppress", "fg_bg_mask_contrast"), cm_ref):
        print(f"[recon+regs] {name} {float(parts[name]):.6f} / {float(ref):.6f}")
        assert abs(float(parts[name]) - float(ref)) < 3e-2 * abs(float(ref)) + 1e-5, name
    assert abs(float(loss) - float(total_ref.detach())) < 5e-3 * float(total_ref.detach())
    ge = rel_err(e_hip.grad.cpu(), e_ref.grad)
    # the regularisers' own share of the gradient: remove the MSE part (computed without them on both sides)
    e_ref2 = emb4.clone().requires_grad_(True)
    ex2 = dict(base, img_mask=im64)
    eps2 = O.unet_forward(usd, ucfg, O.q_sample(sched, x0, t, noise), t, e_ref2[:B].reshape(16 * B, 77, D), ex2)
    O.calc_recon_loss(eps2, noise, im64, fg64, 1.0, 0.1)[0].backward()
    reg_share = float((e_ref.grad - e_ref2.grad).norm() / e_ref.grad.norm())
    print(f"[recon+regs] d loss / d embeddings rel {ge:.3e}; the regularisers carry {reg_share:.2f} of its norm")
    assert reg_share > 0.3          # otherwise this test would not see them
    assert ge < GRAD_CTX_TOL        # measured 1.37e-2


def test_probably_anneal_t_device_path_has_the_reference_distribution():
    """the sync-free device formulation of probably_anneal_t (util.py:1508-1530): every redraw lies in
    [int(t*lb), int(t*ub)] clipped to the schedule, the redraw is uniform over that range, and t is kept with the
    annealed probability."""
    import random
    from adaprompt_amd.ldm.util import probably_anneal_t
    random.seed(0)
    torch.manual_seed(0)
    t = torch.tensor([0, 3, 250, 640, 769, 999], device=dev())
    draws, kept = [], 0
    for _ in range(400):
        out = probably_anneal_t(t, 0.5, 1000, ratio_range=(1, 1.3), keep_prob_range=(0.4, 0.2))
        if torch.equal(out, t):
            kept += 1
        else:
            draws.append(out.cpu())
    assert 0.18 < kept / 400 < 0.42                         # keep probability 0.3 at training_percent 0.5 (+ chance ties)
    d = torch.stack(draws).double()
    lo = torch.tensor([0, 3, 250, 640, 769, 999.0])
    hi = torch.tensor([0, 3, 325, 832, 999, 999.0])         # min(int(t*1.3) + 1, 1000) - 1
    assert bool((d >= lo).all()) and bool((d <= hi).all())
    assert bool((d.max(0).values >= hi - 2).all()) and bool((d.min(0).values <= lo + 2).all())
    mid = (lo + hi) / 2
    assert bool(((d.mean(0) - mid).abs() <= 0.12 * (hi - lo) + 1e-9).all())


def test_training_loop_prodigy_two_optimizer_steps_vs_oracle():
    """a1 end to end: four micro-batches through LatentDiffusion.training_step with the flat-buffer Prodigy, the
    fused 0.5 clip, the GradReducer on the optimiser's buffer and the LR schedule -- gradients accumulate over two
    micro-batches, then clip -> step -> zero -> scheduler (ddpm.py:583-633).  The oracle side runs the same loop
    with the fp32 UNet restatement and ProdigyOracle.  Adam-normalised updates amplify the ~1e-2 bf16 gradient
    error on near-zero components, so the parameter DELTAS are compared at 10 % and d at 10 %: plumbing mistakes
    (missed accumulation, unclipped or doubly-stepped gradients, stale buffers) are O(1) errors."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaprompt_amd.ldm.prodigy import Prodigy
    from adaprompt_amd.ldm.util import prodigy_linear_schedule
    from adaprompt_amd.parallel import GradReducer
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    from oracle import ldm_oracle as O
    from oracle.prodigy_oracle import ProdigyOracle, clip_grad_norm, linear_schedule_lrs
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    torch.manual_seed(3)
    hook_ref = SyntheticSubjBasisGenerator(n_params=3 * 16 * 77 * 128, tokens=77, dim=128, id_dim=32)
    with torch.no_grad():
        hook_ref.bases.mul_(20.0)                       # context of unit scale so that the loss depends on it
    hook = SyntheticSubjBasisGenerator(n_params=3 * 16 * 77 * 128, tokens=77, dim=128, id_dim=32)
    hook.load_state_dict(hook_ref.state_dict())
    hook = hook.to(dev())
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
                         cond_fn=make_cond_fn(hook, capture=False))
    usd = synth.synthetic_unet_state_dict(ucfg)
    ld.load_state_dict(usd, strict=False)
    ld = ld.to(dev())
    ld.freeze_unet()
    kw = dict(betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
    params = list(hook.parameters())
    init = [p.detach().cpu().clone() for p in params]
    opt = Prodigy(params, lr=1.0, **kw)
    red = GradReducer(params, flat=opt.grad_buffer)
    sched = prodigy_linear_schedule(opt, max_steps=4, warm_up_steps=1, scheduler_cycles=1)
    B = 2
    fg64, im64 = ellipse_mask(B, 64, 64), border_mask(B, 64, 64, 5)
    sch = O.make_schedule()
    ref_params = list(hook_ref.parameters())
    orc = ProdigyOracle([p.data for p in ref_params], lr=1.0, **kw)
    lrs = linear_schedule_lrs(1.0, max_steps=4, warm_up_steps=1, scheduler_cycles=1, n=2)
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": False, "img_mask": im64}
    losses = []
    for mb in range(4):
        x0 = synth.synthetic_input(f"loop.x0.{mb}", (B, 4, 64, 64))
        noise = synth.synthetic_input(f"loop.noise.{mb}", (B, 4, 64, 64))
        ids = synth.synthetic_input(f"loop.ids.{mb}", (B, 32))
        t = torch.tensor([150 + 200 * mb, 900 - 100 * mb])
        batch = {"zs_id_embs": ids.to(dev()), "fg_mask": fg64[:, 0].to(dev()), "aug_mask": im64[:, 0].to(dev())}
        loss, _aux = ld.training_step(batch, optimizer=opt, reducer=red, scheduler=sched, t=t.to(dev()),
                                      noise=noise.to(dev()), x_start=x0.to(dev()))
        # ---- oracle side of the same micro-batch
        ctx = hook_ref(ids)
        eps = O.unet_forward(usd, ucfg, O.q_sample(sch, x0, t, noise), t, ctx, extra)
        loss_ref, _ = O.calc_recon_loss(eps, noise, im64, fg64, 1.0, 0.1)
        loss_ref.backward()                                   # accumulates into .grad, as manual_backward does
        losses.append((float(loss), float(loss_ref)))
        if (mb + 1) % 2 == 0:
            grads = [p.grad for p in ref_params]
            clip_grad_norm(grads, 0.5)
            orc.lr = lrs[(mb + 1) // 2 - 1]
            orc.step(grads)
            for p in ref_params:
                p.grad = None
    for a, b in losses:
        assert abs(a - b) / b < 5e-3, losses
    ds = opt.device_state()
    assert ds["k"] == orc.k == 2
    assert abs(ds["d"] - orc.d) / orc.d < 0.1, (ds["d"], orc.d)
    assert float(opt.grad_buffer.abs().max()) == 0.0          # zeroed after the step
    assert abs(opt.param_groups[0]["lr"] - linear_schedule_lrs(1.0, 4, 1, 1, n=3)[2]) < 1e-12
    for p, p_ref, p0 in zip(params, ref_params, init):
        d_hip, d_ref = p.detach().cpu() - p0, p_ref.detach() - p0
        assert float(d_ref.abs().max()) > 0
        assert rel_err(d_hip, d_ref) < 0.1, rel_err(d_hip, d_ref)


def test_recon_micro_batch_bs4_sd15_vs_oracle():
    """The HEADLINE workload's shape against the oracle (missing #3): one config-2 recon micro-batch at bs 4, 512 x 512,
    full SD-1.5 sizes (859.5 M UNet + 34 M VAE encoder) -- VAE encode with the fg / aug masks -> posterior sample -> q_sample ->
    UNet with the 16-way layerwise context and the image mask on the self-attention keys -> masked MSE -> d loss / d context
    (reference: ddpm.py:1178-1256, 2483-2532, 2841-3039, 3571-3595; openaimodel.py:827-1052), then the same micro-batch with the
    iteration's attention regularisers against the oracle's values on its own captured attnscore.
    Gates: loss <= 1e-3 relative (north_star), d loss / d context <= 2.5e-2 relative L2, every regulariser <= 5 %."""
    from adaprompt_amd import hostinfo
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ldm_oracle as O
    from oracle import regs_oracle as R
    hostinfo.limit_torch_threads()
    ucfg, vdd = dict(synth.SD15_UNET), dict(synth.SD15_VAE_DD)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                                  {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    usd, vsd = synth.synthetic_unet_state_dict(ucfg), synth.synthetic_vae_state_dict(vdd)
    _, unexpected = ld.load_state_dict({**usd, **vsd}, strict=False)
    assert not unexpected
    ld = ld.to(dev())
    ld.freeze_unet()
    B = 4
    img = synth.synthetic_input("bs4.img", (B, 512, 512, 3), 0, 0.5).clamp(-1, 1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    fg = torch.stack([((xx / (0.5 + 0.05 * b)) ** 2 + (yy / (0.7 - 0.04 * b)) ** 2 <= 1.0).float() for b in range(B)])
    aug = torch.zeros(B, 512, 512)
    for b, w in enumerate((0, 24, 48, 72)):                  # the dataloader's random-scale borders, 0 .. 76 px
        aug[b, w:512 - w, w:512 - w] = 1
    pn = synth.synthetic_input("bs4.pn", (B, 4, 64, 64))
    noise = synth.synthetic_input("bs4.noise", (B, 4, 64, 64))
    t = torch.tensor([37, 417, 702, 961])
    ctx = synth.synthetic_input("bs4.ctx", (16 * B, 77, 768))
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": True, "placeholder2indices": None}
    batch = {"image": img.to(dev()), "fg_mask": fg.to(dev()), "aug_mask": aug.to(dev())}
    # ---- HIP: the MSE-only micro-batch, then the one with the regularisers
    ctx_d = ctx.to(dev()).requires_grad_(True)
    loss, grad, out, aux = ld.shared_step(batch, t=t.to(dev()), noise=noise.to(dev()), post_noise=pn.to(dev()),
                                          cond=(ctx_d, None, dict(extra)))
    ld.manual_backward(out, grad, aux)
    subj = (torch.arange(B).repeat_interleave(16), torch.arange(4, 20).repeat(B))
    bgi = (torch.arange(B).repeat_interleave(4), torch.arange(24, 28).repeat(B))
    ctx_d2 = ctx.to(dev()).requires_grad_(True)
    extra2 = dict(extra, subj_indices=tuple(i.to(dev()) for i in subj), bg_indices=tuple(i.to(dev()) for i in bgi))
    _, grad2, out2, aux2 = ld.shared_step(batch, t=t.to(dev()), noise=noise.to(dev()), post_noise=pn.to(dev()),
                                          cond=(ctx_d2, None, extra2))
    ld.manual_backward(out2, grad2, aux2)
    torch.cuda.synchronize()
    # ---- oracle: one forward with the capture on, backward to the context
    fg64 = F.interpolate(fg[:, None], size=(64, 64), mode="nearest")
    im64 = F.interpolate(aug[:, None], size=(64, 64), mode="nearest")
    sched = O.make_schedule()
    with torch.no_grad():
        mom = O.autoencoder_encode_moments(vsd, vdd, img.permute(0, 3, 1, 2), {"fg_mask": fg[:, None], "aug_mask": aug[:, None]})
        z = O.get_first_stage_encoding(mom, pn)
        x_noisy = O.q_sample(sched, z, t, noise)
    ctx_r = ctx.clone().requires_grad_(True)
    ex_o = dict(extra, img_mask=im64)
    eps_ref = O.unet_forward(usd, ucfg, x_noisy, t, ctx_r, ex_o)
    loss_ref, _ = O.calc_recon_loss(eps_ref, noise, im64, fg64, 1.0, 0.1)
    (g_ref,) = torch.autograd.grad(loss_ref, ctx_r)
    lh, lr = float(loss), float(loss_ref)
    gerr = rel_err(ctx_d.grad.cpu(), g_ref)
    assert tuple(out.shape) == tuple(eps_ref.shape)
    eerr = rel_err(out.detach().cpu(), eps_ref.detach())
    print(f"[bs4 sd15] loss hip {lh:.6f} oracle {lr:.6f} rel {abs(lh - lr) / lr:.2e}; eps rel L2 {eerr:.2e}; "
          f"d/d context rel L2 {gerr:.2e}")
    assert abs(lh - lr) / lr < LOSS_TOL
    assert eerr < EPS_TOL
    assert gerr < GRAD_CTX_TOL
    with torch.no_grad():
        sc = {k: v.detach() for k, v in ex_o["ca_layers_activations"]["attnscore"].items()}
        want = dict(zip(("fg_xlayer_consist", "bg_xlayer_consist"), R.calc_fg_bg_xlayer_consist_loss(sc, subj, bgi, B)))
        want.update(zip(("fg_bg_complem", "subj_mb_suppress", "bg_mf_suppress", "fg_bg_mask_contrast"),
                        R.calc_fg_bg_complementary_loss(sc, subj, bgi, B, fg_grad_scale=0.1, fg_mask=fg64)))
    for k, v in want.items():
        got = float(aux2["reg_parts"][k])
        assert abs(got - float(v)) < 5e-2 * abs(float(v)) + 1e-4, (k, got, float(v))
    assert torch.isfinite(ctx_d2.grad).all() and float((ctx_d2.grad - ctx_d.grad).abs().max()) > 0
    assert not ops.gn_sync_poisoned()


def test_training_window_on_two_lanes_equals_sequential_training_steps():
    """``LatentDiffusion.training_window`` with ``MicroBatchLanes`` (the two micro-batches of an accumulation window on two HIP
    streams, forward-first, gradients meeting in the shared buffer in micro-batch order behind the lanes' gate) against
    ``training_step`` called on the same four micro-batches one after the other: two optimiser steps, regularisers on the
    captured token maps included.  With the GroupNorms held to their two-launch form in both runs (the single-launch exchange
    belongs to one stream per device, so lane 1 always runs two-pass) every kernel is bit-reproducible and the two loops must
    agree BIT FOR BIT -- losses, Prodigy's d, the parameters -- when the window also keeps the lone-stream split-K plans
    (``ADAP_LANES_KSPLIT_SCALE=100``; by default a window on lanes splits K less, see ``training_window``).  In the shipped mode
    the difference is the summation order of the GroupNorm statistics and of the K slabs, carried through an optimiser step: <= 3e-4 on the losses (measured 1.1e-4; the north-star bar is 1e-3)."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion, MicroBatchLanes
    from adaprompt_amd.ldm.prodigy import Prodigy
    from adaprompt_amd.ldm.util import prodigy_linear_schedule
    from adaprompt_amd.parallel import GradReducer
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    B = 2
    fg64, im64 = ellipse_mask(B, 64, 64), border_mask(B, 64, 64, 5)

    def run(mode, two_pass):
        torch.manual_seed(3)
        hook = SyntheticSubjBasisGenerator(n_params=3 * 16 * 77 * 128, tokens=77, dim=128, id_dim=32)
        with torch.no_grad():
            hook.bases.mul_(20.0)
        hook = hook.to(dev())
        ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                                      {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
                                      cond_fn=make_cond_fn(hook, capture=True, regs=True))
        ld.load_state_dict(synth.synthetic_unet_state_dict(ucfg), strict=False)
        ld = ld.to(dev())
        ld.freeze_unet()
        params = list(hook.parameters())
        opt = Prodigy(params, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
        red = GradReducer(params, flat=opt.grad_buffer)
        sched = prodigy_linear_schedule(opt, max_steps=4, warm_up_steps=1, scheduler_cycles=1)
        mbs = []
        for mb in range(4):
            ids = synth.synthetic_input(f"win.ids.{mb}", (B, 32))
            batch = {"zs_id_embs": ids.to(dev()), "fg_mask": fg64[:, 0].to(dev()), "aug_mask": im64[:, 0].to(dev())}
            kw = dict(t=torch.tensor([150 + 200 * mb, 900 - 100 * mb]).to(dev()),
                      noise=synth.synthetic_input(f"win.noise.{mb}", (B, 4, 64, 64)).to(dev()),
                      x_start=synth.synthetic_input(f"win.x0.{mb}", (B, 4, 64, 64)).to(dev()))
            mbs.append((batch, kw))
        torch.cuda.synchronize()
        ops.gn_two_pass(two_pass)
        if two_pass:
            os.environ["ADAP_LANES_KSPLIT_SCALE"] = "100"
        try:
            losses = []
            if mode == "steps":
                for batch, kw in mbs:
                    losses.append(ld.training_step(batch, optimizer=opt, reducer=red, scheduler=sched, **kw)[0])
            else:
                lanes = MicroBatchLanes(params, n=2)
                mode_seen = []
                for w in range(2):
                    out = ld.training_window([mbs[2 * w][0], mbs[2 * w + 1][0]], opt, red, sched, lanes,
                                             step_kwargs=[mbs[2 * w][1], mbs[2 * w + 1][1]], fuse=(mode == "fused"),
                                             after_forward=lambda k: mode_seen.append(ops._MULTI_STREAM))
                    losses += [o[0] for o in out]
                lanes.remove()
                # a window on lanes runs with cache fills draining their stream (ops.note_cache_fill), and only the window
                assert ops._MULTI_STREAM == 0 and (mode == "fused" or mode_seen == [1] * 4), (mode, mode_seen)
            torch.cuda.synchronize()
        finally:
            ops.gn_two_pass(False)
            os.environ.pop("ADAP_LANES_KSPLIT_SCALE", None)
        assert ld.batch_idx == 4 and opt.device_state()["k"] == 2
        return ([float(x) for x in losses], opt.device_state()["d"], [p.detach().cpu().clone() for p in params])

    la, da, pa = run("steps", True)
    lb, db, pb = run("window", True)
    assert la == lb and da == db, (la, lb, da, db)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    assert all(math.isfinite(v) for v in la) and float((pa[0] - pb[0]).abs().max()) == 0.0
    lc, dc, pc = run("window", False)                # two lanes as shipped: lane 0 single-launch GroupNorm, lane 1 two-pass
    for u, v in zip(la, lc):
        assert abs(u - v) <= 3e-4 * abs(u), (la, lc)
    assert abs(da - dc) <= 1e-3 * abs(da)
    # the window as ONE batched UNet pass (``fuse``, the default for windows of plain recon micro-batches): the same fronts,
    # losses and gradients per micro-batch, one forward and one backward at twice the batch -- different tilings and split-K
    # plans at the larger batch, hence round-off only
    lf, df, pf_ = run("fused", False)
    for i, (u, v) in enumerate(zip(la, lf)):
        # (the second window's losses come after an optimiser step on round-off-different gradients -- a first Prodigy step moves
        # every element by ~ lr d sign(g): measured 2.4e-4 / 3.1e-4 there, 1.4e-5 / 4e-6 before it; the north-star bar is 1e-3)
        assert abs(u - v) <= (3e-4 if i < 2 else 6e-4) * abs(u), (la, lf)
    assert abs(da - df) <= 1e-3 * abs(da)
    assert float(max((x - y).abs().max() for x, y in zip(pa, pf_))) <= 1e-3 * float(max(x.abs().max() for x in pa))


def test_fused_window_with_unfrozen_unet_equals_sequential_steps():
    """``unfreeze_model: True``: a window cannot run on lanes (the UNet's weight gradients are written through raw pointers from
    the start of a backward), so ``training_window`` sends its two recon micro-batches through the UNet as ONE batched pass
    (``fuse``, the default without lanes): one backward at twice the batch = ONE weight-gradient pass per window.  Against two
    sequential ``training_step``s: the losses, Prodigy's d and every parameter of hook and UNet after the optimiser step."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaprompt_amd.ldm.prodigy import Prodigy
    from adaprompt_amd.ldm.util import prodigy_linear_schedule
    from adaprompt_amd.parallel import GradReducer
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    B = 2
    fg64, im64 = ellipse_mask(B, 64, 64), border_mask(B, 64, 64, 5)

    def run(fused):
        torch.manual_seed(3)
        hook = SyntheticSubjBasisGenerator(n_params=3 * 16 * 77 * 128, tokens=77, dim=128, id_dim=32)
        with torch.no_grad():
            hook.bases.mul_(20.0)
        hook = hook.to(dev())
        ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                                      {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
                                      cond_fn=make_cond_fn(hook, capture=True, regs=True))
        ld.load_state_dict(synth.synthetic_unet_state_dict(ucfg), strict=False)
        ld = ld.to(dev())
        for p_ in ld.model.parameters():
            p_.requires_grad_(True)
        groups = [{"params": list(hook.parameters())}, {"params": list(ld.model.parameters())}]
        params = [q for g_ in groups for q in g_["params"]]
        opt = Prodigy(groups, lr=1.0, betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.0)
        red = GradReducer(params, flat=opt.grad_buffer)
        sched = prodigy_linear_schedule(opt, max_steps=4, warm_up_steps=1, scheduler_cycles=1)
        mbs = []
        for mb in range(2):
            ids = synth.synthetic_input(f"win.ids.{mb}", (B, 32))
            batch = {"zs_id_embs": ids.to(dev()), "fg_mask": fg64[:, 0].to(dev()), "aug_mask": im64[:, 0].to(dev())}
            kw = dict(t=torch.tensor([150 + 200 * mb, 900 - 100 * mb]).to(dev()),
                      noise=synth.synthetic_input(f"win.noise.{mb}", (B, 4, 64, 64)).to(dev()),
                      x_start=synth.synthetic_input(f"win.x0.{mb}", (B, 4, 64, 64)).to(dev()))
            mbs.append((batch, kw))
        before = [p_.detach().clone() for p_ in params]
        if fused:
            out = ld.training_window([mbs[0][0], mbs[1][0]], opt, red, sched, None, step_kwargs=[mbs[0][1], mbs[1][1]])
            losses = [float(o[0]) for o in out]
        else:
            losses = [float(ld.training_step(batch, optimizer=opt, reducer=red, scheduler=sched, **kw)[0]) for batch, kw in mbs]
        torch.cuda.synchronize()
        assert ld.batch_idx == 2 and opt.device_state()["k"] == 1
        moved = [(a_.detach() - b_).float().cpu() for a_, b_ in zip(params, before)]
        return losses, opt.device_state()["d"], moved

    la, da, ma = run(False)
    lb, db, mb_ = run(True)
    for u, v in zip(la, lb):
        assert abs(u - v) <= 3e-4 * abs(u), (la, lb)
    assert abs(da - db) <= 2e-3 * abs(da), (da, db)
    num = sum(float((x - y).pow(2).sum()) for x, y in zip(ma, mb_)) ** 0.5
    den = sum(float(x.pow(2).sum()) for x in ma) ** 0.5
    # the step's displacement of hook + all 686 UNet tensors.  A FIRST Adam-type step moves every element by ~ lr d sign(g): the
    # elements whose gradient is at round-off level flip with the summation order of the batch-4 / batch-8 weight-gradient passes
    # (measured 4 % of the displacement's norm; the 5-step trajectory test allows hip-vs-oracle 10 % for the same reason)
    assert den > 0 and num <= 8e-2 * den, (num, den)


def test_batched_token_maps_equal_the_per_layer_launches():
    """``functional.BATCH_TOKMAPS``: the 12 distillation layers' token-map captures as ONE launch behind the UNet's forward
    (``adap_attention_tokmap_fwd_batched``) and their gradient prologues as THREE in front of the backward
    (``prepare_tokmap_backward`` -> ``adap_attention_tokmap_prep_batched``) against one capture launch and three prologue launches
    inside every block: the same arithmetic per layer, so eps, the token maps and the context gradient are equal bit for bit."""
    from adaprompt_amd import functional as HF
    B, M = 2, 77
    unet = build_unet(NARROW)
    x = synth.synthetic_input("btm.x", (B, 4, 64, 64)).to(dev())
    t = torch.tensor([300, 760]).to(dev())
    ctx0 = synth.synthetic_input("btm.ctx", (16 * B, M, NARROW["context_dim"])).to(dev())
    w = synth.synthetic_input("btm.gw", (B, 4, 64, 64)).to(dev())
    subj = (torch.arange(B, device=dev()).repeat_interleave(16), torch.arange(4, 20, device=dev()).repeat(B))
    bgi = (torch.arange(B, device=dev()).repeat_interleave(4), torch.arange(24, 28, device=dev()).repeat(B))

    def run(batched):
        was = HF.BATCH_TOKMAPS
        HF.BATCH_TOKMAPS = batched
        try:
            ctx = ctx0.clone().requires_grad_(True)
            extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon", "is_training": True,
                     "capture_distill_attn": True, "placeholder2indices": None, "img_mask": border_mask(B, 64, 64, 6).to(dev()),
                     "subj_indices": subj, "bg_indices": bgi, "capture_token_maps_only": True}
            eps = unet(x, t, context=ctx, context_in=None, extra_info=extra)
            tms = extra["ca_layers_activations"]["attnscore_tokmap"]
            assert len(tms) == 12
            roots = [eps] + [tms[li] for li in sorted(tms)]
            grads = [w] + [synth.synthetic_input(f"btm.dt{li}", tuple(tms[li].shape)).to(dev()) * 1e-3 for li in sorted(tms)]
            HF.prepare_tokmap_backward(roots, grads)
            given = sum(getattr(r.grad_fn, "tok_prep", None) is not None for r in roots[1:])
            torch.autograd.backward(roots, grads)
            torch.cuda.synchronize()
            return eps.detach().clone(), ctx.grad.detach().clone(), {li: v.detach().clone() for li, v in tms.items()}, given
        finally:
            HF.BATCH_TOKMAPS = was

    e1, g1, tm1, n1 = run(True)
    e0, g0, tm0, n0 = run(False)
    assert n1 == 12 and n0 == 0, (n1, n0)
    assert torch.equal(e1, e0) and torch.equal(g1, g0)
    for li in tm0:
        assert torch.equal(tm1[li], tm0[li]), li
    assert float(g1.abs().max()) > 0 and torch.isfinite(g1).all()


ROLLOUT_EPS_TOL = EPS_TOL  # teacher eps / x0 at every rollout step (measured 1e-2 / 4e-3: the feedback does not amplify)
DISTILL_LOSS_TOL = LOSS_TOL  # sum over steps of masked MSE(student eps, teacher eps): the north-star bar (measured 2.6e-4)


@pytest.mark.parametrize("size", ["narrow", "sd15"])
def test_arc2face_distill_step_vs_oracle(size):
    """a9/a17: three-step teacher rollout (SD-topology UNet with its own weights, [B,21,ctx] context repeated
    over the layers) + student passes on the teacher's predictions + loss / sqrt(3) + gradient into the student's
    context, against oracle/distill_oracle.py with the fp32 UNet restatement in both roles; rand / randn draws
    supplied.  The timesteps the rollout visits are integer work and must be identical.  ``narrow``: HALF_BS = 2, the batched and
    the one-by-one student; ``sd15`` (round 5: config 2's iteration at FULL size, 859.5 M-parameter student and teacher): HALF_BS = 1
    of a batch of 2, the batched student (three passes as one)."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion, Arc2FaceWrapper
    from oracle import ldm_oracle as O
    from oracle import distill_oracle as D
    full = size == "sd15"
    ucfg = dict(synth.SD15_UNET) if full else dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    usd = synth.synthetic_unet_state_dict(ucfg)
    ld.load_state_dict(usd, strict=False)
    ld = ld.to(dev())
    ld.freeze_unet()
    TP = "arc2face.unet."                                    # teacher weights: same shapes, different values
    tsd = synth.synthetic_unet_state_dict(ucfg, prefix=TP)
    teacher = Arc2FaceWrapper(unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                                           "params": ucfg})
    teacher.unet.load_state_dict({k[len(TP):]: v for k, v in tsd.items()}, strict=True)
    ld.set_arc2face_teacher(teacher.to(dev()).eval())
    assert not any(k.startswith("arc2face") for k in ld.state_dict())

    B, nd = (1 if full else 2), 3
    x0 = synth.synthetic_input("distill.x0", (B, 4, 64, 64))
    noises = [synth.synthetic_input(f"distill.noise{i}", (B, 4, 64, 64)) for i in range(nd)]
    rel = [synth.synthetic_input(f"distill.rel{i}", (B,)).sigmoid() for i in range(nd - 1)]     # in (0, 1)
    t = torch.tensor([420, 640])[:B]
    tctx = synth.synthetic_input("distill.tctx", (B, 21, ucfg["context_dim"]))
    ctx = synth.synthetic_input("distill.ctx", (16 * B, 77, ucfg["context_dim"]))
    fg64, im64 = ellipse_mask(B, 64, 64), border_mask(B, 64, 64, 5)
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": True}

    # ---- oracle
    sch = O.make_schedule()
    t_ref = D.shift_t_for_multistep(t, nd)
    ex_t = dict(extra, capture_distill_attn=False, img_mask=None)

    def teacher_fn(x_noisy, t_i, c):
        c16 = c[:, None].expand(B, 16, 21, c.shape[-1]).reshape(16 * B, 21, c.shape[-1])
        return O.unet_forward(tsd, ucfg, x_noisy, t_i, c16, ex_t, prefix=TP)

    tout = D.arc2face_rollout(teacher_fn, sch, x0, noises[0], t_ref, tctx, nd, rel, noises)
    ctx_ref = ctx.clone().requires_grad_(True)
    ex_s = dict(extra, capture_distill_attn=False, img_mask=im64)
    loss_ref, losses_ref, outs_ref, start = D.arc2face_distill_loss(
        lambda xn, t2: O.unet_forward(usd, ucfg, xn, t2, ctx_ref, ex_s), sch, tout, im64, fg64, nd)
    (g_ref,) = torch.autograd.grad(loss_ref, ctx_ref)

    # ---- HIP
    ctx_hip = ctx.to(dev()).clone().requires_grad_(True)
    batch = {"fg_mask": fg64[:, 0].to(dev()), "aug_mask": im64[:, 0].to(dev()),
             "arc2face_prompt_emb": tctx.to(dev())}
    kw = dict(t=t.to(dev()), noise=noises[0].to(dev()), x_start=x0.to(dev()), num_denoising_steps=nd,
              use_arc2face_as_target=True, relative_ts=[r.to(dev()) for r in rel], noises=[n.to(dev()) for n in noises],
              trim_to_half_batch=False)                                    # B = 2 IS the HALF_BS of a batch of 4
    # the student's passes as ONE batched pass (the default) ...
    loss, grads, outs, aux = ld.shared_step(batch, cond=(ctx_hip, None, extra), **kw)
    assert len(outs) == 1 and outs[0].shape[0] == nd * B
    torch.autograd.backward(outs, grads)
    # ... and one after the other, as the reference does.  The two are the same arithmetic in exact terms but not bit
    # for bit: the contraction kernels pick tile sizes and split-K plans from M = batch x pixels, so the summation
    # order of the bf16-product accumulations differs between a batch of 6 and three batches of 2 (measured: loss
    # 3.7e-5 apart).  Each is separately held to the oracle below; against each other they get a fraction of that bar.
    lr_seq = None
    if not full:
        ctx_seq = ctx.to(dev()).clone().requires_grad_(True)
        loss_s, grads_s, outs_s, aux_s = ld.shared_step(batch, cond=(ctx_seq, None, extra), batched_student=False, **kw)
        assert len(outs_s) == nd
        torch.autograd.backward(outs_s, grads_s)
        assert abs(float(loss) - float(loss_s)) / float(loss_s) < LOSS_TOL / 5
        assert rel_err(ctx_hip.grad, ctx_seq.grad) < 5e-3
        for a, b in zip(aux["model_outputs_per_step"], outs_s):
            assert rel_err(a.detach(), b.detach()) < 8e-3             # (measured 4.6e-3 ... 5.1e-3: the batch of 6 and the batches of
            #                                                             2 get different tile / split-K plans and attention kernels)
        lr_seq = float(loss_s)
    outs = aux["model_outputs_per_step"]
    npred, px0, nz, ts = aux["teacher"]
    assert aux["loss_start_step"] == start == 0 and len(outs) == nd
    for a, b in zip(ts, tout[3]):
        assert torch.equal(a.cpu(), b), (a, b)                       # integer work: identical
    errs = []
    for i in range(nd):
        e_eps, e_x0 = rel_err(npred[i].cpu(), tout[0][i]), rel_err(px0[i].cpu(), tout[1][i])
        errs.append((e_eps, e_x0))
        assert e_eps < (EPS_TOL if i == 0 else ROLLOUT_EPS_TOL), (i, e_eps)
        assert e_x0 < ROLLOUT_EPS_TOL, (i, e_x0)
    for a, b in zip(outs, outs_ref):
        assert rel_err(a.detach().cpu(), b.detach()) < ROLLOUT_EPS_TOL
    lr_ = float(loss_ref)
    print(f"[distill {size}] teacher (eps, x0) rel err per step {[(round(a, 4), round(b, 4)) for a, b in errs]}  "
          f"loss hip {float(loss):.6f} ref {lr_:.6f} rel {abs(float(loss) - lr_) / lr_:.2e}  "
          f"grad rel {rel_err(ctx_hip.grad.cpu(), g_ref):.3e}")
    assert abs(float(loss) - lr_) / lr_ < DISTILL_LOSS_TOL
    if lr_seq is not None:
        assert abs(lr_seq - lr_) / lr_ < DISTILL_LOSS_TOL          # the one-by-one path against the oracle too
    assert rel_err(ctx_hip.grad.cpu(), g_ref) < (GRAD_CTX_TOL if full else 1.5e-2)      # measured 6.2e-3 narrow


def test_guided_denoise_cfg_pixel_recon_vs_oracle():
    """a10: the do_pixel_recon branch -- uncond pass on the first half batch, repeated, CFG combine, x0 prediction."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ldm_oracle as O
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    usd = synth.synthetic_unet_state_dict(ucfg)
    ld.load_state_dict(usd, strict=False)
    ld = ld.to(dev())
    B = 4
    x0 = synth.synthetic_input("cfg.x0", (B, 4, 64, 64))
    x0[2:] = x0[:2]                                     # second half shares the initial conditions of the first
    noise = synth.synthetic_input("cfg.noise", (B, 4, 64, 64))
    noise[2:] = noise[:2]
    t = torch.tensor([300, 700, 300, 700])
    ctx = synth.synthetic_input("cfg.ctx", (16 * B, 77, ucfg["context_dim"]))
    uctx = synth.synthetic_input("cfg.uctx", (16 * 2, 77, ucfg["context_dim"]))
    scales = torch.tensor([1.5, 2.0, 3.0, 1.0])
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": False, "img_mask": None}
    sch = O.make_schedule()
    with torch.no_grad():
        xn = O.q_sample(sch, x0, t, noise)
        eps = O.unet_forward(usd, ucfg, xn, t, ctx, extra)
        eps_u = O.unet_forward(usd, ucfg, xn[:2], t[:2], uctx, extra).repeat(2, 1, 1, 1)
        s4 = scales.view(-1, 1, 1, 1)
        x_rec_ref = O.predict_start_from_noise(sch, xn, t, eps * s4 - eps_u * (s4 - 1))
    with torch.no_grad():
        mo, x_rec = ld.guided_denoise(x0.to(dev()), noise.to(dev()), t.to(dev()), (ctx.to(dev()), None, dict(extra)),
                                      unet_has_grad=False, do_pixel_recon=True,
                                      cfg_info={"uncond_context": (uctx.to(dev()), None, dict(extra)),
                                                "cfg_scales": scales.to(dev())})
        _, x_rec_nocfg = ld.guided_denoise(x0.to(dev()), noise.to(dev()), t.to(dev()),
                                           (ctx.to(dev()), None, dict(extra)), unet_has_grad=False,
                                           do_pixel_recon=True,
                                           cfg_info={"uncond_context": (uctx.to(dev()), None, dict(extra)),
                                                     "cfg_scales": None})
    assert rel_err(mo.cpu(), eps) < EPS_TOL
    assert rel_err(x_rec.cpu(), x_rec_ref) < EPS_TOL, rel_err(x_rec.cpu(), x_rec_ref)
    assert rel_err(x_rec_nocfg.cpu(), O.predict_start_from_noise(sch, xn, t, eps)) < EPS_TOL


def test_distill_prefetcher_matches_inline_path():
    """The side-stream front of a distillation micro-batch (VAE encode + trim + t shift + teacher rollout,
    DistillPrefetcher) must hand the student exactly what the inline path computes: same RNG stream, same kernels,
    only another HIP stream -- bit-identical loss and gradient.  Inputs are dropped right after submit() on purpose:
    the prefetcher has to keep them alive for the side stream (record_stream)."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion, Arc2FaceWrapper
    ucfg = dict(NARROW)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=512)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    ld.load_state_dict({**synth.synthetic_unet_state_dict(ucfg), **synth.synthetic_vae_state_dict(vdd)}, strict=False)
    ld = ld.to(dev())
    ld.freeze_unet()
    TP = "arc2face.unet."
    teacher = Arc2FaceWrapper(unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    teacher.unet.load_state_dict({k[len(TP):]: v for k, v in synth.synthetic_unet_state_dict(ucfg, prefix=TP).items()})
    ld.set_arc2face_teacher(teacher.to(dev()).eval())
    B, nd = 4, 3
    img = synth.synthetic_input("pf.img", (B, 512, 512, 3), 0, 0.5).clamp(-1, 1).to(dev())
    batch = {"image": img, "fg_mask": ellipse_mask(B, 512, 512)[:, 0].to(dev()),
             "aug_mask": border_mask(B, 512, 512, 32)[:, 0].to(dev()),
             "arc2face_prompt_emb": synth.synthetic_input("pf.tctx", (B, 21, ucfg["context_dim"])).to(dev())}
    ctx = synth.synthetic_input("pf.ctx", (16 * 2, 77, ucfg["context_dim"]))          # HALF_BS = 2 for ND = 3
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": True, "capture_distill_attn": False}

    def inputs():
        return (synth.synthetic_input("pf.pn", (B, 4, 64, 64)).to(dev()), torch.tensor([300, 720, 5, 999], device=dev()),
                synth.synthetic_input("pf.noise", (B, 4, 64, 64)).to(dev()))

    # with and without the recon iteration's timestep annealing (applied before the multi-step shift on both paths,
    # reference ddpm.py:2851-2861; device tensors draw it from torch's generator, so the seed pins it)
    for anneal in (False, True):
        ld.training_percent = 0.3
        # inline
        import random
        torch.manual_seed(11)
        random.seed(2)                  # the keep / redraw decision of the annealing is one random.random() (2 -> redraw)
        pn, t, noise = inputs()
        c1 = ctx.to(dev()).clone().requires_grad_(True)
        loss1, g1, o1, aux1 = ld.shared_step(batch, t=t, noise=noise, post_noise=pn, cond=(c1, None, extra),
                                             num_denoising_steps=nd, use_arc2face_as_target=True, anneal_t=anneal)
        torch.autograd.backward(o1, g1)
        # prefetched
        torch.manual_seed(11)
        random.seed(2)
        pf = ld.make_distill_prefetcher()
        pn, t, noise = inputs()
        pf.submit(batch, pn, t, noise, nd, anneal_t=anneal)
        del pn, t, noise
        junk = [torch.full((B, 4, 64, 64), float("nan"), device=dev()) for _ in range(8)]      # reuse freed blocks, if any
        junk.append(torch.full((B,), 123456789, device=dev(), dtype=torch.long))
        x_start, t2, noise2, teacher_out, hb = pf.get()
        assert hb == 2
        b2 = {k: v[:hb] for k, v in batch.items()}
        c2 = ctx.to(dev()).clone().requires_grad_(True)
        loss2, g2, o2, aux2 = ld.shared_step(b2, t=t2, noise=noise2, x_start=x_start, cond=(c2, None, extra),
                                             num_denoising_steps=nd, use_arc2face_as_target=True,
                                             trim_to_half_batch=False, teacher_out=teacher_out, anneal_t=anneal)
        torch.autograd.backward(o2, g2)
        del junk
        for a, b in zip(aux1["teacher"][3], aux2["teacher"][3]):
            assert torch.equal(a, b), anneal                      # the timesteps: exactly the same draws
        # the values agree to rounding, not bit for bit: the side stream's GroupNorms run the two-pass kernels (the
        # single-launch kernel's cross-workgroup exchange belongs to ONE stream per device, ops.gn_sync_buffer), and their
        # statistics are summed in a different order
        assert abs(float(loss1) - float(loss2)) / float(loss1) < LOSS_TOL
        assert rel_err(c2.grad, c1.grad) < 5e-3
        if anneal:          # annealing only ever raises t (ratio in [1, 1.3]) before the shift
            shifted_plain = ld.shift_t_for_multistep(inputs()[1][:hb], nd)
            assert bool((aux2["teacher"][3][0] >= shifted_plain).all())


def test_cpu_tensor_fails_loudly():
    unet = build_unet(NARROW)
    with pytest.raises(RuntimeError):
        unet(torch.zeros(1, 4, 64, 64), torch.zeros(1, dtype=torch.long), context=torch.zeros(16, 77, 128),
             extra_info={"use_layerwise_context": True, "use_conv_attn_kernel_size": -1})


@pytest.mark.parametrize("ch,B,H", [(32, 3, 128), (128, 2, 256)])
def test_vae_encode_single_c_entry_equals_the_python_sequencing(ch, B, H):
    """``adap_vae_encode`` (vae_encode(x, masks, weights*, noise, z)) issues the launches of the first stage's
    encode from host code inside the library: bit-identical to the Python mirror issuing them one by one, with the fg / bg
    mask of the mid attention, and the scaled posterior sample.  The second case (full channel widths, 16.8 M-element level-0
    tensors) takes the GroupNorm statistics of the first level from the conv epilogues on both sides (``stats_from_epilogue``)."""
    from adaprompt_amd import ops
    from adaprompt_amd.ldm.util import instantiate_from_config
    vdd = dict(synth.SD15_VAE_DD, ch=ch, resolution=H)
    ae = instantiate_from_config({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}})
    sd = {k[len("first_stage_model."):]: v for k, v in synth.synthetic_vae_state_dict(vdd).items()}
    ae.load_state_dict(sd, strict=False)
    ae = ae.to(dev())
    x = synth.synthetic_input("vaec.x", (B, H, H, 3), 0, 0.5).clamp(-1, 1).to(dev())
    fg = ellipse_mask(B, H, H).to(dev())
    aug = border_mask(B, H, H, 9).to(dev())
    noise = synth.synthetic_input("vaec.noise", (B, H // 8, H // 8, 4)).to(dev())
    for mask in (None, {"fg_mask": fg, "aug_mask": aug}):
        ref_m = ae.encode_moments_nhwc(x, mask)
        ref_z = ops.posterior_sample(ref_m, noise, 0.18215)
        m, z = ae.encode_c_abi(x, mask, noise, 0.18215)
        assert torch.equal(m, ref_m) and torch.equal(z, ref_z)
    m2, z2 = ae.encode_c_abi(x, None, None)
    assert z2 is None and torch.equal(m2, ae.encode_moments_nhwc(x, None))

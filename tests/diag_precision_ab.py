#!/usr/bin/env python3
"""Where does the context-gradient error of the bf16 path come from?  (GPU only; diagnostic behind tests/test_precision_gpu.py; it calls the oracle, hence its place under tests/)

For the narrow smoke case and the UNet-only golden cases: relative L2 error of d loss / d context against the f32 oracle,
per context layer, in the shipped mode and in the f32-storage validation mode (functional.set_f32_storage), and -- for the
smoke case -- with the latent encoded by the HIP VAE vs handed over from the oracle, which separates the error of the
backward pass from the sensitivity of the gradient to a ~1 % perturbed input."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from adaprompt_amd import functional as Fn
from adaprompt_amd import synth
from conftest import load_golden, rel_err

dev = torch.device("cuda:0")


def per_layer(g, ref, B):
    g, ref = g.view(B, 16, *g.shape[1:]), ref.view(B, 16, *ref.shape[1:])
    return [rel_err(g[:, l], ref[:, l]) for l in range(16)]


def unet_case(cfg, tag, gname, subs):
    import test_model_gpu as T
    g = load_golden(gname)
    B, M = g["B"], g["M"]
    unet = T.build_unet(cfg)
    x = synth.synthetic_input(f"unet.{tag}.x", (B, 4, 64, 64)).to(dev)
    ctx0 = synth.synthetic_input(f"unet.{tag}.ctx", (16 * B, M, cfg["context_dim"])).to(dev)
    w = synth.synthetic_input(f"unet.{tag}.gw", (B, 4, 64, 64)).to(dev)
    for mode in (False, True):
        Fn.set_f32_storage(mode)
        ctx = ctx0.clone().requires_grad_(True)
        extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": g["iter_type"], "is_training": True,
                 "capture_distill_attn": bool(g["capture"]), "placeholder2indices": None, "img_mask": None}
        eps = unet(x, g["t"].to(dev), context=ctx, context_in=None, extra_info=extra)
        (eps * w).sum().backward()
        gr = ctx.grad.cpu()
        got = gr[:, ::4, ::8] if subs else gr
        pl = per_layer(got, g["grad_context"], B)
        print(f"[{tag}] f32_storage={int(mode)}  eps {rel_err(eps.detach().cpu(), g['eps']):.3e}  grad_context {rel_err(got, g['grad_context']):.3e}"
              f"  per layer: " + " ".join(f"{e:.3f}" for e in pl), flush=True)
    Fn.set_f32_storage(False)


def smoke_case():
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ldm_oracle as O
    ucfg = dict(synth.SD15_UNET, model_channels=64, context_dim=128)
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=512)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
                                  {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    usd, vsd = synth.synthetic_unet_state_dict(ucfg), synth.synthetic_vae_state_dict(vdd)
    ld.load_state_dict({**usd, **vsd}, strict=False)
    ld = ld.to(dev)
    ld.freeze_unet()
    B = 1
    img = synth.synthetic_input("smoke.img", (B, 512, 512, 3), 0, 0.5).clamp(-1, 1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    fg = ((xx / 0.62) ** 2 + (yy / 0.72) ** 2 <= 1.0).float()[None].repeat(B, 1, 1)
    aug = torch.zeros(B, 512, 512)
    aug[:, 24:488, 24:488] = 1
    pn = synth.synthetic_input("smoke.pn", (B, 4, 64, 64))
    noise = synth.synthetic_input("smoke.noise", (B, 4, 64, 64))
    t = torch.tensor([417])
    ctx = synth.synthetic_input("smoke.ctx", (16 * B, 77, 128))
    fg64 = torch.nn.functional.interpolate(fg[:, None], size=(64, 64), mode="nearest")
    im64 = torch.nn.functional.interpolate(aug[:, None], size=(64, 64), mode="nearest")
    ref = O.recon_step(usd, vsd, ucfg, vdd, img.permute(0, 3, 1, 2), {"fg_mask": fg[:, None], "aug_mask": aug[:, None]},
                       pn, t, noise, ctx, im64, fg64, 0.1, need_grad=True)
    x_start_o = ref["z"]
    e = ref["eps_hat"].clone().requires_grad_(True)
    (ref["grad_eps"],) = torch.autograd.grad(O.calc_recon_loss(e, noise, im64, fg64, 1.0, 0.1)[0], e)
    ref["eps"] = ref["eps_hat"]
    batch = {"image": img.to(dev), "fg_mask": fg.to(dev), "aug_mask": aug.to(dev)}
    for mode in (False, True):
        for latent in ("hip", "oracle"):
            if latent == "oracle" and x_start_o is None:
                continue
            Fn.set_f32_storage(mode)
            extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon", "is_training": True,
                     "capture_distill_attn": True, "placeholder2indices": None}
            ctx_d = ctx.to(dev).requires_grad_(True)
            kw = {} if latent == "hip" else {"x_start": x_start_o.to(dev)}
            loss, grad, out, aux = ld.shared_step(batch, t=t.to(dev), noise=noise.to(dev), post_noise=pn.to(dev),
                                                  cond=(ctx_d, None, extra), **kw)
            gerr_up = rel_err(grad.cpu(), ref["grad_eps"]) if "grad_eps" in ref else float("nan")
            ld.manual_backward(out, grad, aux)
            gr = ctx_d.grad.cpu()
            print(f"[smoke] f32_storage={int(mode)} latent={latent:6s} loss rel {abs(float(loss) - float(ref['loss'])) / float(ref['loss']):.2e}  "
                  f"eps {rel_err(out.detach().cpu(), ref['eps']) if 'eps' in ref else float('nan'):.3e}  upstream grad {gerr_up:.3e}  "
                  f"grad_context {rel_err(gr, ref['grad_context']):.3e}  per layer: "
                  + " ".join(f"{e:.3f}" for e in per_layer(gr, ref['grad_context'], B)), flush=True)
            # the backward alone: the oracle's upstream gradient pushed through the HIP backward
            if "grad_eps" in ref:
                ctx_d2 = ctx.to(dev).requires_grad_(True)
                loss, grad, out, aux = ld.shared_step(batch, t=t.to(dev), noise=noise.to(dev), post_noise=pn.to(dev),
                                                      cond=(ctx_d2, None, extra), **kw)
                ld.manual_backward(out, ref["grad_eps"].to(dev), aux)
                print(f"        ... with the oracle's d loss / d eps as the upstream gradient: grad_context "
                      f"{rel_err(ctx_d2.grad.cpu(), ref['grad_context']):.3e}", flush=True)
    Fn.set_f32_storage(False)


if __name__ == "__main__":
    smoke_case()
    unet_case(dict(synth.SD15_UNET, model_channels=64, context_dim=128), "narrow_recon", "unet_narrow_recon", False)
    unet_case(dict(synth.SD15_UNET), "sd15_recon", "unet_sd15_recon", True)

"""Inference path on the MI355X: VAE decoder against the reference's own outputs, the DDIM
sampler against the oracle loop with the fp32 UNet restatement as the eps-predictor."""
import pytest
import torch

from adaprompt_amd import synth
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

DEC_TOL = 2.5e-2        # decoded image, relative L2 (bf16 operands through ~30 convolutions + one attention)
DDIM_TOL = 3e-2         # latent after 4 guided DDIM steps (each step feeds eps-hat error back into x)


def dev():
    return torch.device("cuda:0")


def build_vae(dd, decoder=True):
    from adaprompt_amd.ldm.models.autoencoder import AutoencoderKL
    vae = AutoencoderKL(dd, embed_dim=4, with_decoder=decoder)
    P = "first_stage_model."
    sd = {k[len(P):]: v for k, v in synth.synthetic_vae_state_dict(dd, decoder=decoder).items()}
    vae.load_state_dict(sd, strict=decoder)          # without the decoder, post_quant_conv has no synthetic weights
    return vae.to(dev()).eval()


@pytest.mark.parametrize("tag,dd", [("narrow", dict(synth.SD15_VAE_DD, ch=32, resolution=64)),
                                    ("sd15", dict(synth.SD15_VAE_DD))])
def test_vae_decode_vs_reference_golden(tag, dd):
    g = load_golden(f"vae_decode_{tag}")
    vae = build_vae(dd)
    res, B, sub = int(g["res"]), int(g["B"]), int(g["sub"])
    z = synth.synthetic_input(f"dec.{tag}.z", (B, 4, res // 8, res // 8), 0, 1.0).to(dev())
    img = vae.decode(z)
    assert img.shape == (B, 3, res, res)
    e = rel_err(img[:, :, ::sub, ::sub].float().cpu(), g["image"])
    print(f"[decode {tag}] rel L2 {e:.3e}")
    assert e < DEC_TOL


def test_decoder_absent_fails_loudly():
    vae = build_vae(dict(synth.SD15_VAE_DD, ch=32, resolution=64), decoder=False)
    with pytest.raises(RuntimeError):
        vae.decode(torch.zeros(1, 4, 8, 8, device=dev()))


@pytest.mark.parametrize("size,B", [("narrow", 1), ("sd15", 1)])
def test_ddim_sampler_vs_oracle_and_decode_first_stage(size, B):
    """4 guided DDIM steps + VAE decode against the oracle loop (itself pinned by the reference's own DDIMSampler,
    tests/test_ddim_golden.py) with the fp32 UNet / decoder restatements: at narrow widths and at the FULL SD-1.5 sizes (859.5 M
    UNet, 49.5 M decoder; one image, ~40 s of CPU oracle)."""
    from adaprompt_amd.ldm.models.diffusion.ddim import DDIMSampler
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ddim_oracle as DO
    from oracle import ldm_oracle as O
    if size == "narrow":
        ucfg = dict(synth.SD15_UNET, model_channels=64, context_dim=128)
        vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=512)
    else:
        ucfg, vdd = dict(synth.SD15_UNET), dict(synth.SD15_VAE_DD)
    ld = LatentDiffusion.hot_path({"target": "ldm.models.autoencoder.AutoencoderKL",
                          "params": {"ddconfig": vdd, "embed_dim": 4, "with_decoder": True}},
                         {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg})
    usd = synth.synthetic_unet_state_dict(ucfg)
    vsd = synth.synthetic_vae_state_dict(vdd, decoder=True)
    missing, unexpected = ld.load_state_dict({**usd, **vsd}, strict=False)
    assert not unexpected
    ld = ld.to(dev()).eval()
    S = 4 if size == "narrow" else 2            # (full size: two guided steps -- the loop's arithmetic is the narrow case's business)
    x_T = synth.synthetic_input("ddim.xT", (B, 4, 64, 64))
    ctx = synth.synthetic_input("ddim.ctx", (16 * B, 77, ucfg["context_dim"]))
    uctx = synth.synthetic_input("ddim.uctx", (16 * B, 77, ucfg["context_dim"]))
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
             "is_training": False, "capture_distill_attn": False, "img_mask": None}
    sch = O.make_schedule()

    def eps_fn(x, t, c):
        return O.unet_forward(usd, ucfg, x, t, c[0], extra)

    with torch.no_grad():
        prompts, negs = ["a photo of z"] * B, [""] * B          # c_in: prompt lists, concatenated by the sampler
        z_ref, preds_ref = DO.ddim_sampling(eps_fn, sch["alphas_cumprod"], (ctx, prompts, extra), x_T, S, (3.0, 1.5),
                                            (uctx, negs, extra))
        img_ref = O.decode_first_stage(vsd, vdd, z_ref)
    sampler = DDIMSampler(ld)
    z, inter = sampler.sample(S=S, batch_size=B, shape=[4, 64, 64], conditioning=(ctx.to(dev()), prompts, dict(extra)),
                              verbose=False, guidance_scale=[3.0, 1.5],
                              unconditional_conditioning=(uctx.to(dev()), negs, dict(extra)), eta=0.0,
                              x_T=x_T.to(dev()))
    assert sampler.ddim_timesteps.tolist() == DO.make_ddim_timesteps("uniform", S, 1000).tolist()
    e = rel_err(z.cpu(), z_ref)
    img = ld.decode_first_stage(z)
    ei = rel_err(img.float().cpu(), img_ref)
    print(f"[ddim {size}] latent rel L2 after {S} guided steps {e:.3e}; decoded image rel L2 {ei:.3e}")
    assert e < DDIM_TOL
    assert img.shape == (B, 3, 512, 512) and ei < 2 * DDIM_TOL
    with pytest.raises(ValueError):
        sampler.sample(S=S, batch_size=B, shape=[4, 64, 64], conditioning=(ctx.to(dev()), None, dict(extra)),
                       verbose=False, guidance_scale=7.5, unconditional_conditioning=(uctx.to(dev()), None, dict(extra)),
                       x_T=x_T.to(dev()))

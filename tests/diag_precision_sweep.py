#!/usr/bin/env python3
"""eps-hat error of the narrow HIP UNet vs the oracle over input variations (GPU only; diagnostic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from adaprompt_amd import synth
from conftest import rel_err
from oracle import ldm_oracle as O
import test_model_gpu as T

dev = torch.device("cuda:0")
cfg = dict(synth.SD15_UNET, model_channels=64, context_dim=128)
usd = synth.synthetic_unet_state_dict(cfg)
unet = T.build_unet(cfg)


def case(name, B, t, ctx_scale, x_scale, seed_tag="sweep"):
    x = synth.synthetic_input(f"{seed_tag}.x", (B, 4, 64, 64)) * x_scale
    ctx = synth.synthetic_input(f"{seed_tag}.ctx", (16 * B, 77, 128)) * ctx_scale
    tt = torch.full((B,), t)
    ex = lambda: {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon", "is_training": True,
                  "capture_distill_attn": False, "placeholder2indices": None, "img_mask": None}
    with torch.no_grad():
        ref = O.unet_forward(usd, cfg, x, tt, ctx, ex())
        got = unet(x.to(dev), tt.to(dev), context=ctx.to(dev), context_in=None, extra_info=ex()).cpu()
    print(f"{name:40s} eps rel err {rel_err(got, ref):.3e}   |eps| rms {float(ref.pow(2).mean().sqrt()):.3f}", flush=True)


case("B1 t417 ctx1 x1", 1, 417, 1.0, 1.0)
case("B2 t417 ctx1 x1", 2, 417, 1.0, 1.0)
case("B1 t100 ctx1 x1", 1, 100, 1.0, 1.0)
case("B1 t900 ctx1 x1", 1, 900, 1.0, 1.0)
case("B1 t417 ctx0.3 x1", 1, 417, 0.3, 1.0)
case("B1 t417 ctx3 x1", 1, 417, 3.0, 1.0)
case("B1 t417 ctx1 x0.78", 1, 417, 1.0, 0.78)
case("B1 t417 ctx1 x1 (smoke inputs)", 1, 417, 1.0, 1.0, "smoke")

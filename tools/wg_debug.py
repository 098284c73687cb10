import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from adaprompt_amd import synth
from conftest import rel_err, border_mask
from test_model_gpu import build_unet, NARROW, dev
from oracle import ldm_oracle as O
cfg = dict(NARROW); B = 2; P = "model.diffusion_model."
usd = synth.synthetic_unet_state_dict(cfg, prefix=P)
x = synth.synthetic_input("unet.wg.x", (B, 4, 64, 64)); t = torch.tensor([120, 870])
ctx = synth.synthetic_input("unet.wg.ctx", (16 * B, 77, cfg["context_dim"]))
im = border_mask(B, 64, 64, 6)
extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
         "is_training": True, "capture_distill_attn": False, "placeholder2indices": None, "img_mask": im}
ctx_ref = ctx.clone().requires_grad_(True)
eps_ref = O.unet_forward(usd, cfg, x, t, ctx_ref, dict(extra), prefix=P)
g_eps = synth.synthetic_input("unet.wg.gw", tuple(eps_ref.shape))
eps_ref.backward(g_eps)
def run(train):
    unet = build_unet(cfg, train=train)
    c = ctx.to(dev()).clone().requires_grad_(True)
    e = dict(extra, img_mask=im.to(dev()))
    eps = unet(x.to(dev()), t.to(dev()), context=c, context_in=None, extra_info=e)
    eps.backward(g_eps.to(dev()))
    return eps.detach().cpu(), c.grad.cpu()
e1, g1 = run(True); e0, g0 = run(False); e0b, g0b = run(False)
print("eps train vs frozen", rel_err(e1, e0), " frozen vs frozen", rel_err(e0b, e0))
print("eps vs oracle: train", rel_err(e1, eps_ref.detach()), "frozen", rel_err(e0, eps_ref.detach()))
print("gctx train vs frozen", rel_err(g1, g0), " frozen vs frozen", rel_err(g0b, g0))
print("gctx vs oracle: train", rel_err(g1, ctx_ref.grad), "frozen", rel_err(g0, ctx_ref.grad))
for i in range(0, 32, 2):
    print(i // 2, "layer ctx grad train vs frozen %.3e   train vs ref %.3e  frozen vs ref %.3e" % (
        rel_err(g1[i:i + 2], g0[i:i + 2]), rel_err(g1[i:i+2], ctx_ref.grad[i:i+2]), rel_err(g0[i:i+2], ctx_ref.grad[i:i+2])))

"""ORACLE -- test infrastructure, not product code.

A plain-PyTorch fp32 restatement of the reference's SD-1.5 hot path (askerlee/adaprompt,
mounted read-only at /root/reference).  It exists so that the HIP path can be checked on a
box where the reference itself cannot travel.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
``adaprompt_amd`` never does and fails loudly when its HIP extension is missing.

Pinning: every function below is checked against outputs of the reference's own modules
(``UNetModel``, ``CrossAttention``, ``SpatialTransformer``, ``ResBlock``, VAE ``Encoder``,
``AttnBlock`` ...), imported from /root/reference on CPU and fed the same deterministic
synthetic state-dict; the captured vectors are committed under ``tests/golden/`` together
with the script that made them (``tests/golden/make_golden.py``).  The reference has no
tests or golden vectors of its own.  The few functions whose reference
module cannot be imported here (``ddpm.py`` needs pytorch_lightning etc.) -- schedule,
``q_sample``, ``calc_recon_loss``, posterior sample -- follow the source text at the cited
lines and are pinned by closed-form known-answer tests (tests/test_oracle_golden.py).

Everything is functional: a state dict with the reference's key names
(``model.diffusion_model.*``, ``first_stage_model.*``) plus tensors in, tensors out.
All citations are file:line into /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------


def timestep_embedding(timesteps, dim, max_period=10000):
    """ldm/modules/diffusionmodules/util.py:154-174 (repeat_only=False)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half
                      ).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x, w, b, eps):
    """GroupNorm32: 32 groups, computed in fp32 (util.py:217-219); eps 1e-5 in ResBlocks and
    ``out`` (nn.GroupNorm default), 1e-6 in SpatialTransformer.norm (attention.py:71-72) and
    in every VAE norm (model.py:39-40)."""
    return F.group_norm(x.float(), 32, w, b, eps).type(x.dtype)


def silu(x):
    """util.py:212-214 / model.py:34-36 (swish)."""
    return x * torch.sigmoid(x)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


# ---------------------------------------------------------------------------------------
# UNet blocks
# ---------------------------------------------------------------------------------------


def res_block(sd, p, x, emb):
    """ResBlock._forward, openaimodel.py:259-279 (no up/down, no scale-shift norm):
    GN32(1e-5)->SiLU->conv3x3 ; + Linear(SiLU(emb)) ; GN32->SiLU->Dropout(0)->conv3x3 ; +skip."""
    h = silu(group_norm32(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], 1e-5))
    h = _conv(sd, p + ".in_layers.2", h, padding=1)
    emb_out = _lin(sd, p + ".emb_layers.1", silu(emb)).type(h.dtype)
    h = h + emb_out[:, :, None, None]
    h = silu(group_norm32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], 1e-5))
    h = _conv(sd, p + ".out_layers.3", h, padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x)
    return x + h


def cross_attention(sd, p, x, context=None, mask=None, heads=8, save=None):
    """CrossAttention.forward, attention.py:172-257.

    ``context`` is None (self-attention), a tensor, or a (v_context, k_context) tuple
    (attention.py:188-191).  ``mask`` [B,1,h,w] masks *keys* with -finfo.max (:223-232).
    The scale dim_head**-0.5 multiplies ``sim`` after the einsum (:199).  conv-attn row
    replacement (:208-216) is off in the shipped config (use_conv_attn_kernel_size=-1,
    embedding_manager.py:967) and not restated.  When ``save`` is a dict it receives the
    side outputs of :245-255: q*scale**0.5 [B,h,N,d], attn and attnscore [B,h,N,M]."""
    h = heads
    q = _lin(sd, p + ".to_q", x)
    if context is None:
        context = x
    if isinstance(context, (list, tuple)):
        v_ctx, k_ctx = context
    else:
        v_ctx = k_ctx = context
    k = _lin(sd, p + ".to_k", k_ctx)
    v = _lin(sd, p + ".to_v", v_ctx)
    B, N, C = q.shape
    d = C // h
    scale = d ** -0.5

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], h, d).permute(0, 2, 1, 3).reshape(
            t.shape[0] * h, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * scale
    if mask is not None:
        m = mask.reshape(mask.shape[0], -1).bool()                       # [B, M]
        m = m[:, None, None, :].expand(B, h, 1, m.shape[-1]).reshape(B * h, 1, -1)
        sim = sim.masked_fill(~m, -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(B, h, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    out = _lin(sd, p + ".to_out.0", out)
    if save is not None:
        save["q"] = q.reshape(B, h, N, d) * math.sqrt(scale)
        save["attn"] = attn.reshape(B, h, N, -1)
        save["attnscore"] = sim.reshape(B, h, N, -1)
    return out


def feed_forward_geglu(sd, p, x):
    """FeedForward with GEGLU, attention.py:32-59: proj -> chunk(2) -> x*gelu(gate) -> Linear."""
    a, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


def basic_transformer_block(sd, p, x, context, mask, heads, save=None):
    """BasicTransformerBlock._forward, attention.py:275-285."""
    def ln(n, t):
        return F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)
    x1 = cross_attention(sd, p + ".attn1", ln("norm1", x), None, mask, heads) + x
    x2 = x1 + cross_attention(sd, p + ".attn2", ln("norm2", x1), context, None, heads, save)
    return feed_forward_geglu(sd, p + ".ff", ln("norm3", x2)) + x2


def spatial_transformer(sd, p, x, context, mask=None, heads=8, save=None):
    """SpatialTransformer.forward, attention.py:321-341: GN(1e-6) -> proj_in 1x1 ->
    'b c h w -> b (h w) c' -> block -> back -> proj_out 1x1 -> + x_in.  ``mask`` is resized
    to the level's (h, w) with mode='nearest' (:332)."""
    b, c, hh, ww = x.shape
    x_in = x
    x = group_norm32(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    x = _conv(sd, p + ".proj_in", x)
    x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    mask2 = F.interpolate(mask, size=(hh, ww), mode="nearest") if mask is not None else None
    x = basic_transformer_block(sd, p + ".transformer_blocks.0", x, context, mask2, heads, save)
    x = x.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    x = _conv(sd, p + ".proj_out", x)
    return x + x_in


# layer_idx -> index into the 16-way layerwise context (openaimodel.py:876-877)
LAYER2CA = {1: 0, 2: 1, 4: 2, 5: 3, 7: 4, 8: 5, 12: 6, 16: 7, 17: 8, 18: 9, 19: 10, 20: 11,
            21: 12, 22: 13, 23: 14, 24: 15}
# layers whose cross-attention activations are captured (openaimodel.py:949)
DISTILL_LAYERS = (7, 8, 12, 16, 17, 18, 19, 20, 21, 22, 23, 24)


def unet_forward(sd, cfg, x, timesteps, context, extra_info, prefix="model.diffusion_model."):
    """UNetModel.forward, openaimodel.py:827-1052, for the layerwise-context path (the only
    one that works in the reference).

    ``context`` [16*B, M, Cctx] with the 16 layers of an instance contiguous ->
    reshape(B,16,M,C).permute(1,0,2,3) (:866).  ``iter_type == 'mix_hijk'`` splits each
    layer context into (V half, K half) along tokens (:885-891).  Writes
    ``extra_info['ca_layers_activations']`` = {outfeat, attn, attnscore, q}->{layer: tensor}
    (:1031-1035); with ``capture_distill_attn`` false those dicts are empty."""
    P = prefix
    mc = cfg["model_channels"]
    heads = cfg["num_heads"]
    mult = tuple(cfg["channel_mult"])
    nres = cfg["num_res_blocks"]
    attn_res = tuple(cfg["attention_resolutions"])
    assert extra_info is not None and extra_info.get("use_layerwise_context", False)
    iter_type = extra_info.get("iter_type", "normal_recon")
    capture = extra_info.get("capture_distill_attn", False)
    img_mask = extra_info.get("img_mask", None)
    B = x.shape[0]

    emb = timestep_embedding(timesteps, mc)
    emb = _lin(sd, P + "time_embed.2", silu(_lin(sd, P + "time_embed.0", emb)))
    context = context.reshape(B, 16, -1, context.shape[-1]).permute(1, 0, 2, 3)

    def layer_context(layer_idx):
        c = context[LAYER2CA[layer_idx]]
        if iter_type == "mix_hijk":
            v, k = c.chunk(2, dim=1)
            return (v, k)
        return (c, c)

    acts = {}

    def run_st(p, h, layer_idx):
        save = {} if (capture and layer_idx in DISTILL_LAYERS) else None
        h = spatial_transformer(sd, p, h, layer_context(layer_idx), img_mask, heads, save)
        if save is not None:
            acts[layer_idx] = save
        return h

    hs = []
    h = _conv(sd, P + "input_blocks.0.0", x, padding=1)
    hs.append(h)
    layer_idx, ds, idx = 1, 1, 1
    for level in range(len(mult)):
        for _ in range(nres):
            h = res_block(sd, f"{P}input_blocks.{idx}.0", h, emb)
            if ds in attn_res:
                h = run_st(f"{P}input_blocks.{idx}.1", h, layer_idx)
            if layer_idx in acts:
                acts[layer_idx]["outfeat"] = h
            hs.append(h)
            idx += 1
            layer_idx += 1
        if level != len(mult) - 1:
            h = _conv(sd, f"{P}input_blocks.{idx}.0.op", h, stride=2, padding=1)   # :138-164
            hs.append(h)
            idx += 1
            layer_idx += 1
            ds *= 2
    h = res_block(sd, P + "middle_block.0", h, emb)
    h = run_st(P + "middle_block.1", h, layer_idx)
    h = res_block(sd, P + "middle_block.2", h, emb)
    if layer_idx in acts:
        acts[layer_idx]["outfeat"] = h
    layer_idx += 1
    idx = 0
    for level in reversed(range(len(mult))):
        for i in range(nres + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            h = res_block(sd, f"{P}output_blocks.{idx}.0", h, emb)
            sub = 1
            if ds in attn_res:
                h = run_st(f"{P}output_blocks.{idx}.1", h, layer_idx)
                sub = 2
            if level and i == nres:
                h = F.interpolate(h, scale_factor=2, mode="nearest")               # :95-123
                h = _conv(sd, f"{P}output_blocks.{idx}.{sub}.conv", h, padding=1)
                ds //= 2
            if layer_idx in acts:
                acts[layer_idx]["outfeat"] = h
            idx += 1
            layer_idx += 1
    extra_info["ca_layers_activations"] = {
        key: {li: acts[li][key] for li in acts} for key in ("outfeat", "attn", "attnscore", "q")}
    h = silu(group_norm32(h, sd[P + "out.0.weight"], sd[P + "out.0.bias"], 1e-5))
    return _conv(sd, P + "out.2", h, padding=1)


# ---------------------------------------------------------------------------------------
# VAE encoder (first stage)
# ---------------------------------------------------------------------------------------


def vae_resnet_block(sd, p, x):
    """ResnetBlock.forward with temb=None, model.py:122-142 (GN eps 1e-6, swish)."""
    h = silu(group_norm32(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6))
    h = _conv(sd, p + ".conv1", h, padding=1)
    h = silu(group_norm32(h, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6))
    h = _conv(sd, p + ".conv2", h, padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def vae_downsample(sd, p, x):
    """Downsample.forward, model.py:73-77: pad (0,1,0,1) then conv3x3 stride 2 pad 0."""
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    return _conv(sd, p + ".conv", x, stride=2, padding=0)


def vae_attn_block(sd, p, x, mask=None):
    """AttnBlock.forward, model.py:179-242: single head, scale C**-0.5, softmax over keys,
    then -- when fg/aug masks are given -- POST-softmax zero-fill of every (query, key) pair
    that is not fg-fg or bg-bg (:196-232); no renormalisation."""
    h_ = group_norm32(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    q, k, v = _conv(sd, p + ".q", h_), _conv(sd, p + ".k", h_), _conv(sd, p + ".v", h_)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    if mask is not None:
        aug = mask["aug_mask"].type(x.dtype) if mask.get("aug_mask") is not None \
            else torch.ones_like(x[:, [0]])
        fg = mask["fg_mask"].type(x.dtype) if mask.get("fg_mask") is not None else None
        if fg is not None:
            fg = F.interpolate(fg, size=x.shape[-2:], mode="nearest")
            bg = 1 - fg
            aug = F.interpolate(aug, size=x.shape[-2:], mode="nearest")
            fg, bg = fg * aug, bg * aug
            fg2, bg2 = fg.reshape(b, 1, -1), bg.reshape(b, 1, -1)
            fg_pair = torch.matmul(fg2.transpose(-1, -2), fg2).bool()
            bg_pair = torch.matmul(bg2.transpose(-1, -2), bg2).bool()
            w_ = w_.masked_fill(~(fg_pair | bg_pair), 0)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h_)


def vae_encoder_forward(sd, dd, x, mask=None, prefix="first_stage_model.encoder."):
    """Encoder.forward, model.py:474-499 (attn_resolutions=[] -> attention only in mid)."""
    P = prefix
    mult = tuple(dd["ch_mult"])
    h = _conv(sd, P + "conv_in", x, padding=1)
    for lvl in range(len(mult)):
        for b in range(dd["num_res_blocks"]):
            h = vae_resnet_block(sd, f"{P}down.{lvl}.block.{b}", h)
        if lvl != len(mult) - 1:
            h = vae_downsample(sd, f"{P}down.{lvl}.downsample", h)
    h = vae_resnet_block(sd, P + "mid.block_1", h)
    h = vae_attn_block(sd, P + "mid.attn_1", h, mask)
    h = vae_resnet_block(sd, P + "mid.block_2", h)
    h = silu(group_norm32(h, sd[P + "norm_out.weight"], sd[P + "norm_out.bias"], 1e-6))
    return _conv(sd, P + "conv_out", h, padding=1)


def autoencoder_encode_moments(sd, dd, x, mask=None, prefix="first_stage_model."):
    """AutoencoderKL.encode, autoencoder.py:324-328, up to the posterior parameters."""
    h = vae_encoder_forward(sd, dd, x, mask, prefix + "encoder.")
    return _conv(sd, prefix + "quant_conv", h)


def vae_upsample(sd, p, x):
    """Upsample.forward, model.py:52-58: nearest x2 then conv3x3 pad 1."""
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return _conv(sd, p + ".conv", x, padding=1)


def vae_decoder_forward(sd, dd, z, prefix="first_stage_model.decoder."):
    """Decoder.forward, model.py:575-608 (attn_resolutions=[] -> attention only in mid, give_pre_end/tanh off)."""
    P = prefix
    mult = tuple(dd["ch_mult"])
    h = _conv(sd, P + "conv_in", z, padding=1)
    h = vae_resnet_block(sd, P + "mid.block_1", h)
    h = vae_attn_block(sd, P + "mid.attn_1", h)
    h = vae_resnet_block(sd, P + "mid.block_2", h)
    for lvl in reversed(range(len(mult))):
        for b in range(dd["num_res_blocks"] + 1):
            h = vae_resnet_block(sd, f"{P}up.{lvl}.block.{b}", h)
        if lvl != 0:
            h = vae_upsample(sd, f"{P}up.{lvl}.upsample", h)
    h = silu(group_norm32(h, sd[P + "norm_out.weight"], sd[P + "norm_out.bias"], 1e-6))
    return _conv(sd, P + "conv_out", h, padding=1)


def autoencoder_decode(sd, dd, z, prefix="first_stage_model."):
    """AutoencoderKL.decode, autoencoder.py:330-333: post_quant_conv then the decoder."""
    return vae_decoder_forward(sd, dd, _conv(sd, prefix + "post_quant_conv", z), prefix + "decoder.")


def decode_first_stage(sd, dd, z, scale_factor=0.18215, prefix="first_stage_model."):
    """LatentDiffusion.decode_first_stage, ddpm.py:1260-1267 (+ the un-tiled tail): z / scale_factor, decode."""
    return autoencoder_decode(sd, dd, z / scale_factor, prefix)


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution, distributions.py:24-37, with the noise supplied:
    chunk(2,1); clamp(logvar,-30,20); mean + exp(0.5*logvar)*noise."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def get_first_stage_encoding(moments, noise, scale_factor=0.18215):
    """ddpm.py:955-962: scale_factor * posterior.sample()."""
    return scale_factor * gaussian_sample(moments, noise)


# ---------------------------------------------------------------------------------------
# diffusion schedule / loss (ddpm.py cannot be imported here: restated from source text)
# ---------------------------------------------------------------------------------------


def make_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.012):
    """make_beta_schedule('linear') util.py:21-26 + register_schedule ddpm.py:240-292:
    betas = linspace(sqrt(ls), sqrt(le), T, fp64)**2; cumprod in fp64; buffers cast to fp32."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps,
                            dtype=torch.float64) ** 2).numpy()
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)

    def f32(a):
        return torch.tensor(a, dtype=torch.float32)
    return {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
    }


def _extract(a, t, x):
    """extract_into_tensor, util.py:99-102."""
    return a.to(x.device).gather(-1, t).reshape(t.shape[0], *((1,) * (x.dim() - 1)))


def q_sample(sched, x_start, t, noise):
    """ddpm.py:416-419."""
    return (_extract(sched["sqrt_alphas_cumprod"], t, x_start) * x_start
            + _extract(sched["sqrt_one_minus_alphas_cumprod"], t, x_start) * noise)


def predict_start_from_noise(sched, x_t, t, noise):
    """ddpm.py:358-362."""
    return (_extract(sched["sqrt_recip_alphas_cumprod"], t, x_t) * x_t
            - _extract(sched["sqrt_recipm1_alphas_cumprod"], t, x_t) * noise)


def calc_recon_loss(model_output, target, img_mask=None, fg_mask=None,
                    fg_pixel_weight=1.0, bg_pixel_weight=1.0):
    """ddpm.py:3571-3595 with get_loss('l2', mean=False) ddpm.py:422-438."""
    if img_mask is None:
        img_mask = torch.ones_like(model_output)
    if fg_mask is None:
        fg_mask = torch.ones_like(model_output)
    mo = model_output * img_mask
    tg = target * img_mask
    pix = F.mse_loss(tg, mo, reduction="none")
    wfg = (fg_mask * img_mask * fg_pixel_weight).expand_as(pix)
    wbg = ((1 - fg_mask) * img_mask * bg_pixel_weight).expand_as(pix)
    loss = ((pix * wfg).sum() + (pix * wbg).sum()) / (wfg.sum() + wbg.sum() + 1e-6)
    return loss, pix


# ---------------------------------------------------------------------------------------
# one pure-recon micro-batch (the bench "step"), used as the CPU baseline and in smoke()
# ---------------------------------------------------------------------------------------


def recon_step(unet_sd, vae_sd, unet_cfg, vae_dd, image_nchw, masks, post_noise, t, noise,
               context, img_mask, fg_mask, bg_pixel_weight=0.1, need_grad=True):
    """VAE encode (no grad) -> q_sample -> UNet eps-pred -> masked MSE -> d loss / d context.
    Follows get_input ddpm.py:1178-1256, guided_denoise :2483-2532, calc_recon_loss :3571."""
    sched = make_schedule()
    with torch.no_grad():
        moments = autoencoder_encode_moments(vae_sd, vae_dd, image_nchw, masks)
        z = get_first_stage_encoding(moments, post_noise)
        x_noisy = q_sample(sched, z, t, noise)
    ctx = context.detach().clone().requires_grad_(need_grad)
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1,
             "iter_type": "normal_recon", "is_training": True, "capture_distill_attn": False,
             "img_mask": img_mask}          # ddpm.py:2876: the image mask also gates the self-attention keys
    eps_hat = unet_forward(unet_sd, unet_cfg, x_noisy, t, ctx, extra)
    loss, _ = calc_recon_loss(eps_hat, noise, img_mask, fg_mask, 1.0, bg_pixel_weight)
    grad = None
    if need_grad:
        (grad,) = torch.autograd.grad(loss, ctx)
    return {"z": z, "x_noisy": x_noisy, "eps_hat": eps_hat.detach(), "loss": loss.detach(),
            "grad_context": grad}

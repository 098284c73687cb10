"""Deterministic synthetic weights and inputs for the SD-1.5 hot path.

No pretrained checkpoint is available offline, so parity fixtures and the bench use a
state-dict that is a pure function of (tensor name, shape, seed).  The same function
fills the reference modules when the golden vectors are captured
(tests/golden/make_golden.py) and the HIP path on the GPU box, so only inputs' seeds
and outputs have to be committed.

The reference zero-initialises 39 tensors (``zero_module``: ResBlock out conv
openaimodel.py:233, ``proj_out`` attention.py:313, final conv openaimodel.py:696); a
fresh-init fixture would therefore be identically zero.  Here
every tensor, including those, gets non-zero values.

Key namespace follows the reference checkpoints (ddpm.py:321-344):
``model.diffusion_model.*`` for the UNet, ``first_stage_model.*`` for the VAE.
"""
import zlib

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synthetic_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """fp32 CPU tensor for state-dict entry ``name``.

    * >=2-D weights: N(0, gain/fan_in) -- keeps activations O(1) through the 25 blocks.
    * 1-D ``weight`` (norm scales): 1 + 0.1*N(0,1).
    * 1-D ``bias``: 0.05*N(0,1).
    """
    shape = tuple(int(s) for s in shape)
    g = _gen(name, seed)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g, dtype=torch.float32) * (0.8 / fan_in ** 0.5)
    if name.endswith("weight"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return 0.05 * torch.randn(shape, generator=g, dtype=torch.float32)


def synthetic_like(named_shapes, seed: int = 0, prefix: str = ""):
    """State dict for an iterable of (name, shape)."""
    return {prefix + n: synthetic_tensor(prefix + n, s, seed) for n, s in named_shapes}


def synthetic_input(tag: str, shape, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    g = _gen("input:" + tag, seed)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale


# --------------------------------------------------------------------------------------
# Shapes of the ldm state dict, derived from the config alone (no module construction).
# --------------------------------------------------------------------------------------

def unet_param_shapes(model_channels=320, in_channels=4, out_channels=4, num_res_blocks=2,
                      attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4),
                      context_dim=768, **_unused):
    """(name, shape) for every tensor of ``UNetModel`` built with the yaml params
    (configs/stable-diffusion/v1-finetune-ada.yaml:107-122; openaimodel.py:447-703)."""
    mc = model_channels
    ted = 4 * mc
    out = []

    def lin(p, i, o, bias=True):
        out.append((p + ".weight", (o, i)))
        if bias:
            out.append((p + ".bias", (o,)))

    def conv(p, i, o, k):
        out.append((p + ".weight", (o, i, k, k)))
        out.append((p + ".bias", (o,)))

    def norm(p, c):
        out.append((p + ".weight", (c,)))
        out.append((p + ".bias", (c,)))

    def res(p, ci, co):
        norm(p + ".in_layers.0", ci)
        conv(p + ".in_layers.2", ci, co, 3)
        lin(p + ".emb_layers.1", ted, co)
        norm(p + ".out_layers.0", co)
        conv(p + ".out_layers.3", co, co, 3)
        if ci != co:
            conv(p + ".skip_connection", ci, co, 1)

    def st(p, c):
        norm(p + ".norm", c)
        conv(p + ".proj_in", c, c, 1)
        b = p + ".transformer_blocks.0"
        for a, cd in (("attn1", c), ("attn2", context_dim)):
            lin(f"{b}.{a}.to_q", c, c, bias=False)
            lin(f"{b}.{a}.to_k", cd, c, bias=False)
            lin(f"{b}.{a}.to_v", cd, c, bias=False)
            lin(f"{b}.{a}.to_out.0", c, c)
        lin(f"{b}.ff.net.0.proj", c, 8 * c)
        lin(f"{b}.ff.net.2", 4 * c, c)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{b}.{n}", c)
        conv(p + ".proj_out", c, c, 1)

    lin("time_embed.0", mc, ted)
    lin("time_embed.2", ted, ted)
    conv("input_blocks.0.0", in_channels, mc, 3)
    chans = [mc]
    ch, ds, idx = mc, 1, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            res(f"input_blocks.{idx}.0", ch, mult * mc)
            ch = mult * mc
            if ds in attention_resolutions:
                st(f"input_blocks.{idx}.1", ch)
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            conv(f"input_blocks.{idx}.0.op", ch, ch, 3)
            chans.append(ch)
            idx += 1
            ds *= 2
    res("middle_block.0", ch, ch)
    st("middle_block.1", ch)
    res("middle_block.2", ch, ch)
    idx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            res(f"output_blocks.{idx}.0", ch + ich, mc * mult)
            ch = mc * mult
            sub = 1
            if ds in attention_resolutions:
                st(f"output_blocks.{idx}.1", ch)
                sub = 2
            if level and i == num_res_blocks:
                conv(f"output_blocks.{idx}.{sub}.conv", ch, ch, 3)
                ds //= 2
            idx += 1
    norm("out.0", ch)
    conv("out.2", mc, out_channels, 3)
    return out


def vae_encoder_param_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3,
                             z_channels=4, embed_dim=4, double_z=True, **_unused):
    """(name, shape) for ``AutoencoderKL.encoder`` + ``quant_conv``
    (model.py:408-472; autoencoder.py:297-303)."""
    out = []

    def conv(p, i, o, k):
        out.append((p + ".weight", (o, i, k, k)))
        out.append((p + ".bias", (o,)))

    def norm(p, c):
        out.append((p + ".weight", (c,)))
        out.append((p + ".bias", (c,)))

    def res(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", ci, co, 3)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", ci, co, 1)

    conv("encoder.conv_in", in_channels, ch, 3)
    in_mult = (1,) + tuple(ch_mult)
    bi = ch
    for lvl in range(len(ch_mult)):
        bi = ch * in_mult[lvl]
        bo = ch * ch_mult[lvl]
        for b in range(num_res_blocks):
            res(f"encoder.down.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != len(ch_mult) - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", bi, bi, 3)
    res("encoder.mid.block_1", bi, bi)
    norm("encoder.mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"encoder.mid.attn_1.{n}", bi, bi, 1)
    res("encoder.mid.block_2", bi, bi)
    norm("encoder.norm_out", bi)
    conv("encoder.conv_out", bi, 2 * z_channels if double_z else z_channels, 3)
    conv("quant_conv", 2 * z_channels, 2 * embed_dim, 1)
    return out


def vae_decoder_param_shapes(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, embed_dim=4,
                             **_unused):
    """(name, shape) for ``AutoencoderKL.decoder`` + ``post_quant_conv`` (model.py:502-573; autoencoder.py:304)."""
    out = []

    def conv(p, i, o, k):
        out.append((p + ".weight", (o, i, k, k)))
        out.append((p + ".bias", (o,)))

    def norm(p, c):
        out.append((p + ".weight", (c,)))
        out.append((p + ".bias", (c,)))

    def res(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", ci, co, 3)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", ci, co, 1)

    nres = len(ch_mult)
    bi = ch * ch_mult[nres - 1]
    conv("decoder.conv_in", z_channels, bi, 3)
    res("decoder.mid.block_1", bi, bi)
    norm("decoder.mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"decoder.mid.attn_1.{n}", bi, bi, 1)
    res("decoder.mid.block_2", bi, bi)
    for lvl in reversed(range(nres)):
        bo = ch * ch_mult[lvl]
        for b in range(num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", bi, bi, 3)
    norm("decoder.norm_out", bi)
    conv("decoder.conv_out", bi, out_ch, 3)
    conv("post_quant_conv", embed_dim, z_channels, 1)
    return out


SD15_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320,
                 attention_resolutions=(4, 2, 1), num_res_blocks=2, channel_mult=(1, 2, 4, 4),
                 num_heads=8, use_spatial_transformer=True, transformer_depth=1,
                 context_dim=768, use_checkpoint=True, legacy=False)

SD15_VAE_DD = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128,
                   ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=(), dropout=0.0)


def synthetic_unet_state_dict(cfg=None, seed=0, prefix="model.diffusion_model."):
    cfg = dict(SD15_UNET if cfg is None else cfg)
    return synthetic_like(unet_param_shapes(**cfg), seed, prefix)


def synthetic_vae_state_dict(ddconfig=None, embed_dim=4, seed=0, prefix="first_stage_model.", decoder=False):
    dd = dict(SD15_VAE_DD if ddconfig is None else ddconfig)
    shapes = vae_encoder_param_shapes(embed_dim=embed_dim, **dd)
    if decoder:
        shapes = shapes + vae_decoder_param_shapes(embed_dim=embed_dim, **dd)
    return synthetic_like(shapes, seed, prefix)


def clip_vision_param_shapes(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, image_size=224, patch_size=14,
                             **_unused):
    """(name, shape) of HF ``CLIPVisionModel`` (transformers modeling_clip.py: CLIPVisionEmbeddings / CLIPEncoderLayer),
    with the ``vision_model.`` prefix transformers < 5 uses (the reference's checkpoints, ddpm.py:904-914)."""
    H, I = hidden_size, intermediate_size
    P = "vision_model."
    out = [(P + "embeddings.class_embedding", (H,)), (P + "embeddings.patch_embedding.weight", (H, 3, patch_size, patch_size)),
           (P + "embeddings.position_embedding.weight", ((image_size // patch_size) ** 2 + 1, H)),
           (P + "pre_layrnorm.weight", (H,)), (P + "pre_layrnorm.bias", (H,))]
    for i in range(num_hidden_layers):
        L = f"{P}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(L + f"self_attn.{n}.weight", (H, H)), (L + f"self_attn.{n}.bias", (H,))]
        out += [(L + "layer_norm1.weight", (H,)), (L + "layer_norm1.bias", (H,)),
                (L + "mlp.fc1.weight", (I, H)), (L + "mlp.fc1.bias", (I,)), (L + "mlp.fc2.weight", (H, I)), (L + "mlp.fc2.bias", (H,)),
                (L + "layer_norm2.weight", (H,)), (L + "layer_norm2.bias", (H,))]
    out += [(P + "post_layernorm.weight", (H,)), (P + "post_layernorm.bias", (H,))]
    return out


def synthetic_clip_vision_state_dict(cfg, seed=0):
    sd = synthetic_like(clip_vision_param_shapes(**cfg), seed)
    # the class token and the position table at the scale of the patch tokens (a 1-D "embedding" is not a bias)
    sd["vision_model.embeddings.class_embedding"] = sd["vision_model.embeddings.class_embedding"] * 10.0
    sd["vision_model.embeddings.position_embedding.weight"] = sd["vision_model.embeddings.position_embedding.weight"] * 10.0
    return sd

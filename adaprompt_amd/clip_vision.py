"""The CLIP vision encoder of the zero-shot feature front end on the MI355X kernels.

The reference loads HF ``CLIPVisionModel`` ("openai/clip-vit-large-patch14": 24 layers, width 1024, 16 heads of 64, MLP 4096,
quick_gelu, 14x14 patches of a 224x224 image -> 257 tokens; ddpm.py:904-914) into its own subclass
``CLIPVisionModelWithMask`` (adaface/subj_basis_generator.py:664-757), whose forward takes a foreground mask: the mask is
resized to the 16x16 patch grid (nearest), a 1 is prepended for the class token, and the outer product m m^T [B,1,257,257] is
handed to the encoder as ``attention_mask`` -- which HF's ``CLIPAttention`` ADDS to the scaled scores.  So the "mask" is an
additive rank-1 bias: +1 on the logit of every (foreground, foreground) pair, nothing masked out.  It is called twice per
image (mask, 1 - mask) without gradient in every zero-shot training iteration (ddpm.py:2415-2431) and the features are the
hidden states BEFORE the last layer (``hidden_states[-2]``): 2 x 0.16 TFLOP per image.

Here: same module tree and parameter names as HF's (so ``CLIPVisionModel.state_dict()`` loads, with or without the
``vision_model.`` prefix that transformers < 5 has), inference only.  Per layer: LayerNorm -> bf16, ONE fused q|k|v
contraction, flash attention, out-projection with the residual in its epilogue, LayerNorm, fc1, activation kernel, fc2 with
the residual.  The rank-1 bias costs no extra kernel: q gets an extra column m_i / scale and k an extra column m_j (head
dim 64 -> 72, zero-padded in v), so scale * q'.k' = scale * q.k + m_i m_j comes out of the same MFMA product.  The patch
embedding (a 14x14 stride-14 conv without bias) is one contraction over the unfolded patches."""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .functional import WeightCache

F32, BF16 = torch.float32, torch.bfloat16


class _Embeddings(nn.Module):
    def __init__(self, width, image_size, patch_size):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(width))
        self.patch_embedding = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        self.num_patches = (image_size // patch_size) ** 2
        self.position_embedding = nn.Embedding(self.num_patches + 1, width)


class _Attention(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.k_proj, self.v_proj = nn.Linear(width, width), nn.Linear(width, width)
        self.q_proj, self.out_proj = nn.Linear(width, width), nn.Linear(width, width)


class _MLP(nn.Module):
    def __init__(self, width, inner):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(width, inner), nn.Linear(inner, width)


class _Layer(nn.Module):
    def __init__(self, width, inner, eps):
        super().__init__()
        self.self_attn = _Attention(width)
        self.layer_norm1 = nn.LayerNorm(width, eps=eps)
        self.mlp = _MLP(width, inner)
        self.layer_norm2 = nn.LayerNorm(width, eps=eps)


class _Encoder(nn.Module):
    def __init__(self, n, width, inner, eps):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(width, inner, eps) for _ in range(n)])


class _VisionTransformer(nn.Module):
    def __init__(self, n, width, inner, image_size, patch_size, eps):
        super().__init__()
        self.embeddings = _Embeddings(width, image_size, patch_size)
        self.pre_layrnorm = nn.LayerNorm(width, eps=eps)                  # (sic: HF's attribute name)
        self.encoder = _Encoder(n, width, inner, eps)
        self.post_layernorm = nn.LayerNorm(width, eps=eps)


class CLIPVisionModelWithMask(nn.Module):
    """``forward(pixel_values, attn_mask=None, output_hidden_states=True)`` -> an object with ``last_hidden_state``,
    ``pooler_output``, ``hidden_states`` (tuple: embeddings output + one per layer) and ``attn_mask`` [B,257,1], as the
    reference's class returns (subj_basis_generator.py:727-737).  ``layers_needed``: stop after that many layers
    (the front end reads ``hidden_states[-2]`` only: ``num_hidden_layers - 1`` skips the last layer's work)."""

    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                 patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, **unused):
        super().__init__()
        assert hidden_act in ("quick_gelu", "gelu"), hidden_act
        assert hidden_size % num_attention_heads == 0 and (hidden_size // num_attention_heads) % 8 == 0
        self.config = SimpleNamespace(hidden_size=hidden_size, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      image_size=image_size, patch_size=patch_size, hidden_act=hidden_act,
                                      layer_norm_eps=layer_norm_eps)
        self.vision_model = _VisionTransformer(num_hidden_layers, hidden_size, intermediate_size, image_size, patch_size,
                                               layer_norm_eps)
        self._packs = WeightCache()
        self.requires_grad_(False)

    @property
    def stop_before_last_layer(self):
        """the ``layers_needed`` that makes ``hidden_states[-1]`` the full model's ``hidden_states[-2]``"""
        return self.config.num_hidden_layers - 1

    @classmethod
    def from_config(cls, config):
        """``config``: a HF ``CLIPVisionConfig`` (or anything with its attribute names)."""
        keys = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "image_size", "patch_size",
                "hidden_act", "layer_norm_eps")
        return cls(**{k: getattr(config, k) for k in keys})

    def load_hf_state_dict(self, sd, strict=True):
        """a HF ``CLIPVisionModel`` state dict; transformers >= 5 dropped the ``vision_model.`` prefix, both forms load."""
        sd = {(k if k.startswith("vision_model.") else "vision_model." + k): v for k, v in sd.items()
              if not k.endswith("position_ids")}
        out = self.load_state_dict(sd, strict=strict)
        self._packs.clear()
        return out

    # ------------------------------------------------------------------------------------------------------------------
    def _embed(self, pixel_values):
        cfg, emb = self.config, self.vision_model.embeddings
        B, P = pixel_values.shape[0], cfg.patch_size
        g = cfg.image_size // P
        assert tuple(pixel_values.shape[1:]) == (3, cfg.image_size, cfg.image_size), pixel_values.shape
        # [B,3,g,P,g,P] -> [B, g*g, 3*P*P]: the conv's receptive fields as rows, in the conv weight's (c, ky, kx) order
        patches = pixel_values.float().view(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
        pk = self._packs.get("patch", emb.patch_embedding.weight.view(cfg.hidden_size, 3 * P * P))
        x16 = ops.pad_cast_bf16(patches, pk.I8)
        tok, _ = ops.linear(x16, pk.fwd, pk.O4)
        h = torch.cat([emb.class_embedding.view(1, 1, -1).expand(B, 1, -1), tok[..., :cfg.hidden_size]], dim=1)
        h = h + emb.position_embedding.weight[None]
        ln = self.vision_model.pre_layrnorm
        return torch.nn.functional.layer_norm(h, (cfg.hidden_size,), ln.weight, ln.bias, ln.eps).contiguous()

    @staticmethod
    def token_mask(attn_mask, grid):
        """[B,H,W] image-space mask -> [B,1,1+grid*grid]: nearest resize to the patch grid, 1 for the class token
        (subj_basis_generator.py:693-702)."""
        m = torch.nn.functional.interpolate(attn_mask.unsqueeze(1).float(), size=(grid, grid), mode="nearest").flatten(2)
        return torch.cat([torch.ones_like(m[:, :, :1]), m], dim=-1)

    def _layer(self, idx, lyr, h, tm):
        cfg = self.config
        B, N, C = h.shape
        heads = cfg.num_attention_heads
        d = C // heads
        at = lyr.self_attn
        n1, _, _ = ops.layernorm_fwd(h, lyr.layer_norm1.weight, lyr.layer_norm1.bias, lyr.layer_norm1.eps)
        qkv = self._packs.get(("qkv", idx), [at.q_proj.weight, at.k_proj.weight, at.v_proj.weight],
                              [at.q_proj.bias, at.k_proj.bias, at.v_proj.bias])
        _, t = ops.linear(n1, qkv.fwd, 3 * C, bias=qkv.bias, out_f32=False, out_bf16=True)
        q, k, v = t[..., :C], t[..., C:2 * C], t[..., 2 * C:]
        if tm is None:
            o, _ = ops.attention_fwd(q, k, v, heads)
        else:
            # rank-1 additive bias m_i m_j through one more head-dim column (8 for alignment): q' = [q | m_i/scale, 0..],
            # k' = [k | m_j, 0..], v' = [v | 0..] -> scale * q'.k' = scale * q.k + m_i m_j
            scale = float(d) ** -0.5
            ext = torch.zeros(B, N, heads, 8, device=h.device, dtype=BF16)
            ext[..., 0] = tm.view(B, N, 1).to(BF16)
            qx = torch.cat([q.reshape(B, N, heads, d), ext * (1.0 / scale)], dim=-1).view(B, N, heads * (d + 8))
            kx = torch.cat([k.reshape(B, N, heads, d), ext], dim=-1).view(B, N, heads * (d + 8))
            vx = torch.cat([v.reshape(B, N, heads, d), torch.zeros_like(ext)], dim=-1).view(B, N, heads * (d + 8))
            ox, _ = ops.attention_fwd(qx, kx, vx, heads, scale=scale)
            o = ox.view(B, N, heads, d + 8)[..., :d].reshape(B, N, C)
        po = self._packs.get(("out", idx), at.out_proj.weight, at.out_proj.bias)
        h1, _ = ops.linear(o, po.fwd, C, bias=po.bias, residual=h)
        n2, _, _ = ops.layernorm_fwd(h1, lyr.layer_norm2.weight, lyr.layer_norm2.bias, lyr.layer_norm2.eps)
        f1 = self._packs.get(("fc1", idx), lyr.mlp.fc1.weight, lyr.mlp.fc1.bias)
        f2 = self._packs.get(("fc2", idx), lyr.mlp.fc2.weight, lyr.mlp.fc2.bias)
        a32, _ = ops.linear(n2, f1.fwd, cfg.intermediate_size, bias=f1.bias)
        a16 = ops.act_fwd(a32, cfg.hidden_act)
        h2, _ = ops.linear(a16, f2.fwd, C, bias=f2.bias, residual=h1)
        return h2

    @torch.no_grad()
    def forward(self, pixel_values=None, attn_mask=None, output_attentions=None, output_hidden_states=True, return_dict=True,
                layers_needed=None):
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        if output_attentions:
            raise NotImplementedError("attention probabilities are not materialised by the flash kernel")
        if not pixel_values.is_cuda:
            raise RuntimeError("CLIPVisionModelWithMask runs on the MI355X only (no CPU fallback)")
        cfg = self.config
        h = self._embed(pixel_values)
        tm = None
        if attn_mask is not None:
            tm = self.token_mask(attn_mask, int(math.isqrt(h.shape[1] - 1)))
        n = cfg.num_hidden_layers if layers_needed is None else int(layers_needed)
        hidden = [h]
        for idx, lyr in enumerate(self.vision_model.encoder.layers[:n]):
            h = self._layer(idx, lyr, h, None if tm is None else tm[:, 0])
            hidden.append(h)
        pl = self.vision_model.post_layernorm
        pooled = torch.nn.functional.layer_norm(h[:, 0], (cfg.hidden_size,), pl.weight, pl.bias, pl.eps)
        return SimpleNamespace(last_hidden_state=h, pooler_output=pooled,
                               hidden_states=tuple(hidden) if output_hidden_states else None, attentions=None,
                               attn_mask=None if tm is None else tm.permute(0, 2, 1))

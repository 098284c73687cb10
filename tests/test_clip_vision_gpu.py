"""The zero-shot front end's image encoder on the MI355X against the vectors from the transformers package's CLIP blocks
(bf16 matrix-core operands: tolerances at that level), the rank-1 additive attention bias through the extra head-dim column,
and the activation kernel."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from adaprompt_amd import ops, synth          # noqa: E402
from conftest import rel_err          # noqa: E402
import make_golden_clip_vision as G          # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("kind", ["quick_gelu", "gelu"])
def test_activation_kernel(kind):
    x = torch.randn(3, 257, 512, device=DEV) * 2
    ref = x * torch.sigmoid(1.702 * x) if kind == "quick_gelu" else F.gelu(x)
    got = ops.act_fwd(x, kind)
    assert got.dtype == torch.bfloat16 and rel_err(got.float(), ref) < 3e-3


@pytest.mark.parametrize("heads,d", [(4, 32), (16, 64)])
def test_rank1_additive_bias_through_an_extra_head_column(heads, d):
    """softmax(scale q.k + m_i m_j) v for N = 257 tokens (not a multiple of any tile) vs torch."""
    B, N = 2, 257
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, N, heads, d, generator=g).to(torch.bfloat16).to(DEV) for _ in range(3))
    m = torch.rand(B, N, generator=g).to(DEV)
    m = torch.where(m > 0.6, torch.ones_like(m), torch.where(m < 0.3, torch.zeros_like(m), m)).to(torch.bfloat16)
    scale = d ** -0.5
    ext = torch.zeros(B, N, heads, 8, device=DEV, dtype=torch.bfloat16)
    ext[..., 0] = m.view(B, N, 1)
    qx = torch.cat([q, ext / scale], -1).view(B, N, -1)
    kx = torch.cat([k, ext], -1).view(B, N, -1)
    vx = torch.cat([v, torch.zeros_like(ext)], -1).view(B, N, -1)
    out, _ = ops.attention_fwd(qx, kx, vx, heads, scale=scale)
    out = out.view(B, N, heads, d + 8)[..., :d].float()
    s = torch.einsum("bihd,bjhd->bhij", q.float(), k.float()) * scale + (m.float()[:, :, None] * m.float()[:, None, :])[:, None]
    ref = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1), v.float())
    assert rel_err(out, ref) < 6e-3


@pytest.mark.parametrize("name", list(G.CASES))
def test_clip_vision_with_mask_vs_transformers_blocks(name):
    from adaprompt_amd.clip_vision import CLIPVisionModelWithMask
    cfg = G.CASES[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"clip_vision_{name}.npz"))
    m = CLIPVisionModelWithMask(**cfg)
    m.load_hf_state_dict(synth.synthetic_clip_vision_state_dict(cfg))
    m = m.to(DEV)
    x, mask = G.case_inputs(name)
    errs = {}
    for tag, mk in (("masked", mask), ("invmask", 1 - mask), ("plain", None)):
        out = m(x.to(DEV), attn_mask=None if mk is None else mk.to(DEV), output_hidden_states=True)
        assert len(out.hidden_states) == cfg["num_hidden_layers"] + 1
        errs[tag] = (rel_err(G.sub(out.hidden_states[-2].cpu()), torch.from_numpy(g[f"{tag}.penultimate"])),
                     rel_err(G.sub(out.last_hidden_state.cpu()), torch.from_numpy(g[f"{tag}.last"])),
                     rel_err(out.pooler_output.cpu(), torch.from_numpy(g[f"{tag}.pooled"])))
        if mk is not None:
            assert torch.equal(out.attn_mask.permute(0, 2, 1).cpu(), torch.from_numpy(g[f"{tag}.token_mask"]))
        else:
            assert rel_err(G.sub(out.hidden_states[0].cpu()), torch.from_numpy(g["plain.embeddings"])) < 4e-3
    print(f"[clip vision {name}] rel L2 (penultimate, last, pooled):", {k: tuple(round(e, 5) for e in v) for k, v in errs.items()})
    for tag, (e2, e1, ep) in errs.items():
        assert e2 < 6e-3 and e1 < 6e-3 and ep < 6e-3, (tag, e2, e1, ep)          # measured 2.5e-3 / 2.8e-3 / 2.2e-3
    # the front end's short cut: stop before the last layer, hidden_states[-1] is then the reference's hidden_states[-2]
    out = m(x.to(DEV), attn_mask=mask.to(DEV), layers_needed=cfg["num_hidden_layers"] - 1)
    assert rel_err(G.sub(out.hidden_states[-1].cpu()), torch.from_numpy(g["masked.penultimate"])) < 1e-2


def test_zero_shot_front_end_with_the_hip_encoder_vs_oracle_encoder():
    """``encode_zero_shot_image_features`` end to end: preprocessor stand-in -> HIP image encoder (foreground pass, background
    pass, cached zero-image pass, stopping before the last layer) -> [BS,514,D] features, against the same host code driving
    the oracle encoder on the CPU."""
    import types
    import make_golden_zeroshot as Z
    from adaprompt_amd.clip_vision import CLIPVisionModelWithMask
    from adaprompt_amd.ldm.models.diffusion.conditioning import ConditioningMixin
    from oracle import clip_vision_oracle as O
    cfg = G.CASES["narrow_quick"]
    sd = synth.synthetic_clip_vision_state_dict(cfg)
    hip = CLIPVisionModelWithMask(**cfg)
    hip.load_hf_state_dict(sd)
    hip = hip.to(DEV)

    def oracle_encoder(pixel_values, attn_mask=None, output_hidden_states=True):
        out = O.clip_vision_forward(sd, cfg, pixel_values.float(), None if attn_mask is None else attn_mask.float())
        return types.SimpleNamespace(hidden_states=out["hidden_states"], attn_mask=out["attn_mask"])

    img, mask = Z.images_case(11, B=3, hw=96)
    res = []
    for dev, enc in ((DEV, hip), ("cpu", oracle_encoder)):
        obj = types.SimpleNamespace(device=torch.device(dev), clip_preprocessor=Z.FakePreprocessor(), clip_image_encoder=enc,
                                    insightface_app=Z.FakeInsightFace(), dino_encoder=None, dino_preprocess=None,
                                    neg_image_features=None, zs_image_encoders_instantiated=True)
        feats, ids, faceless = ConditioningMixin.encode_zero_shot_image_features(obj, img.to(dev), mask.to(dev),
                                                                                 image_paths=["a", "b", "c"], calc_avg=True)
        res.append((feats.cpu(), ids.cpu(), faceless))
    (fh, ih, nh), (fo, io, no) = res
    assert tuple(fh.shape) == (1, 514, cfg["hidden_size"]) and nh == no == 0
    assert torch.allclose(ih, io, atol=1e-6)
    assert rel_err(fh, fo) < 1e-2, rel_err(fh, fo)

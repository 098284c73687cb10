"""The zero-shot front end's image encoder on the CPU: the oracle restatement against the vectors
generated from the transformers package's own CLIP blocks driven in the reference's order
(tests/golden/make_golden_clip_vision.py), and the product module's parameter tree against HF's names."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from adaprompt_amd import synth          # noqa: E402
from conftest import rel_err          # noqa: E402
import make_golden_clip_vision as G          # noqa: E402


def load(name):
    return np.load(os.path.join(ROOT, "tests", "golden", f"clip_vision_{name}.npz"))


@pytest.mark.parametrize("name", list(G.CASES))
def test_oracle_matches_the_transformers_blocks_in_the_references_order(name):
    from oracle import clip_vision_oracle as O
    cfg, g = G.CASES[name], load(name)
    sd = synth.synthetic_clip_vision_state_dict(cfg)
    x, mask = G.case_inputs(name)
    assert np.allclose(x.flatten()[::3001].numpy(), g["pixels_sample"]) and np.allclose(mask.flatten()[::997].numpy(), g["mask_sample"])
    for tag, m in (("masked", mask), ("invmask", 1 - mask), ("plain", None)):
        with torch.no_grad():
            out = O.clip_vision_forward(sd, cfg, x, m)
        assert rel_err(G.sub(out["hidden_states"][-2]), torch.from_numpy(g[f"{tag}.penultimate"])) < 2e-5, (name, tag)
        assert rel_err(G.sub(out["last_hidden_state"]), torch.from_numpy(g[f"{tag}.last"])) < 2e-5
        assert rel_err(out["pooler_output"], torch.from_numpy(g[f"{tag}.pooled"])) < 2e-5
        if m is not None:
            assert torch.equal(out["attn_mask"].permute(0, 2, 1), torch.from_numpy(g[f"{tag}.token_mask"]))
        else:
            assert rel_err(G.sub(out["hidden_states"][0]), torch.from_numpy(g["plain.embeddings"])) < 2e-5
    # the mask is an additive bias, not a hard mask: it changes the features, and by a bounded amount
    a, b = torch.from_numpy(g["masked.penultimate"]), torch.from_numpy(g["plain.penultimate"])
    assert 1e-3 < rel_err(a, b) < 0.5


def test_module_tree_has_the_hf_parameter_names():
    from adaprompt_amd.clip_vision import CLIPVisionModelWithMask
    cfg = G.CASES["narrow_quick"]
    m = CLIPVisionModelWithMask(**cfg)
    want = dict(synth.clip_vision_param_shapes(**cfg))
    have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert have == want
    assert not any(p.requires_grad for p in m.parameters())
    # both key forms load (transformers >= 5 dropped the prefix)
    sd = synth.synthetic_clip_vision_state_dict(cfg)
    m.load_hf_state_dict(sd)
    m.load_hf_state_dict({k[len("vision_model."):]: v for k, v in sd.items()})
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 224, 224))          # CPU tensors: no fallback

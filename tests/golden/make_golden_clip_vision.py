"""Golden vectors of the zero-shot front end's image encoder.

    python tests/golden/make_golden_clip_vision.py        # writes tests/golden/clip_vision_<case>.npz

Source of truth: the reference's OWN ``CLIPVisionTransformer_forward`` (adaface/subj_basis_generator.py:670-737), imported
from /root/reference and called UNBOUND on the vision transformer of this image's ``transformers`` (5.15.0) ``CLIPVisionModel``
-- the function the reference's ``CLIPVisionModelWithMask`` binds to ``self.vision_model`` (:740-744).  (The subclass itself
cannot be constructed under transformers 5 -- ``CLIPVisionModel`` no longer has the ``vision_model`` attribute it patches -- but
its forward runs as written on the module that attribute used to be.)  The mask handling, the order of the blocks, the pooled
output and the returned token mask are therefore the reference's; the per-layer hidden states are read with forward hooks on
``encoder.layers[i]`` (transformers 5's encoder no longer returns them from that call).  Weights:
``synth.synthetic_clip_vision_state_dict`` (a function of tensor name and seed: only seeds and outputs are committed).  Inputs:
seeded pixel values, a mask with hard and fractional values (the reference hands a bilinear-resized mask over,
ddpm.py:2395-2409)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from adaprompt_amd import synth          # noqa: E402

CASES = {
    "narrow_quick": dict(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, image_size=224,
                         patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5),
    "narrow_gelu": dict(hidden_size=128, intermediate_size=384, num_hidden_layers=2, num_attention_heads=2, image_size=224,
                        patch_size=14, hidden_act="gelu", layer_norm_eps=1e-5),
    "vitl_2layers": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=2, num_attention_heads=16, image_size=224,
                         patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5),
}


def case_inputs(name, B=2):
    x = synth.synthetic_input(f"clipv.{name}.pixels", (B, 3, 224, 224))
    blob = synth.synthetic_input(f"clipv.{name}.mask", (B, 1, 8, 8))
    mask = torch.sigmoid(4 * F.interpolate(blob, size=(224, 224), mode="bilinear", align_corners=False))[:, 0]
    mask = torch.where(mask > 0.7, torch.ones_like(mask), torch.where(mask < 0.3, torch.zeros_like(mask), mask))
    return x, mask


def sub(t):
    """what of a [B,257,H] tensor is committed: every 4th token and channel at full width, every 2nd token when narrow"""
    return t[:, ::4, ::4].contiguous() if t.shape[-1] >= 1024 else t[:, ::2].contiguous()


_REF = {}


def reference_forward():
    """adaface.subj_basis_generator.CLIPVisionTransformer_forward, imported from the reference tree (process 2)"""
    if "f" not in _REF:
        import types
        sys.dont_write_bytecode = True
        sys.modules.setdefault("cv2", types.ModuleType("cv2"))            # adaface/util.py:6, import time only
        sys.path.insert(0, "/root/reference")
        import adaface.subj_basis_generator as sbg
        assert sbg.__file__.startswith("/root/reference")
        _REF["f"] = sbg.CLIPVisionTransformer_forward
    return _REF["f"]


def run_hf(cfg, sd, x, mask):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    m = CLIPVisionModel(CLIPVisionConfig(**cfg)).eval()
    own = m.state_dict()
    strip = "vision_model." if not any(k.startswith("vision_model.") for k in own) else ""
    missing, unexpected = m.load_state_dict({k[len(strip):]: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    vt = m if strip else m.vision_model
    hidden = []
    hooks = [vt.pre_layrnorm.register_forward_hook(lambda mod, i, o: hidden.append(o))]
    for lyr in vt.encoder.layers:
        hooks.append(lyr.register_forward_hook(lambda mod, i, o: hidden.append(o[0] if isinstance(o, tuple) else o)))
    with torch.no_grad():
        out = reference_forward()(vt, pixel_values=x, attn_mask=mask, output_attentions=False, output_hidden_states=True,
                                  return_dict=True)
    for h in hooks:
        h.remove()
    assert len(hidden) == cfg["num_hidden_layers"] + 1
    assert torch.equal(out.last_hidden_state, hidden[-1])
    tm = None if out.attn_mask is None else out.attn_mask.permute(0, 2, 1)
    return hidden, out.pooler_output, tm


def main():
    import transformers
    for name, cfg in CASES.items():
        sd = synth.synthetic_clip_vision_state_dict(cfg)
        x, mask = case_inputs(name)
        rec = {"transformers_version": np.array(transformers.__version__), "pixels_sample": x.flatten()[::3001].numpy(),
               "mask_sample": mask.flatten()[::997].numpy()}
        for tag, m in (("masked", mask), ("invmask", 1 - mask), ("plain", None)):
            hidden, pooled, tm = run_hf(cfg, sd, x, m)
            rec[f"{tag}.penultimate"] = sub(hidden[-2]).numpy()
            rec[f"{tag}.last"] = sub(hidden[-1]).numpy()
            if m is None:
                rec[f"{tag}.embeddings"] = sub(hidden[0]).numpy()
            rec[f"{tag}.pooled"] = pooled.numpy()
            if tm is not None:
                rec[f"{tag}.token_mask"] = tm.numpy()
        out = os.path.join(HERE, f"clip_vision_{name}.npz")
        np.savez_compressed(out, **rec)
        print(name, os.path.getsize(out) // 1024, "KB", {k: v.shape for k, v in rec.items() if k.startswith("masked")})


if __name__ == "__main__":
    main()

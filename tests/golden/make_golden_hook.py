"""Golden vectors of the embedding hook (SURVEY.md 8c fixture 4): the reference's own
``adaface.subj_basis_generator.SubjBasisGenerator`` (background path -- the foreground path needs HF CLIP weights and a
``transformers`` API this image no longer has) imported unmodified from /root/reference on CPU, in a process of its own (the
module rebinds ``sys.modules['ldm']``, subj_basis_generator.py:23).

    python tests/golden/make_golden_hook.py        # writes tests/golden/hook_bg_sbg.npz

Weights: the module's own default initialisation under ``torch.manual_seed(1234)`` (no checkpoint exists offline), with the
output LayerNorms / latent queries re-drawn so nothing is at a trivial value.  Recorded: the output for a fixed
``clip_features`` [2,257,1024] -> [2,16,4,768], the gradient of <output, U> w.r.t. every parameter (norm + a strided sample)
for a fixed upstream U, and the gradient w.r.t. the input features.  tests/test_hook_boundary.py replays the same call through
``adaprompt_amd.hook_bridge`` and through ``LatentDiffusion``'s conditioning assembly."""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hook_bg_sbg.npz")


def build_reference_bg_sbg(seed=1234):
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))                      # adaface/util.py:6, import-time only
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import adaface.subj_basis_generator as sbg
    sbg.CLIPTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: object())   # offline; unused on the bg path
    torch.manual_seed(seed)
    g = sbg.SubjBasisGenerator(num_out_embs_per_layer=4, num_out_layers=16, image_embedding_dim=1024, output_dim=768,
                               placeholder_is_bg=True, prompt2token_proj_grad_scale=1,
                               bg_prompt_translator_has_to_out_proj=False, zs_extra_words_scale=0.5)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in g.named_parameters():
            if p.dim() == 1 and (n.endswith("weight") or n.endswith("bias")):
                p.add_(0.05 * torch.randn(p.shape, generator=gen))
    return g


def sample(t, n=64):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].clone()


def main():
    g = build_reference_bg_sbg()
    gen = torch.Generator().manual_seed(7)
    feats = (torch.randn(2, 257, 1024, generator=gen) * 0.5).requires_grad_(True)
    U = torch.randn(2, 16, 4, 768, generator=gen)
    g.train()
    out, prompt_embs = g(None, feats, None, 1.0, is_face=False, is_training=True, adaface_prompt_embs_inf_type="full_half_pad")
    assert prompt_embs is None and tuple(out.shape) == (2, 16, 4, 768), (prompt_embs, out.shape)
    (out * U).sum().backward()
    # inputs are re-drawn from the seed by the test (torch.Generator().manual_seed(7): feats, then U); only checks of them travel
    rec = {"input_seed": np.int64(7), "clip_features_sample": sample(feats, 256).numpy(), "upstream_sample": sample(U, 256).numpy(),
           "out": out.detach().numpy().astype(np.float32), "grad_clip_features_norm": np.float64(feats.grad.double().norm()),
           "grad_clip_features_sample": sample(feats.grad, 1024).numpy(),
           "n_params": np.int64(sum(p.numel() for p in g.parameters()))}
    names = []
    for n, p in g.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        rec["gnorm/" + n] = np.float64(p.grad.double().norm())
        rec["gsamp/" + n] = sample(p.grad).numpy()
    rec["param_names"] = np.array(names)
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, "params", int(rec["n_params"]), "with grad", len(names), "out norm", float(out.norm()))


if __name__ == "__main__":
    main()

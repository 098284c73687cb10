"""Synthetic stand-in for the embedding hook, used ONLY by bench.py / smoke() / tests when the real one
cannot run (no HF CLIP weights offline).

In the reference the layerwise text context [16*B, 77, 768] is produced by
``get_learned_conditioning`` -> ``EmbeddingManager`` -> ``adaface.subj_basis_generator
.SubjBasisGenerator.forward`` (ddpm.py:970-1085, embedding_manager.py:1292-1588,
subj_basis_generator.py:470-567); those stay the reference's own classes behind ``cond_fn`` and are
the only trainable part (~149 M fp32 parameters in the shipped config).  This module
has the same interface towards the hot path -- id embedding in, context out, gradients into
~149 M parameters that the optimizer owns and the data-parallel all-reduce exchanges -- with
negligible FLOPs of its own: the context is an id-dependent mixture of learnable bases."""
import torch
import torch.nn as nn


class SyntheticSubjBasisGenerator(nn.Module):
    def __init__(self, n_params=149_000_000, num_layers=16, tokens=77, dim=768, id_dim=512):
        super().__init__()
        per = num_layers * tokens * dim
        self.G = max(1, n_params // per)
        self.shape = (num_layers, tokens, dim)
        self.bases = nn.Parameter(torch.randn(self.G, per) * 0.05)
        self.gate = nn.Linear(id_dim, self.G)

    def forward(self, id_embs):
        """id_embs [B, id_dim] -> context [16*B, 77, 768] with the 16 layers of an instance contiguous
        (embedding_manager.py:1345-1349)."""
        a = torch.softmax(self.gate(id_embs), dim=-1)               # [B, G]
        ctx = a @ self.bases                                         # [B, 16*77*768]
        B = id_embs.shape[0]
        L, T, D = self.shape
        return ctx.view(B * L, T, D)


def make_cond_fn(hook, capture=True):
    """cond_fn(batch) -> (c_static_emb, prompts, extra_info), the triple ``DiffusionWrapper`` unpacks
    (ddpm.py:5523-5533) with the extra_info keys UNetModel.forward reads (openaimodel.py:849-859)."""
    def cond_fn(batch):
        ctx = hook(batch["zs_id_embs"])
        extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
                 "is_training": True, "capture_distill_attn": capture, "placeholder2indices": None}
        return ctx, None, extra
    return cond_fn

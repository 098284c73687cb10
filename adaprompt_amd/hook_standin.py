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


SUBJ_TOKENS = (4, 20)       # prompt positions of the 16 subject embeddings (num_vectors_per_subj_token = 16)
BG_TOKENS = (24, 28)        # ... and of the 4 background embeddings (num_vectors_per_bg_token = 4)


def make_cond_fn(hook, capture=True, regs=False):
    """cond_fn(batch) -> (c_static_emb, prompts, extra_info), the triple ``DiffusionWrapper`` unpacks
    (ddpm.py:5523-5533) with the extra_info keys UNetModel.forward reads (openaimodel.py:849-859).

    ``regs``: also supply what the recon iteration's regularisers read (ddpm.py:3207-3270), shaped as the reference's
    conditioning side produces it: the (instance, token) indices of the subject / background embeddings
    (``placeholder2indices`` joined per ddpm.py:2566-2569), the four-way static embeddings ``c_static_emb_4b``
    [4B, 16, 77, 768] = (subject-single, subject-comp, class-single, class-comp) of which the first block is the
    context the UNet sees (ddpm.py:2040-2179), and ``prompt_emb_mask`` [4B, 77, 1] (1 = token, 0.5 = padding).  The
    subject blocks depend on the hook's parameters, the class blocks are constants -- as in the reference."""
    consts = {}

    def cond_fn(batch):
        ids = batch["zs_id_embs"]
        ctx = hook(ids)
        extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon",
                 "is_training": True, "capture_distill_attn": capture, "placeholder2indices": None}
        if regs:
            B, dev = ids.shape[0], ids.device
            L = ctx.shape[0] // B
            key = (B, str(dev), tuple(ctx.shape[1:]))
            if key not in consts:
                g = torch.Generator(device="cpu").manual_seed(4242)
                cls = (torch.randn(2, B, L, *ctx.shape[1:], generator=g) * 0.05).to(dev)
                mask = torch.full((4 * B, ctx.shape[1], 1), 0.5, device=dev)
                for blk, n_tok in enumerate((SUBJ_TOKENS[1] + 1, 31, SUBJ_TOKENS[1] + 1, 31)):   # single / comp prompt lengths
                    mask[blk * B:(blk + 1) * B, :n_tok] = 1.0
                inst = torch.arange(B, device=dev)
                consts[key] = (cls, mask,
                               (inst.repeat_interleave(SUBJ_TOKENS[1] - SUBJ_TOKENS[0]),
                                torch.arange(*SUBJ_TOKENS, device=dev).repeat(B)),
                               (inst.repeat_interleave(BG_TOKENS[1] - BG_TOKENS[0]),
                                torch.arange(*BG_TOKENS, device=dev).repeat(B)))
            cls, mask, subj_idx, bg_idx = consts[key]
            single = ctx.view(B, L, *ctx.shape[1:])
            comp = single.roll(1, dims=0) * 0.25 + single * 0.75 + cls[1] * 0.5      # a "compositional" variant
            extra.update(subj_indices=subj_idx, bg_indices=bg_idx, prompt_emb_mask=mask.clone(),
                         c_static_emb_4b=torch.cat([single, comp, cls[0], cls[0] + cls[1] * 0.5], dim=0))
        return ctx, None, extra
    return cond_fn

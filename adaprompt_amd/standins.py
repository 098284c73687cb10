"""Stand-ins for the two boundary callees the yaml names (``cond_stage_config`` -> FrozenCLIPEmbedder,
``personalization_config`` -> EmbeddingManager) with the call contract ``LatentDiffusion`` relies on
(reference ldm/modules/encoders/modules.py:449-463, ldm/modules/embedding_manager.py:1292-1588, 1668, 1824, 2078-2095):
tests of the constructor surface and of the conditioning assembly, and the bench legs that need a conditioning side, run
without HF CLIP weights.  Like ``hook_standin.SyntheticSubjBasisGenerator`` these are payload stand-ins for third-party
models behind the boundary, never part of a parity claim (``tests/stubs.py`` re-exports them under their test names)."""
import torch
import torch.nn as nn


class StubTextEncoder(nn.Module):
    """``encode(prompts, embedding_manager=) -> [16 * len(prompts), 77, dim]``: a deterministic per-token embedding of each
    prompt's words, the same at every layer, then the embedding manager's subject rows written in."""

    def __init__(self, dim=32, last_layers_skip_weights=(0.5, 0.5)):
        super().__init__()
        self.dim = dim
        self.table = nn.Parameter(torch.randn(997, dim, generator=torch.Generator(device="cpu").manual_seed(5), device="cpu"), requires_grad=False)
        self.sampled = 0
        self.device = torch.device("cpu")

    def sample_last_layers_skip_weights(self):
        self.sampled += 1

    def tokenize(self, prompt):
        ids = [1] + [2 + (sum(map(ord, w)) % 900) for w in prompt.split()][:75]
        return ids + [0] * (77 - len(ids))

    def encode(self, prompts, embedding_manager=None):
        ids = torch.tensor([self.tokenize(p) for p in prompts], device=self.table.device)
        emb = self.table[ids]                                                        # [B, 77, dim]
        emb = emb[:, None].expand(-1, 16, -1, -1).reshape(len(prompts) * 16, 77, self.dim).clone()
        if embedding_manager is not None:
            emb = embedding_manager(ids, emb, prompts)
        return emb


class StubEmbeddingManager(nn.Module):
    """Places K learnable subject vectors (per layer) at the positions of the word ``z`` and Kb background vectors at ``y``,
    records ``placeholder2indices`` / ``prompt_emb_mask`` the way the reference's forward does."""

    def __init__(self, text_embedder=None, subject_strings=("z",), background_strings=("y",), num_vectors_per_subj_token=9,
                 num_vectors_per_bg_token=4, dim=None, **unused):
        super().__init__()
        self.text_embedder = text_embedder
        dim = dim or getattr(text_embedder, "dim", 32)
        self.subject_string_dict = {s: True for s in subject_strings}
        self.background_string_dict = {s: True for s in background_strings}
        self.K = {**{s: num_vectors_per_subj_token for s in subject_strings},
                  **{s: num_vectors_per_bg_token for s in background_strings}}
        self.vectors = nn.ParameterDict({s: nn.Parameter(torch.randn(16, k, dim, generator=torch.Generator(device="cpu").manual_seed(11 + i), device="cpu"))
                                         for i, (s, k) in enumerate(self.K.items())})
        self.use_conv_attn_kernel_size = -1
        self.placeholder2indices, self.prompt_emb_mask = {}, None
        self.cls_delta_string_indices, self.subj_name_to_cls_delta_token_weights = [], {}
        self.subj_name_to_being_faces = {"arc2face": True, "zs_default": True, "alice": True, "bob": True}
        self.training_percent = 0.0
        self.calls = []

    def set_zs_image_features(self, feats, ids, zs_out_id_embs_scale_range=(1.0, 1.0), add_noise_to_zs_id_embs=True):
        self.calls.append(("set_zs_image_features", None if feats is None else tuple(feats.shape),
                           None if ids is None else tuple(ids.shape), add_noise_to_zs_id_embs))
        self.zs_id_embs = ids

    def set_curr_iter_type(self, t):
        self.calls.append(("set_curr_iter_type", t))

    def set_curr_batch_subject_names(self, names, iter_type):
        self.calls.append(("set_curr_batch_subject_names", tuple(names), iter_type))

    def make_frozen_copy_of_subj_basis_generators(self):
        self.calls.append(("make_frozen_copy",))

    def forward(self, ids, emb, prompts):
        B = len(prompts)
        emb = emb.view(B, 16, 77, -1).clone()
        self.placeholder2indices = {}
        mask = torch.full((B, 77, 1), 0.5, device=emb.device)
        dev = emb.device
        for b, p in enumerate(prompts):
            words = p.split()
            mask[b, :len(words) + 2] = 1.0
            for s, k in self.K.items():
                if s in words:
                    pos = 1 + words.index(s)
                    vec = self.vectors[s]
                    if getattr(self, "zs_id_embs", None) is not None and self.zs_id_embs.shape[0] > b % self.zs_id_embs.shape[0]:
                        vec = vec * (1 + self.zs_id_embs[b % self.zs_id_embs.shape[0]].mean())
                    emb[b, :, pos:pos + k] = vec
                    ib, it = self.placeholder2indices.get(s, (torch.zeros(0, dtype=torch.long, device=dev), torch.zeros(0, dtype=torch.long, device=dev)))
                    self.placeholder2indices[s] = (torch.cat([ib, torch.full((k,), b, device=dev)]), torch.cat([it, torch.arange(pos, pos + k, device=dev)]))
        self.prompt_emb_mask = mask
        return emb.view(B * 16, 77, -1)

    def optimized_parameters(self):
        return [{"params": list(self.vectors.parameters()), "lr_ratio": 1.0, "excluded_from_prodigy": False}]

    def save(self, path):
        torch.save({"vectors": {k: v.detach() for k, v in self.vectors.items()}}, path)

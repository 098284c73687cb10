"""The embedding hook through the boundary: ``adaface.subj_basis_generator.SubjBasisGenerator`` stays the
reference's own class and is CALLED with the reference's argument list; this module owns only what is around the call.

In the reference the call sits inside ``EmbeddingManager.get_static_embedding`` (embedding_manager.py:1434-1442):

    adaface_subj_embs, adaface_prompt_embs = subj_basis_generator(
        arc2face_id_embs, zs_clip_features, zs_id_embs, zs_out_id_embs_scale_range[0],
        is_face=..., is_training=..., adaface_prompt_embs_inf_type=...)          # -> [BS, 16, K, 768], [BS, 77, 768] | None

and the K vectors of every layer are written over the placeholder token's K positions of the CLIP prompt embedding
(:1516-1562), the 16 layers of an instance contiguous (:1345-1349).  With the full reference stack on the path
(yaml-instantiated ``EmbeddingManager`` + ``FrozenCLIPEmbedder``) ``LatentDiffusion`` runs exactly that
(``conditioning.ConditioningMixin``).  ``make_cond_fn_from_reference_hook`` is the same call and the same placement for a
hook on its own -- a frozen prompt embedding as the background of the context -- so the real hook can be trained through the
MI355X UNet without the CLIP text tower (bench / tests on a box without HF weights), and ``hook_optimized_parameters`` hands
its parameters to ``configure_optimizers`` in the shape ``EmbeddingManager.optimized_parameters()`` has (:2078-2095)."""
import torch


def call_hook(hook, clip_features, arc2face_id_embs=None, raw_id_embs=None, out_id_embs_scale=1.0, is_face=False,
              is_training=True, adaface_prompt_embs_inf_type="full_half_pad"):
    """the reference's call (embedding_manager.py:1434-1442), positional order included."""
    return hook(arc2face_id_embs, clip_features, raw_id_embs, out_id_embs_scale, is_face=is_face, is_training=is_training,
                adaface_prompt_embs_inf_type=adaface_prompt_embs_inf_type)


def place_subject_embeddings(base_context, subj_embs, token_start):
    """base_context [16, 77, D] (one prompt, layerwise) or [BS, 16, 77, D]; subj_embs [BS, 16, K, D] -> context
    [16 * BS, 77, D] with rows token_start .. token_start + K of every layer replaced, differentiable w.r.t. subj_embs."""
    BS, L, K, D = subj_embs.shape
    if base_context.dim() == 3:
        base_context = base_context.unsqueeze(0).expand(BS, -1, -1, -1)
    assert base_context.shape[:2] == (BS, L) and base_context.shape[3] == D
    ctx = torch.cat([base_context[:, :, :token_start], subj_embs.to(base_context.dtype),
                     base_context[:, :, token_start + K:]], dim=2)
    return ctx.reshape(BS * L, ctx.shape[2], D)


def make_cond_fn_from_reference_hook(hook, base_context, token_start, clip_feature_key="zs_clip_features",
                                     id_key="zs_id_embs", is_face=False, extra_info=None):
    """-> cond_fn(batch) for ``LatentDiffusion(cond_fn=...)``: ``batch[clip_feature_key]`` [BS, 257, D_clip] (and, for a
    face hook, ``batch[id_key]``) -> ``hook`` -> context.  ``extra_info`` defaults to the recon iteration's keys."""
    info = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "normal_recon", "is_training": True,
            "capture_distill_attn": True, "placeholder2indices": None}
    info.update(extra_info or {})

    def cond_fn(batch):
        feats = batch[clip_feature_key]
        subj_embs, _prompt_embs = call_hook(hook, feats, raw_id_embs=batch.get(id_key), is_face=is_face,
                                            is_training=hook.training)
        K = subj_embs.shape[2]
        ctx = place_subject_embeddings(base_context.to(subj_embs.device), subj_embs, token_start)
        ex = dict(info)
        BS = subj_embs.shape[0]
        inst = torch.arange(BS, device=ctx.device)
        ex["subj_indices"] = (inst.repeat_interleave(K), torch.arange(token_start, token_start + K, device=ctx.device).repeat(BS))
        return ctx, None, ex
    return cond_fn


def hook_optimized_parameters(hooks, extra_slow_params=()):
    """``EmbeddingManager.optimized_parameters()`` (embedding_manager.py:2078-2095) for bare hooks: one group with every
    requires-grad parameter of the subject-basis generators (lr_ratio 1), one with the slow parameters (lr_ratio 0.1:
    ``emb_global_scale_scores`` there), none excluded from Prodigy."""
    hooks = hooks if isinstance(hooks, (list, tuple)) else [hooks]
    sbg = [p for h in hooks for p in h.parameters() if p.requires_grad]
    groups = [{"params": sbg, "lr_ratio": 1, "excluded_from_prodigy": False}]
    if extra_slow_params:
        groups.append({"params": list(extra_slow_params), "lr_ratio": 0.1, "excluded_from_prodigy": False})
    return groups

"""Stage-2 compositional distillation (BASELINE config 4): the host-side arithmetic around the UNet passes of a
compositional iteration -- the K/V-split prompt mixing, the teacher selection by CLIP score, the elastic-matching / delta-
alignment losses on the captured ``attnscore`` / ``q`` / ``outfeat`` -- with the reference's names, arguments and results
(ldm/util.py:386-408, 543-594, 648-684, 1384-1408, 1600-1855, 2093-2101, 2163-2224, 2241-2368; pinned by golden vectors the
reference's own functions produced, tests/golden/make_golden_ddpm.py).  Everything here is device-agnostic torch on tensors
that are small next to the UNet's (token maps, pooled feature maps); the UNet passes themselves -- 154-token split K/V
context, guidance pass, differentiable side outputs -- run on the HIP kernels (``LatentDiffusion.compos_distill_step``)."""
import math
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F

from .util import (calc_ref_cosine_loss, gen_gradient_scaler, masked_mean, ortho_subtract)

SYNC_LAYER_INDICES = [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]      # CA layers 7, 8, 12, 16 ... 24 (ldm/util.py:1726)


# ---- index helpers (ldm/util.py:1384-1408) -------------------------------------------------------------------------------
def extend_indices_B_by_n_times(indices, n, block_offset):
    """(idx_B, idx_N) of block 0 -> the same token positions in n consecutive blocks of ``block_offset`` instances."""
    if indices is None:
        return None, None
    ib, it = indices
    blocks = [(ib + block_offset * i, it) for i in range(n)]
    return (torch.cat([b for b, _ in blocks]), torch.cat([t for _, t in blocks])), blocks


def double_token_indices(token_indices, bs_offset):
    return None if token_indices is None else extend_indices_B_by_n_times(token_indices, 2, bs_offset)[0]


def chunk_list(lst, num_chunks):
    size = len(lst) // num_chunks
    return [lst[i * size:(i + 1) * size] for i in range(num_chunks)]


# ---- prompt mixing (ldm/util.py:1600-1855) --------------------------------------------------------------------------------
def mix_embeddings(mix_scheme, c1, c2, mix_indices=None, c1_mix_scale=1., c2_mix_weight=None, use_ortho_subtract=True):
    """'add': c1 everywhere except at the token positions ``mix_indices``, where it is c1 * scale + c2 * (1 - scale);
    ``c1_mix_scale`` may be a per-row tensor (one scale per layer-instance row of c1), repeated to cover the batch."""
    if c2 is None:
        return c1
    assert c1.shape == c2.shape
    if mix_scheme != "add":
        raise NotImplementedError(f"mix scheme {mix_scheme!r}: only 'add' is on the mix_hijk path")
    if torch.is_tensor(c1_mix_scale):
        if bool((c1_mix_scale == 1).all()):
            return c1
    elif c1_mix_scale == 1:
        return c1
    if mix_indices is None:
        return c1 * c1_mix_scale + c2 * (1 - c1_mix_scale)
    scale_mask = torch.ones_like(c1)
    if torch.is_tensor(c1_mix_scale):
        if len(c1_mix_scale) < len(scale_mask):
            assert len(scale_mask) % len(c1_mix_scale) == 0
            c1_mix_scale = c1_mix_scale.repeat(len(scale_mask) // len(c1_mix_scale))
        while c1_mix_scale.ndim < 3:
            c1_mix_scale = c1_mix_scale.unsqueeze(-1)
    scale_mask[:, mix_indices] = c1_mix_scale
    return c1 * scale_mask + c2 * (1 - scale_mask)


def gen_emb_mixer(BS, subj_indices_1b_N, CLS_SCALE_LAYERWISE_RANGE, device, use_layerwise_embedding=True, N_CA_LAYERS=16,
                  sync_layer_indices=SYNC_LAYER_INDICES):
    """-> (mixer(cls_emb, subj_emb, c1_mix_scale=...), per-layer class-embedding scales [BS, 16]): 1 on the first four
    layers, then linear from the first to the last value of the range over the 12 synchronised layers."""
    first, last = CLS_SCALE_LAYERWISE_RANGE
    if use_layerwise_embedding:
        step = (last - first) / (len(sync_layer_indices) - 1)
        scales = torch.ones(BS, N_CA_LAYERS, device=device)
        scales[:, sync_layer_indices] = first + torch.arange(0, len(sync_layer_indices), device=device).repeat(BS, 1) * step
    else:
        scales = (first + last) / 2 * torch.ones(N_CA_LAYERS, device=device).repeat(BS, 1)
    return partial(mix_embeddings, "add", mix_indices=subj_indices_1b_N), scales


def mix_static_vk_embeddings(c_static_emb, subj_indices_1b_N, training_percent, t_frac=1.0, use_layerwise_embedding=True,
                             N_CA_LAYERS=16, K_CLS_SCALE_LAYERWISE_RANGE=(1.0, 1.0), V_CLS_SCALE_LAYERWISE_RANGE=(1.0, 0.7),
                             sync_layer_indices=SYNC_LAYER_INDICES):
    """c_static_emb = (subject half | class half) [2 * BS * 16, 77, D] -> [2 * BS * 16, 154, D]: per instance the V context
    (first 77 tokens) and the K context (last 77).  The subject half is the subject embedding twice; the class ("mix") half
    has part of the subject embedding blended into the class prompt at the subject's token positions (per-layer scales,
    separately for V and K; gradient x 0.05), and on the synchronised layers it is itself blended with the subject
    context by 1 - t_frac * (1 - 0.3 * training_percent): the noisier the sample, the more of the class prompt."""
    subj_emb, cls_emb = c_static_emb.chunk(2)
    BS = subj_emb.shape[0] // N_CA_LAYERS
    if not torch.is_tensor(t_frac):
        t_frac = torch.tensor(t_frac, dtype=c_static_emb.dtype, device=c_static_emb.device).reshape(-1)
    if len(t_frac) == 1:
        t_frac = t_frac.repeat(BS)
    assert len(t_frac) == BS and -1e-6 <= training_percent <= 1 + 1e-6
    t_frac = t_frac.unsqueeze(1)
    emb_v_mixer, v_scales = gen_emb_mixer(BS, subj_indices_1b_N, V_CLS_SCALE_LAYERWISE_RANGE, c_static_emb.device,
                                          use_layerwise_embedding, N_CA_LAYERS, sync_layer_indices)
    mix_emb_v = emb_v_mixer(cls_emb, subj_emb, c1_mix_scale=v_scales.view(-1))
    emb_k_mixer, k_scales = gen_emb_mixer(BS, subj_indices_1b_N, K_CLS_SCALE_LAYERWISE_RANGE, c_static_emb.device,
                                          use_layerwise_embedding, N_CA_LAYERS, sync_layer_indices)
    mix_emb_k = emb_k_mixer(cls_emb, subj_emb, c1_mix_scale=k_scales.view(-1))
    mix_all = gen_gradient_scaler(0.05)(torch.cat([mix_emb_v, mix_emb_k], dim=1))
    subj_emb2 = subj_emb.repeat(1, 2, 1)
    if use_layerwise_embedding:
        layer_mask = torch.zeros_like(mix_all).reshape(-1, N_CA_LAYERS, *mix_all.shape[1:])
        layer_mask[:, sync_layer_indices] = 1 - t_frac.view(-1, 1, 1, 1) * (1 - training_percent * 0.3)
        layer_mask = layer_mask.reshape(-1, *mix_all.shape[1:])
        mix_emb = subj_emb2 * layer_mask + mix_all * (1 - layer_mask)
    else:
        mix_emb = mix_all
    return torch.cat([subj_emb2, mix_emb], dim=0), emb_v_mixer, v_scales, emb_k_mixer, k_scales


def gen_cfg_scales_for_stu_tea(tea_scale, stu_scale, num_teachers, device):
    return torch.cat([torch.ones(num_teachers) * stu_scale, torch.ones(num_teachers) * tea_scale]).to(device)


def calc_dyn_loss_scale(loss, loss_base, loss_scale_base, min_scale_base_ratio=1, max_scale_base_ratio=2):
    if loss_base == 0:
        return 0
    scale = float(loss.detach()) * loss_scale_base / loss_base
    return max(min(loss_scale_base * max_scale_base_ratio, scale), loss_scale_base * min_scale_base_ratio)


def rand_annealed(training_percent, final_percent, mean_range, fluct_range=(0.8, 1.2), legal_range=(0, 1), np_random=np.random):
    """ldm/util.py:1487-1493: one uniform draw within +-20 % of an annealed mean, clipped to the legal range."""
    from .util import anneal_value
    mean = anneal_value(training_percent, final_percent, mean_range)
    return np_random.uniform(max(mean * fluct_range[0], legal_range[0]), min(mean * fluct_range[1], legal_range[1]))


def init_x_with_fg_from_training_image(x_start, fg_mask, filtered_fg_mask, training_percent, base_scale_range=(0.7, 1.0),
                                       fg_noise_anneal_mean_range=(0.1, 0.5), np_random=np.random):
    """the compositional iteration's initial latent: the training image's foreground, shrunk by a random factor (more when
    it fills more than a tenth of the image), centred, on fresh noise, then partly re-noised (ldm/util.py:2163-2217).
    Consumes ``np.random.uniform`` once for the scale, ``randn`` three times, ``np.random.uniform`` once for the amount."""
    x_orig = torch.where(filtered_fg_mask.bool(), x_start, torch.randn_like(x_start))
    pct = filtered_fg_mask.float().sum() / filtered_fg_mask.numel()
    lb, ub = base_scale_range
    if pct > 0.1:
        extra = math.pow(0.1 / pct.item(), 0.35)
        scale = np_random.uniform(lb * extra, max(0.5, ub * extra))
    else:
        scale = np_random.uniform(lb, ub)
    x_mask = F.interpolate(torch.cat([x_orig, fg_mask, filtered_fg_mask], dim=1), scale_factor=scale, mode="bilinear",
                           align_corners=False)
    pw1 = int((x_start.shape[3] - x_mask.shape[3]) / 2)
    pw2 = x_start.shape[3] - x_mask.shape[3] - pw1
    ph1 = int((x_start.shape[2] - x_mask.shape[2]) / 2)
    ph2 = x_start.shape[2] - x_mask.shape[2] - ph1
    x_mask = F.pad(x_mask, (pw1, pw2, ph1, ph2), mode="constant", value=0)
    x_scaled, fg_mask, filtered_fg_mask = x_mask[:, :4], x_mask[:, [4]], x_mask[:, [5]]
    x_start = torch.where(filtered_fg_mask.bool(), x_scaled, torch.randn_like(x_start))
    amount = rand_annealed(training_percent, final_percent=1, mean_range=fg_noise_anneal_mean_range, np_random=np_random)
    return torch.randn_like(x_start) * amount + x_start * (1 - amount), fg_mask, filtered_fg_mask


# ---- losses (ldm/util.py:386-395, 543-594, 648-684, 2241-2368) -------------------------------------------------------------
def ortho_l2loss(a, b, mean=True, do_sqrt=False):
    r = ortho_subtract(a, b)
    loss = r * r
    if mean:
        loss = loss.mean()
    return loss.sqrt() if do_sqrt else loss


def calc_delta_alignment_loss(feat_base, feat_ex, ref_feat_base, ref_feat_ex, ref_grad_scale=0.1, feat_base_grad_scale=0.05,
                              use_cosine_loss=True, cosine_exponent=2, delta_types=("feat_to_ref", "ex_to_base")):
    """the change (orthogonal component) from base to extended features should point the way the reference's does."""
    if not use_cosine_loss:
        raise NotImplementedError("calc_delta_alignment_loss: only the cosine form is used (ddpm.py:3822-3827)")
    ref_gs = gen_gradient_scaler(ref_grad_scale)
    ref_base, ref_ex = ref_gs(ref_feat_base), ref_gs(ref_feat_ex)
    if feat_base_grad_scale == -1:
        feat_base_grad_scale = min(ref_grad_scale / 2, 1)
    base = gen_gradient_scaler(feat_base_grad_scale)(feat_base)
    out = {}
    for choice in delta_types:
        if choice == "feat_to_ref":
            src, tgt = ortho_subtract(base, ref_base), ortho_subtract(feat_ex, ref_ex)
        elif choice == "ex_to_base":
            src, tgt = ortho_subtract(ref_ex, ref_base), ortho_subtract(feat_ex, base)
        else:
            raise ValueError(choice)
        out[choice] = calc_ref_cosine_loss(tgt, src, exponent=cosine_exponent, do_demean_first=False,
                                           first_n_dims_to_flatten=feat_base.ndim - 1, ref_grad_scale=1, aim_to_align=True)
    return out


def convert_attn_to_spatial_weight(flat_attn, BS, out_spatial_shape, reversed=True):
    """subject attention [BS * n, heads, N] -> per-pixel weight [BS, 1, H, W] with mean 1 that is SMALL where the subject
    attends (``reversed``): exp(-(a - mean) / max(std + 0.001, mean / 2)), capped at 1.  Detached."""
    flat_attn = flat_attn.detach().reshape(BS, -1, *flat_attn.shape[1:])
    out_numel = int(out_spatial_shape[0]) * int(out_spatial_shape[1])
    scale = np.sqrt(flat_attn.shape[-1] / out_numel)
    shape2 = (int(out_spatial_shape[0] * scale), int(out_spatial_shape[1] * scale))
    attn = flat_attn.mean(dim=2).sum(dim=1).reshape(BS, 1, *shape2)
    attn = F.interpolate(attn, size=tuple(int(s) for s in out_spatial_shape), mode="bilinear", align_corners=False)
    mean, std = attn.mean(dim=(2, 3), keepdim=True), attn.std(dim=(2, 3), keepdim=True)
    denom = torch.clamp(std + 0.001, min=mean / 2)
    w = torch.exp((-1 if reversed else 1) * (attn - mean) / denom).clamp(max=1)
    return w / w.mean(dim=(2, 3), keepdim=True), attn


ELASTIC_FUSED = True      # CUDA tensors take the one-call HIP form (functional.ElasticMatchFn); False: the torch expressions
                          # below (what CPU tensors always take) -- the tests compare the two


def calc_elastic_matching_loss(ca_q, ca_outfeat, fg_mask, fg_bg_cutoff_prob=0.25, single_q_grad_scale=0.1,
                               single_feat_grad_scale=0.01, mix_feat_grad_scale=0.05, fg_any=None):
    """ca_q, ca_outfeat [4 blocks (subject single, subject comp, mix single, mix comp), C, N]; fg_mask [1, 1, N] of the
    single instances.  Soft correspondence comp -> single by the queries' dot products (softmax over the comp tokens);
    -> (|P_subj - P_mix| on foreground pairs, cosine loss of the comp features carried onto the single foreground vs the
    single foreground features, cosine loss between subject-comp and mix-comp features on the tokens that map to no
    foreground, and the two soft background weights [1, 1, N])."""
    fg = fg_mask.bool().squeeze(1)
    if not (fg.any().item() if fg_any is None else fg_any):      # (``fg_any``: the caller already knows -- no host sync here)
        return 0, 0, 0, None, None
    if ca_q.is_cuda and ELASTIC_FUSED:
        from .. import functional as HF
        assert ca_q.shape[0] == 4 and ca_outfeat.shape[0] == 4, "the four blocks of ONE instance (ddpm.py:3041: BLOCK_SIZE 1)"
        return HF.ElasticMatchFn.apply(ca_q, ca_outfeat, fg, fg_bg_cutoff_prob, single_q_grad_scale, single_feat_grad_scale,
                                       mix_feat_grad_scale)
    q_gs, feat_gs = gen_gradient_scaler(single_q_grad_scale), gen_gradient_scaler(single_feat_grad_scale)
    ss_q, sc_q, ms_q, mc_q = ca_q.chunk(4)
    # softmax over the comp tokens (dim 1 of [1, N_comp, N_single]), taken on the transposed product so that it runs over the
    # LAST dim: torch's kernel for an inner dim is ~20x slower at these sizes (961 x 961: 250 us forward, 175 us backward)
    sc_map_ss_prob = F.softmax(torch.matmul(q_gs(ss_q).transpose(1, 2), sc_q), dim=-1).transpose(1, 2)
    mc_map_ms_prob = F.softmax(torch.matmul(q_gs(ms_q).transpose(1, 2), mc_q), dim=-1).transpose(1, 2)
    ss_feat, sc_feat, ms_feat, mc_feat = ca_outfeat.chunk(4)
    _, fg_N = fg.nonzero(as_tuple=True)
    sc_recon_ss_fg = torch.matmul(sc_feat, sc_map_ss_prob[:, :, fg_N]).permute(0, 2, 1)
    ss_fg_feat = feat_gs(ss_feat.permute(0, 2, 1)[:, fg_N])
    fg_hw = fg.unsqueeze(1) * fg.unsqueeze(2)
    loss_map_align = masked_mean((sc_map_ss_prob - mc_map_ms_prob).abs(), fg_hw)
    loss_sc_ss_fg = calc_ref_cosine_loss(sc_recon_ss_fg, ss_fg_feat, exponent=2, do_demean_first=False,
                                         first_n_dims_to_flatten=2, ref_grad_scale=1)
    fgf = fg.to(sc_map_ss_prob.dtype).unsqueeze(2)
    sc_fg_prob = torch.matmul(sc_map_ss_prob, fgf).permute(0, 2, 1)
    mc_fg_prob = torch.matmul(mc_map_ms_prob, fgf).permute(0, 2, 1)
    sc_below = torch.clamp(fg_bg_cutoff_prob - sc_fg_prob, min=0)
    mc_below = torch.clamp(fg_bg_cutoff_prob - mc_fg_prob, min=0)
    loss_sc_mc_bg = calc_ref_cosine_loss(sc_feat.permute(0, 2, 1), mc_feat.permute(0, 2, 1), emb_mask=mc_below.permute(0, 2, 1),
                                         exponent=2, do_demean_first=False, first_n_dims_to_flatten=2,
                                         ref_grad_scale=mix_feat_grad_scale)
    return loss_map_align, loss_sc_ss_fg, loss_sc_mc_bg, sc_below, mc_below


def select_teacher(losses_clip_comp, clip_loss_thres=0.28, cls_subj_clip_margin=0.002):
    """ddpm.py:3636-3679: (student | teacher) CLIP losses -> (teachable mask, index of the candidate whose teacher beats
    its student by the largest margin among the teachable ones)."""
    subj, mix = losses_clip_comp.chunk(2)
    diffs = subj - mix
    teachable = (mix <= clip_loss_thres) & (diffs > cls_subj_clip_margin)
    diffs = diffs.clone()
    diffs[~teachable] = -1e4
    return teachable, int(torch.argmax(diffs).item())

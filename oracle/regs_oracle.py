"""CPU restatement of the two regularisers the recon iteration adds to the masked MSE (ddpm.py:3207-3270).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline); the product path never imports it.

  * helpers of ldm/util.py -- IMPORTABLE from /root/reference, pinned by tests/golden/regs_*.npz:
      ``ortho_subtract`` util.py:280-316, ``demean`` :425-431, ``calc_ref_cosine_loss`` :437-535,
      ``gen_gradient_scaler`` / ``ScaleGrad`` :1084-1131, ``normalize_dict_values`` :1423-1430,
      ``normalized_sum`` :2110-2121, ``calc_prompt_emb_delta_loss`` :2037-2092
      ``masked_mean`` :1450-1466, ``resize_mask_for_feat_or_attn`` :1570-1594, ``sel_emb_attns_by_indices`` :1945-1977
  * ``calc_fg_bg_xlayer_consist_loss`` ddpm.py:4259-4387 and ``calc_fg_bg_complementary_loss`` /
    ``calc_fg_mb_suppress_loss`` ddpm.py:3932-4258 -- ddpm.py cannot be imported here (pytorch_lightning etc.,
    SURVEY 8c), restated from the source text on top of the pinned helpers; PARITY UNPINNED against reference outputs
    for the methods themselves, known-answer tests in tests/test_regs_oracle.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------- util.py:280-316
def ortho_subtract(a, b):
    """a minus its projection on b along the last dim (denominator + 1e-6)."""
    assert a.ndim == b.ndim
    w = (a * b).sum(-1) / ((b * b).sum(-1) + 1e-6)
    return a - b * w.unsqueeze(-1)


# ---------------------------------------------------------------------------- util.py:425-431
def demean(x):
    return x - x.mean(dim=-1, keepdim=True)


# ---------------------------------------------------------------------------- util.py:1084-1131
class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.alpha, None


def scale_gradient(x, alpha):
    """identity in forward; the gradient is multiplied by alpha (alpha == 0: detached, alpha == 1: untouched)."""
    if alpha == 1:
        return x
    if alpha == 0:
        return x.detach()
    assert alpha > 0
    return _ScaleGrad.apply(x, float(alpha))


# ---------------------------------------------------------------------------- util.py:437-535
def calc_ref_cosine_loss(delta, ref_delta, batch_mask=None, emb_mask=None, exponent=2, do_demean_first=False,
                         first_n_dims_to_flatten=3, ref_grad_scale=0, aim_to_align=True, margin=0):
    """per sample: cosine-embedding loss between the rows of ``delta`` and of sign-preserving ``ref^exponent``
    (ref's gradient scaled by ref_grad_scale), rows weighted by ``emb_mask`` (rows with mask <= 0 dropped); samples
    weighted by ``batch_mask``; the sum is divided by batch_mask.sum()."""
    B = delta.shape[0]
    if batch_mask is not None:
        assert batch_mask.shape == (B,)
        if batch_mask.sum() == 0:
            return 0
    else:
        batch_mask = torch.ones(B, device=delta.device)
    loss = 0
    for i in range(B):
        d_i, r_i = delta[i:i + 1], ref_delta[i:i + 1]
        lead = d_i.shape[:first_n_dims_to_flatten]
        if emb_mask is not None:
            m_i = emb_mask[i:i + 1].squeeze(-1).expand(lead)
            keep = m_i > 0
            d_i, r_i, w_i = d_i[keep], r_i[keep], m_i[keep]
        else:
            d_i = d_i.reshape(lead.numel(), -1)
            r_i = r_i.reshape(d_i.shape)
            w_i = None
        if do_demean_first:
            d_i, r_i = demean(d_i), demean(r_i)
        r_i = scale_gradient(r_i, ref_grad_scale)
        r_pow = r_i * r_i.abs().pow(exponent - 1)
        label = torch.full_like(d_i[:, 0], 1.0 if aim_to_align else -1.0)
        per_row = F.cosine_embedding_loss(d_i, r_pow, label, reduction="none")
        l_i = per_row.mean() if w_i is None else (per_row * w_i).sum() / (w_i.sum() + 1e-8)
        l_i = l_i * batch_mask[i]
        if margin > 0:
            l_i = torch.clamp(l_i - margin, min=0)
        loss = loss + l_i
    return loss / batch_mask.sum()


# ---------------------------------------------------------------------------- util.py:1423-1430, 2110-2121
def normalize_dict_values(d):
    s = np.sum(list(d.values()))
    return d if s == 0 else {k: v / s for k, v in d.items()}


def normalized_sum(losses, norm_pow=0):
    total = sum(losses)
    if norm_pow == 0 or len(losses) == 0:
        return total
    scaled = sum(l / np.power(np.abs(float(l)) + 1e-8, norm_pow) for l in losses)
    return scaled * float(total) / (float(scaled) + 1e-8)


# ---------------------------------------------------------------------------- util.py:2037-2092
def calc_prompt_emb_delta_loss(static_embeddings, prompt_emb_mask, cls_delta_grad_scale=0.05):
    """static_embeddings [4*BS, 16, 77, 768] = (subj single, subj comp, class single, class comp) blocks;
    prompt_emb_mask [4*BS, 77, 1] (1 token / 0.5 padding).  The subject delta (comp - single, ortho-subtracted) should
    align with the class delta.  NOTE: like the reference this zeroes column 0 of ``prompt_emb_mask`` IN PLACE."""
    ss, sc, cs, cc = static_embeddings.chunk(4)
    weights = None
    if prompt_emb_mask is not None:
        prompt_emb_mask[:, 0] = 0
        m_ss, m_sc, _m_cs, _m_cc = prompt_emb_mask.chunk(4)
        weights = ((m_ss + m_sc).pow(2) / 4).unsqueeze(1)          # [BS,1,77,1]
    subj_delta = ortho_subtract(sc, ss)
    cls_delta = ortho_subtract(cc, cs)
    return calc_ref_cosine_loss(subj_delta, cls_delta, emb_mask=weights, do_demean_first=True,
                                first_n_dims_to_flatten=3, ref_grad_scale=cls_delta_grad_scale, aim_to_align=True)


# ---------------------------------------------------------------------------- ddpm.py:4259-4387
XLAYER_WEIGHTS = {8: 0.5, 12: 1., 16: 1., 17: 1., 18: 1., 19: 0.5, 20: 0.5, 21: 0.5, 22: 0.25, 23: 0.25, 24: 0.25}
XLAYER_BELOW = {8: 7, 12: 8, 16: 12, 17: 16, 18: 17, 19: 18, 20: 19, 21: 20, 22: 21, 23: 22, 24: 23}


def calc_fg_bg_xlayer_consist_loss(ca_attnscores, subj_indices, bg_indices, SSB_SIZE):
    """ca_attnscores {layer: [B, heads, N, 77]}; subj_indices / bg_indices = (batch idx, token idx) tuples listing, per
    instance, the K_fg / K_bg prompt positions of the subject / background embeddings.  Each aligned layer's head-mean,
    token-summed subject (background) score map is compared by demeaned cosine with the map of the layer below it,
    the larger one bilinearly resized to the smaller.  -> (loss_fg, loss_bg)"""
    weights = normalize_dict_values(dict(XLAYER_WEIGHTS))
    K_fg = len(subj_indices[0]) // len(torch.unique(subj_indices[0]))
    subj_indices = (subj_indices[0][:SSB_SIZE * K_fg], subj_indices[1][:SSB_SIZE * K_fg])
    if bg_indices is not None:
        K_bg = len(bg_indices[0]) // len(torch.unique(bg_indices[0]))
        bg_indices = (bg_indices[0][:SSB_SIZE * K_bg], bg_indices[1][:SSB_SIZE * K_bg])

    def token_map(mat, idx, K):                 # mat [B,77,heads,N] -> [SSB, N]
        return mat[idx].reshape(SSB_SIZE, K, *mat.shape[2:]).mean(dim=2).sum(dim=1)

    def resized(m, H, Hx):
        m = F.interpolate(m.reshape(SSB_SIZE, 1, H, H), size=(Hx, Hx), mode="bilinear", align_corners=False)
        return m.reshape(SSB_SIZE, Hx * Hx)

    fg, bg = [], []
    for layer, score in ca_attnscores.items():
        if layer not in weights:
            continue
        mat = score.permute(0, 3, 1, 2)
        mat_x = ca_attnscores[XLAYER_BELOW[layer]].permute(0, 3, 1, 2)
        if mat_x.shape[-1] > mat.shape[-1]:
            mat, mat_x = mat_x, mat
        H, Hx = int(np.sqrt(mat.shape[-1])), int(np.sqrt(mat_x.shape[-1]))
        s_map = resized(token_map(mat, subj_indices, K_fg), H, Hx)
        s_map_x = token_map(mat_x, subj_indices, K_fg)
        fg.append(calc_ref_cosine_loss(s_map, s_map_x, exponent=2, do_demean_first=True, first_n_dims_to_flatten=1,
                                       ref_grad_scale=1, aim_to_align=True) * weights[layer])
        if bg_indices is not None:
            b_map = resized(token_map(mat, bg_indices, K_bg), H, Hx)
            b_map_x = token_map(mat_x, bg_indices, K_bg)
            bg.append(calc_ref_cosine_loss(b_map, b_map_x, exponent=2, do_demean_first=True, first_n_dims_to_flatten=1,
                                           ref_grad_scale=1, aim_to_align=True) * weights[layer])
    return normalized_sum(fg), normalized_sum(bg)


# ---------------------------------------------------------------------------- util.py:1450-1466
def masked_mean(ts, mask, instance_weights=None, dim=None, keepdim=False):
    if instance_weights is None:
        instance_weights = 1
    if isinstance(instance_weights, torch.Tensor):
        instance_weights = instance_weights.view(list(instance_weights.shape) + [1] * (ts.ndim - instance_weights.ndim))
    if mask is None:
        return (ts * instance_weights).mean()
    mask = mask.expand(ts.shape)
    mask_sum = mask.sum(dim=dim, keepdim=keepdim)
    mask_sum = torch.maximum(mask_sum, torch.ones_like(mask_sum) * 1e-6)
    return (ts * instance_weights * mask).sum(dim=dim, keepdim=keepdim) / mask_sum


# ---------------------------------------------------------------------------- util.py:1570-1594
def resize_mask_for_feat_or_attn(feat_or_attn, mask, num_spatial_dims=1, mode="nearest|bilinear"):
    """mask [B,1,H,W] -> [B,1,h,h] with h = sqrt(spatial size of feat_or_attn): the larger of the nearest and the
    bilinear resize (a small subject must not vanish)."""
    h = int(np.sqrt(feat_or_attn.shape[-num_spatial_dims:].numel()))
    near = F.interpolate(mask.float(), size=(h, h), mode="nearest")
    if mode == "nearest|bilinear":
        return torch.maximum(near, F.interpolate(mask.float(), size=(h, h), mode="bilinear", align_corners=False))
    return near


# ---------------------------------------------------------------------------- util.py:1945-1977 (do_sum / do_sqrt_norm)
def sel_emb_attns_by_indices(attn_mat, indices, do_sum=True, do_sqrt_norm=False):
    """attn_mat [B, 77, heads, N]; indices (instance idx, token idx) -> per instance the rows of its listed tokens,
    summed over the tokens (optionally / sqrt(#tokens)) -> [n_instances, heads, N]."""
    inst, tok = indices
    out = []
    for b in torch.unique(inst).tolist():
        sel = inst == b
        rows = attn_mat[inst[sel], tok[sel]].unsqueeze(0)              # [1, K, heads, N]
        if do_sum:
            rows = rows.sum(dim=1)
        if do_sqrt_norm:
            rows = rows / np.sqrt(int(sel.sum()))
        out.append(rows)
    return torch.cat(out, dim=0)


# ---------------------------------------------------------------------------- ddpm.py:3932-4258
COMPLEM_WEIGHTS = {7: 0.5, 8: 0.5, 12: 1., 16: 1., 17: 1., 18: 1., 19: 1., 20: 1., 21: 1., 22: 1., 23: 1., 24: 1.}


def calc_fg_bg_complementary_loss(ca_attnscores, subj_indices, bg_indices, BLOCK_SIZE, fg_grad_scale=0.1, fg_mask=None,
                                  instance_mask=None, do_sqrt_norm=False):
    """-> (fg_bg_complementary, subj_mb_suppress, bg_mf_suppress, fg_bg_mask_contrast).  Per layer, from the per-head
    score maps of the subject tokens (sum over K_fg) and of the background tokens (sum over K_bg):
      * complementary: cosine_embedding(bg, subj*|subj|, label -1) -- the two maps should be orthogonal (subject side's
        gradient x fg_grad_scale);
      * with fg_mask: hinge terms (margin 0.4) keeping the subject's score on background pixels below its mean on
        foreground pixels (x0.05), the background tokens' score on foreground pixels below their mean on background
        pixels (x0.1), and the two cross contrasts (x0.05; margin 0.4*K_fg/K_bg for bg-vs-subject on the foreground).
    bg_indices None -> only the first hinge term (calc_fg_mb_suppress_loss, ddpm.py:3932-4040)."""
    if subj_indices is None:
        return 0, 0, 0, 0
    weights = normalize_dict_values(dict(COMPLEM_WEIGHTS))
    K_fg = len(subj_indices[0]) // len(torch.unique(subj_indices[0]))
    subj_indices = (subj_indices[0][:BLOCK_SIZE * K_fg], subj_indices[1][:BLOCK_SIZE * K_fg])
    have_bg = bg_indices is not None
    if have_bg:
        K_bg = len(bg_indices[0]) // len(torch.unique(bg_indices[0]))
    use_mask = (fg_mask is not None) and (instance_mask is None or instance_mask.sum() > 0)
    if not have_bg and not use_mask:
        return 0, 0, 0, 0
    margin = 0.4
    margin_subj_bg_mf = 0.4 * K_fg / K_bg if have_bg else None
    comple, subj_mb, bg_mf, contrast = [], [], [], []
    for layer, score in ca_attnscores.items():
        if layer not in weights:
            continue
        w = weights[layer]
        mat = score.permute(0, 3, 1, 2)
        subj = sel_emb_attns_by_indices(mat, subj_indices, do_sum=True, do_sqrt_norm=do_sqrt_norm)       # [BLOCK, heads, N]
        if have_bg:
            bg = sel_emb_attns_by_indices(mat, bg_indices, do_sum=True, do_sqrt_norm=do_sqrt_norm)
            comple.append(calc_ref_cosine_loss(bg, subj, exponent=2, do_demean_first=False, first_n_dims_to_flatten=2,
                                               ref_grad_scale=fg_grad_scale, aim_to_align=False) * w)
        if not use_mask:
            continue
        m = resize_mask_for_feat_or_attn(subj, fg_mask, num_spatial_dims=1, mode="nearest|bilinear")
        m = m.reshape(BLOCK_SIZE, 1, -1).repeat(1, subj.shape[1], 1)
        fgm = torch.zeros_like(m)
        fgm[m > 1e-6] = 1.
        bgm = 1 - fgm
        if (fgm.sum(dim=(1, 2)) == 0).any() or (bgm.sum(dim=(1, 2)) == 0).any():
            continue
        subj_at_mf = scale_gradient(subj * fgm, 0.5)
        subj_at_mb = subj * bgm
        avg_subj_mf = masked_mean(subj_at_mf, fgm, dim=(1, 2), keepdim=True)
        excess = subj_at_mb + margin - avg_subj_mf
        subj_mb.append(masked_mean(excess, excess > 0, instance_weights=instance_mask) * w * 0.05)
        if not have_bg:
            continue
        bg_at_mf, bg_at_mb = bg * fgm, bg * bgm
        avg_bg_mb = masked_mean(bg_at_mb, bgm, dim=(1, 2), keepdim=True)
        e2 = bg_at_mf + margin - avg_bg_mb
        bg_mf.append(masked_mean(e2, e2 > 0, instance_weights=instance_mask) * w * 0.1)
        e3 = bg_at_mf + margin_subj_bg_mf - avg_subj_mf
        e4 = subj_at_mb + margin - avg_bg_mb
        contrast.append((masked_mean(e3, e3 > 0, instance_weights=instance_mask)
                         + masked_mean(e4, e4 > 0, instance_weights=instance_mask)) * w * 0.05)
    return normalized_sum(comple), normalized_sum(subj_mb), normalized_sum(bg_mf), normalized_sum(contrast)

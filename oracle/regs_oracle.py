"""CPU restatement of the two regularisers the recon iteration adds to the masked MSE (ddpm.py:3207-3270).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline); the product path never imports it.

  * helpers of ldm/util.py -- IMPORTABLE from /root/reference, pinned by tests/golden/regs_*.npz:
      ``ortho_subtract`` util.py:280-316, ``demean`` :425-431, ``calc_ref_cosine_loss`` :437-535,
      ``gen_gradient_scaler`` / ``ScaleGrad`` :1084-1131, ``normalize_dict_values`` :1423-1430,
      ``normalized_sum`` :2110-2121, ``calc_prompt_emb_delta_loss`` :2037-2092
  * ``calc_fg_bg_xlayer_consist_loss`` ddpm.py:4259-4387 -- ddpm.py cannot be imported here (pytorch_lightning etc.,
    SURVEY 8c), restated from the source text on top of the pinned helpers; PARITY UNPINNED against reference outputs
    for the method itself, known-answer tests in tests/test_regs_oracle.py.
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

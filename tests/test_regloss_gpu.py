"""The fused attention regularisers (csrc/regloss.hip: values and gradients of calc_fg_bg_xlayer_consist_loss and
calc_fg_bg_complementary_loss in one C call) against the host expressions of the same losses (ldm/models/diffusion/ddpm.py --
themselves pinned by the reference's vectors in tests/test_ddpm_golden.py / test_regs_oracle.py) under autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYER_SIDE = {7: 16, 8: 16, 12: 8, 16: 16, 17: 16, 18: 16, 19: 32, 20: 32, 21: 32, 22: 64, 23: 64, 24: 64}


def _model(zero_shot, w_c=2e-4, w_x=5e-5):
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    m = object.__new__(LatentDiffusion)
    torch.nn.Module.__init__(m)
    m.fg_bg_complementary_loss_weight, m.fg_bg_xlayer_consist_loss_weight = w_c, w_x
    m.do_zero_shot, m.prompt_emb_delta_reg_weight, m.optimizer_type = zero_shot, 0.0, "Prodigy"
    return m


def _inputs(dev, Bt, have_bg, seed, side_scale=1):
    from adaprompt_amd.ldm.util import token_weight_matrix
    g = torch.Generator(device=dev).manual_seed(seed)
    subj = (torch.arange(Bt, device=dev).repeat_interleave(16), torch.arange(4, 20, device=dev).repeat(Bt))
    bg = (torch.arange(Bt, device=dev).repeat_interleave(4), torch.arange(24, 28, device=dev).repeat(Bt)) if have_bg else None
    groups = [subj] + ([bg] if have_bg else [])
    w = token_weight_matrix(groups, Bt, 77)
    maps, scores = {}, {}
    for li, s in LAYER_SIDE.items():
        s = s // side_scale
        n = s * s
        # smooth-ish positive maps plus noise, so that the hinges have both signs and the cosines are not degenerate
        t = torch.randn(Bt, 8, n, len(groups), device=dev, generator=g) * 0.6 + 0.3
        maps[li] = t.requires_grad_(True)
        scores[li] = torch.empty(1, device=dev).expand(Bt, 8, n, 77)           # only its shape is read when token maps exist
    return maps, scores, w, subj, bg


def _run(m, maps, scores, w, subj, bg, Bk, fg_mask, inst, fused):
    from adaprompt_amd.ldm.models.diffusion import ddpm
    ddpm.set_fused_reg_losses(fused)
    try:
        for t in maps.values():
            t.grad = None
        extra = {"subj_indices": subj, "bg_indices": bg, "ca_tokmap_weights": w,
                 "ca_layers_activations": {"attnscore": scores, "attnscore_tokmap": maps}}
        total, parts = m.recon_regularizers(extra, Bk, do_static_prompt_delta_reg=False, fg_mask=fg_mask, instance_mask=inst,
                                            do_complementary=True)
        if "reg_tokmap_grads" in extra:
            assert fused
            roots, grads = extra["reg_tokmap_grads"]
            torch.autograd.backward(roots, grads)
        else:
            assert not fused
            total.backward()
        torch.cuda.synchronize()
        return float(total), {k: float(v) for k, v in parts.items()}, {li: t.grad.clone() for li, t in maps.items()}
    finally:
        ddpm.set_fused_reg_losses(True)


def _fg_mask(dev, Bt, kind):
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64, device=dev), torch.linspace(-1, 1, 64, device=dev), indexing="ij")
    ms = []
    for b in range(Bt):
        r = 0.55 + 0.1 * b
        m = ((xx / r) ** 2 + ((yy + 0.1 * b) / (r + 0.1)) ** 2 <= 1.0).float()
        ms.append(m)
    fg = torch.stack(ms)[:, None]
    if kind == "tiny":                 # instance 1's foreground is one latent pixel off the coarse grids' sample points:
        fg[1] = 0                      # the 8 x 8 / 16 x 16 resized masks lose it -> those resolutions drop out (valid = 0)
        fg[1, 0, 5, 9] = 1
    return fg


@pytest.mark.parametrize("zero_shot,have_bg,mask,Bk,use_iw", [(True, True, "ellipse", 3, False), (False, True, "ellipse", 2, True),
                                                               (True, False, "ellipse", 3, False), (True, True, None, 3, False),
                                                               (True, True, "tiny", 3, False)])
def test_fused_regularisers_match_the_host_expressions(zero_shot, have_bg, mask, Bk, use_iw):
    dev = torch.device("cuda:0")
    Bt = 3
    m = _model(zero_shot)
    maps, scores, w, subj, bg = _inputs(dev, Bt, have_bg, 7)
    fg = None if mask is None else _fg_mask(dev, Bt, mask)
    inst = torch.tensor([1.0, 0.0, 1.0], device=dev) if use_iw else None
    tot_h, parts_h, g_h = _run(m, maps, scores, w, subj, bg, Bk, fg, inst, fused=False)
    tot_f, parts_f, g_f = _run(m, maps, scores, w, subj, bg, Bk, fg, inst, fused=True)
    assert set(parts_h) == set(parts_f), (parts_h.keys(), parts_f.keys())
    for k in parts_h:
        assert abs(parts_h[k] - parts_f[k]) <= 2e-5 * abs(parts_h[k]) + 1e-7, (k, parts_h[k], parts_f[k])
    assert abs(tot_h - tot_f) <= 2e-5 * abs(tot_h) + 1e-9, (tot_h, tot_f)
    num = sum(float((g_h[li] - g_f[li]).double().pow(2).sum()) for li in g_h)
    den = sum(float(g_h[li].double().pow(2).sum()) for li in g_h)
    assert den > 0 and np.sqrt(num / den) < 2e-4, np.sqrt(num / den)
    for li in g_h:                     # and layer by layer (a layer's gradient norm can be tiny next to the whole)
        d = float((g_h[li] - g_f[li]).norm()) / (float(g_h[li].norm()) + 1e-12)
        assert d < 1e-3, (li, d)


def test_fused_regularisers_are_bit_reproducible():
    dev = torch.device("cuda:0")
    m = _model(True)
    maps, scores, w, subj, bg = _inputs(dev, 2, True, 3)
    fg = _fg_mask(dev, 2, "ellipse")
    a = _run(m, maps, scores, w, subj, bg, 2, fg, None, fused=True)
    b = _run(m, maps, scores, w, subj, bg, 2, fg, None, fused=True)
    assert a[0] == b[0] and all(torch.equal(a[2][li], b[2][li]) for li in a[2])


@pytest.mark.parametrize("Bs,L,T,D", [(2, 16, 77, 768), (3, 4, 20, 100)])
def test_fused_prompt_delta_loss_matches_the_host_expression(Bs, L, T, D):
    """adap_prompt_delta_loss (value + gradient in one call) against ldm/util.py calc_prompt_emb_delta_loss under autograd -- the
    mirror that tests/test_regs_oracle.py holds to the reference's vectors."""
    from adaprompt_amd import ops
    from adaprompt_amd.ldm.util import calc_prompt_emb_delta_loss
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    emb = (torch.randn(4 * Bs, L, T, D, device=dev, generator=g) * 0.7).requires_grad_(True)
    mask = torch.full((4 * Bs, T, 1), 0.5, device=dev)
    for blk, n in enumerate((T // 3, T // 2, T // 3, T // 2)):
        mask[blk * Bs:(blk + 1) * Bs, :n] = 1.0
    mask[Bs + 1, 3:6] = 0.0
    coef = 2e-4 * 0.1
    m_h = mask.clone()
    loss_h = calc_prompt_emb_delta_loss(emb, m_h)
    (loss_h * coef).backward()
    g_h = emb.grad.clone()
    m_f = mask.clone()
    out, d_emb = ops.prompt_delta_loss(emb.detach(), m_f, coef)
    torch.cuda.synchronize()
    assert torch.equal(m_f, m_h) and float(m_f[:, 0].abs().sum()) == 0          # the reference's in-place start-token zeroing
    assert abs(float(out[0]) - float(loss_h)) <= 2e-5 * abs(float(loss_h)), (float(out[0]), float(loss_h))
    assert abs(float(out[1]) - coef * float(loss_h)) <= 2e-5 * coef * abs(float(loss_h))
    rel = float((d_emb - g_h).norm() / g_h.norm())
    assert rel < 2e-4, rel
    for blk in range(4):                    # each of the four blocks on its own (the class blocks carry 0.05 of the gradient)
        a, b = d_emb[blk * Bs:(blk + 1) * Bs], g_h[blk * Bs:(blk + 1) * Bs]
        assert float((a - b).norm() / (b.norm() + 1e-20)) < 5e-4, blk

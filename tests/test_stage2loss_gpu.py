"""csrc/stage2loss.hip: Stage 2's elastic matching loss as one C call each way (functional.ElasticMatchFn) against
(a) the reference's OWN numbers for it (tests/golden/ddpm_methods.npz, section H: values, both weight vectors, both gradients),
(b) the torch expressions of ``stage2.calc_elastic_matching_loss`` in float64 on the CPU at the shapes config 4 runs it on
    (N = 961 / 225 / 49 / 64 pooled tokens), with incoming gradients on all five outputs,
and bit-equality of two runs (the property the vendor-GEMM form did not have)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_ddpm as G          # noqa: E402   (input builders only)
from adaprompt_amd.ldm import stage2 as S          # noqa: E402

pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(ROOT, "tests", "golden", "ddpm_methods.npz"))


def dev():
    return torch.device("cuda:0")


def test_elastic_match_fused_against_the_reference_fixture():
    qe, fe = G.seeded((4, 10, 49), 210).to(dev()).requires_grad_(True), G.seeded((4, 14, 49), 211).to(dev()).requires_grad_(True)
    me = (torch.rand(1, 1, 49, generator=torch.Generator().manual_seed(212)) > 0.55).float().to(dev())
    assert S.ELASTIC_FUSED
    lm, lf, lb, scb, mcb = S.calc_elastic_matching_loss(qe, fe, me)
    assert np.allclose([float(v.detach()) for v in (lm, lf, lb)], FIX["s2/elastic/losses"], rtol=2e-5)
    assert np.allclose(np.stack([scb.detach().cpu().numpy(), mcb.detach().cpu().numpy()]), FIX["s2/elastic/below"], rtol=1e-5, atol=1e-7)
    (lm + lf + lb).backward()
    assert torch.allclose(qe.grad.cpu(), torch.from_numpy(FIX["s2/elastic/grad_q"]), rtol=2e-4, atol=1e-8)
    assert torch.allclose(fe.grad.cpu(), torch.from_numpy(FIX["s2/elastic/grad_f"]), rtol=2e-4, atol=1e-8)
    assert S.calc_elastic_matching_loss(qe, fe, torch.zeros(1, 1, 49, device=dev()))[3] is None


def _run(q, f, m, gw, fused):
    """-> (five outputs, dq, df) of sum(gw_k * output_k)."""
    q, f = q.clone().requires_grad_(True), f.clone().requires_grad_(True)
    old = S.ELASTIC_FUSED
    S.ELASTIC_FUSED = fused
    try:
        outs = S.calc_elastic_matching_loss(q, f, m)
    finally:
        S.ELASTIC_FUSED = old
    tot = sum((o * w).sum() for o, w in zip(outs, gw))
    tot.backward()
    return [o.detach() for o in outs], q.grad, f.grad


@pytest.mark.parametrize("Cq,Cf,N,qscale", [(320, 320, 961, 0.25), (640, 640, 225, 0.2), (1280, 1280, 49, 0.1), (1280, 1280, 64, 0.1),
                                            (40, 72, 130, 0.5)])
def test_elastic_match_fused_equals_the_torch_expressions_in_f64(Cq, Cf, N, qscale):
    g = torch.Generator().manual_seed(1000 + N)
    q = torch.randn(4, Cq, N, generator=g) * qscale
    f = torch.randn(4, Cf, N, generator=g)
    f[1] = 0.6 * f[1] + 0.4 * f[3]                                   # comp features of subject and mix correlate, as in the model
    m = (torch.rand(1, 1, N, generator=g) > 0.6).float()
    gw = [torch.tensor(0.7), torch.tensor(1.3), torch.tensor(0.9), torch.rand(1, 1, N, generator=g), torch.rand(1, 1, N, generator=g)]
    ref_out, ref_dq, ref_df = _run(q.double(), f.double(), m.double(), [w.double() for w in gw], fused=False)
    d = dev()
    out, dq, df = _run(q.to(d), f.to(d), m.to(d), [w.to(d) for w in gw], fused=True)
    for k in range(3):
        assert abs(float(out[k]) - float(ref_out[k])) <= 2e-5 * abs(float(ref_out[k])) + 1e-7, (k, float(out[k]), float(ref_out[k]))
    for k in (3, 4):
        assert torch.allclose(out[k].cpu().double(), ref_out[k], rtol=1e-4, atol=2e-6)
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert rel(dq, ref_dq) < 2e-4, rel(dq, ref_dq)
    assert rel(df, ref_df) < 2e-4, rel(df, ref_df)
    assert float(ref_dq.norm()) > 0 and float(ref_df[2].abs().max()) == 0 and float(df[2].abs().max()) == 0
    # the same call again: bit-equal values and gradients
    out2, dq2, df2 = _run(q.to(d), f.to(d), m.to(d), [w.to(d) for w in gw], fused=True)
    assert all(torch.equal(a, b) for a, b in zip(out, out2)) and torch.equal(dq, dq2) and torch.equal(df, df2)


# ---- the other per-layer terms: torch expressions (STAGE2_FUSED = False, float64 on the CPU) against the one-launch forms ---------
def _prompt_mix_inputs(H, N, C, hw, seed):
    g = torch.Generator().manual_seed(seed)
    score = torch.rand(4, hw * hw, 77, H, generator=g) * 0.2           # [4B, N, 77, heads] as captured; permuted inside
    score = score.permute(0, 3, 1, 2).contiguous()                     # ca_attnscores[li]: [4B, heads, N, 77]
    feat = torch.randn(4, C, hw, hw, generator=g)
    feat[1] = 0.7 * feat[1] + 0.3 * feat[3]
    feat[0] = 0.7 * feat[0] + 0.3 * feat[2]
    return score, feat


def _run_prompt_mix(ld_cls, score, feat, idx, li, fused):
    import adaprompt_amd.ldm.models.diffusion.ddpm as D
    score, feat = score.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    old = D.STAGE2_FUSED
    D.STAGE2_FUSED = fused
    try:
        outs = ld_cls.calc_prompt_mix_loss(None, {li: feat}, None, {li: score}, idx, 1)
    finally:
        D.STAGE2_FUSED = old
    (outs[0] * 0.9 + outs[1] * 1.1 + outs[2] * 0.7).backward()
    return [o.detach() for o in outs], score.grad, feat.grad


@pytest.mark.parametrize("H,C,hw,li", [(8, 320, 64, 23), (8, 640, 32, 20), (8, 1280, 16, 17), (8, 1280, 8, 12)])
def test_prompt_mix_terms_fused_equal_the_torch_expressions_in_f64(H, C, hw, li):
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    score, feat = _prompt_mix_inputs(H, hw * hw, C, hw, 3000 + hw)
    idx2 = (torch.tensor([0] * 4 + [1] * 4), torch.tensor([5, 6, 7, 8] * 2))           # subject tokens of the (single, comp) pair
    ref, ref_ds, ref_df = _run_prompt_mix(LatentDiffusion, score.double(), feat.double(), idx2, li, fused=False)
    d = dev()
    idx_d = tuple(t.to(d) for t in idx2)
    out, ds, df = _run_prompt_mix(LatentDiffusion, score.to(d), feat.to(d), idx_d, li, fused=True)
    for k in range(3):
        assert abs(float(out[k]) - float(ref[k])) <= 3e-5 * abs(float(ref[k])) + 1e-9, (k, float(out[k]), float(ref[k]))
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert rel(ds, ref_ds) < 3e-4, rel(ds, ref_ds)
    assert rel(df, ref_df) < 3e-4, rel(df, ref_df)
    out2, ds2, df2 = _run_prompt_mix(LatentDiffusion, score.to(d), feat.to(d), idx_d, li, fused=True)
    assert all(torch.equal(a, b) for a, b in zip(out, out2)) and torch.equal(ds, ds2) and torch.equal(df, df2)


@pytest.mark.parametrize("H,N", [(8, 961), (8, 225), (8, 49), (5, 64)])
def test_bg_suppress_fused_equals_masked_means(H, N):
    from adaprompt_amd import functional as HF
    from adaprompt_amd.ldm.util import gen_gradient_scaler, masked_mean
    g = torch.Generator().manual_seed(4000 + N)
    a = torch.randn(4, H, N, generator=g)
    scb, mcb = torch.rand(1, 1, N, generator=g).clamp(min=0.3) - 0.3, torch.rand(1, 1, N, generator=g).clamp(min=0.5) - 0.5

    def torch_form(a, scb, mcb):
        _, sc_a, _, mc_a = a.chunk(4)
        return masked_mean(sc_a.clamp(min=0), scb), masked_mean(gen_gradient_scaler(0.02)(mc_a).clamp(min=0), mcb)

    ins = [t.double().requires_grad_(True) for t in (a, scb, mcb)]
    r = torch_form(*ins)
    (r[0] * 0.8 + r[1] * 1.2).backward()
    d = dev()
    ind = [t.to(d).requires_grad_(True) for t in (a, scb, mcb)]
    o = HF.BgSuppressFn.apply(*ind, 0.02)
    (o[0] * 0.8 + o[1] * 1.2).backward()
    for k in range(2):
        assert abs(float(o[k].detach()) - float(r[k].detach())) <= 1e-5 * abs(float(r[k].detach())) + 1e-9
    for x, y in zip(ind, ins):
        assert float((x.grad.cpu().double() - y.grad).norm() / y.grad.norm()) < 1e-5


# ---- edges -----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,hw", [(8, 64), (8, 32), (5, 16), (1, 8)])
def test_attn_spatial_weight_equals_convert_attn_to_spatial_weight(H, hw):
    from adaprompt_amd import ops
    g = torch.Generator().manual_seed(5000 + hw)
    a0 = torch.rand(1, H, hw * hw, generator=g) * 0.3 + 0.01
    a1 = torch.rand(1, H, hw * hw, generator=g) * 0.1
    a1[..., : hw] += 2.0                                   # a map with a strong peak: the exp is clamped to 1 elsewhere
    for rev in (True, False):
        w0, _ = S.convert_attn_to_spatial_weight(a0.double(), 1, torch.Size([hw, hw]), reversed=rev)
        w1, _ = S.convert_attn_to_spatial_weight(a1.double(), 1, torch.Size([hw, hw]), reversed=rev)
        ref_pair, ref_one = ((w0 + w1) / 2).reshape(-1), w1.reshape(-1)
        d = dev()
        out_pair = ops.attn_spatial_weight(a0[0].to(d).contiguous(), a1[0].to(d).contiguous(), reversed=rev)
        out_one = ops.attn_spatial_weight(a1[0].to(d).contiguous(), None, reversed=rev)
        assert torch.allclose(out_pair.cpu().double(), ref_pair, rtol=2e-5, atol=1e-7)
        assert torch.allclose(out_one.cpu().double(), ref_one, rtol=2e-5, atol=1e-7)
        assert abs(float(out_one.mean()) - 1.0) < 1e-5


def test_bg_suppress_with_empty_masks_is_zero_with_finite_gradients():
    from adaprompt_amd import functional as HF
    d = dev()
    a = torch.randn(4, 8, 225, device=d, requires_grad=True)
    z = torch.zeros(1, 1, 225, device=d, requires_grad=True)
    w = torch.rand(1, 1, 225, device=d).requires_grad_(True)
    ls, lm = HF.BgSuppressFn.apply(a, z, w, 0.02)
    (ls + lm).backward()
    assert float(ls.detach()) == 0.0 and float(lm.detach()) > 0.0
    for t in (a, z, w):
        assert torch.isfinite(t.grad).all()
    assert float(a.grad[0].abs().max()) == 0 and float(a.grad[2].abs().max()) == 0 and float(a.grad[1].abs().max()) == 0


@pytest.mark.parametrize("nfg", [1, 49])
def test_elastic_match_with_one_and_with_all_foreground_tokens(nfg):
    N, C = 49, 24
    g = torch.Generator().manual_seed(6000 + nfg)
    q, f = torch.randn(4, C, N, generator=g) * 0.3, torch.randn(4, C, N, generator=g)
    m = torch.zeros(1, 1, N)
    m[..., :nfg] = 1.0
    gw = [torch.tensor(1.0), torch.tensor(1.0), torch.tensor(1.0), torch.rand(1, 1, N, generator=g), torch.rand(1, 1, N, generator=g)]
    ref_out, ref_dq, ref_df = _run(q.double(), f.double(), m.double(), [w.double() for w in gw], fused=False)
    d = dev()
    out, dq, df = _run(q.to(d), f.to(d), m.to(d), [w.to(d) for w in gw], fused=True)
    for k in range(3):
        assert abs(float(out[k]) - float(ref_out[k])) <= 2e-5 * abs(float(ref_out[k])) + 1e-7, (k, float(out[k]), float(ref_out[k]))
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert rel(dq, ref_dq) < 2e-4 and rel(df, ref_df) < 2e-4
    assert torch.isfinite(dq).all() and torch.isfinite(df).all()

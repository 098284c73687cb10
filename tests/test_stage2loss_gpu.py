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

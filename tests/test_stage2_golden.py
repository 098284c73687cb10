"""Stage-2 (compositional distillation) host arithmetic -- ``adaprompt_amd.ldm.stage2`` and the two loss methods of
``LatentDiffusion`` -- against numbers produced by the reference's OWN ``ldm/util.py`` functions and ``ddpm.py`` methods
(tests/golden/make_golden_ddpm.py, section H): K/V prompt mixing with its gradient, elastic matching, delta alignment,
attention-derived spatial weights, the foreground-initialised latent (host RNG order included), calc_prompt_mix_loss,
calc_comp_fg_bg_preserve_loss (values and per-layer gradient norms) and the teacher selection."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_ddpm as G          # noqa: E402   (input builders only)
from adaprompt_amd.ldm import stage2 as S          # noqa: E402
from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion          # noqa: E402

FIX = np.load(os.path.join(ROOT, "tests", "golden", "ddpm_methods.npz"))


def T(name):
    return torch.from_numpy(FIX[name])


def test_mix_static_vk_embeddings_value_and_gradient():
    emb = G.seeded((2 * 2 * 16, 77, 12), 200, 0.3).requires_grad_(True)
    for tag, kw in (("a", dict(training_percent=0.25, t_frac=torch.tensor([0.9, 0.85]), K=[1.0, 0.8], V=[1.0, 0.6])),
                    ("b", dict(training_percent=0.8, t_frac=torch.tensor([0.55, 0.95]), K=[1.0, 1.0], V=[1.0, 0.7]))):
        emb.grad = None
        out, _, v_sc, _, k_sc = S.mix_static_vk_embeddings(emb, torch.arange(4, 13), kw["training_percent"], t_frac=kw["t_frac"],
                                                           K_CLS_SCALE_LAYERWISE_RANGE=kw["K"], V_CLS_SCALE_LAYERWISE_RANGE=kw["V"])
        assert tuple(out.shape) == (64, 154, 12)
        assert torch.allclose(out.detach(), T(f"s2/mixvk/{tag}/out"), rtol=1e-6, atol=1e-7)
        (out * G.seeded(tuple(out.shape), 201)).sum().backward()
        assert torch.allclose(emb.grad, T(f"s2/mixvk/{tag}/grad"), rtol=1e-5, atol=1e-7)
        assert np.allclose(np.stack([v_sc.numpy(), k_sc.numpy()]), FIX[f"s2/mixvk/{tag}/scales"], rtol=1e-6)
    assert np.array_equal(S.gen_cfg_scales_for_stu_tea(6, 5, 2, "cpu").numpy(), FIX["s2/cfg_scales"])
    assert np.allclose([S.calc_dyn_loss_scale(torch.tensor(v), 0.2, 2, 1, 3) for v in (0.05, 0.3, 0.9)], FIX["s2/dyn_scale"])


def test_elastic_matching_delta_alignment_spatial_weight():
    qe, fe = G.seeded((4, 10, 49), 210).requires_grad_(True), G.seeded((4, 14, 49), 211).requires_grad_(True)
    me = (torch.rand(1, 1, 49, generator=torch.Generator().manual_seed(212)) > 0.55).float()
    lm, lf, lb, scb, mcb = S.calc_elastic_matching_loss(qe, fe, me)
    assert np.allclose([float(lm), float(lf), float(lb)], FIX["s2/elastic/losses"], rtol=2e-5)
    assert np.allclose(np.stack([scb.detach().numpy(), mcb.detach().numpy()]), FIX["s2/elastic/below"], rtol=1e-5, atol=1e-7)
    (lm + lf + lb).backward()
    assert torch.allclose(qe.grad, T("s2/elastic/grad_q"), rtol=2e-4, atol=1e-8)
    assert torch.allclose(fe.grad, T("s2/elastic/grad_f"), rtol=2e-4, atol=1e-8)
    assert S.calc_elastic_matching_loss(qe, fe, torch.zeros(1, 1, 49))[3] is None
    fb, fx, rb, rx = (G.seeded((1, 2, 64), 220 + i).requires_grad_(True) for i in range(4))
    dl = S.calc_delta_alignment_loss(fb, fx, rb, rx, ref_grad_scale=0.05, feat_base_grad_scale=1, cosine_exponent=3,
                                     delta_types=["feat_to_ref"])["feat_to_ref"]
    dl.backward()
    assert abs(float(dl) - float(FIX["s2/delta/loss"])) < 1e-6
    assert np.allclose(G.grad_norms([fb, fx, rb, rx]), FIX["s2/delta/gnorm"], rtol=2e-4)
    sw, sa = S.convert_attn_to_spatial_weight(G.seeded((2, 2, 256), 230, 1.0) + 0.5, 1, torch.Size([32, 32]), reversed=True)
    assert torch.allclose(sw, T("s2/spatial_weight"), rtol=1e-5, atol=1e-6) and torch.allclose(sa, T("s2/spatial_attn"), rtol=1e-5)


def test_init_x_with_fg_from_training_image_and_rng_order():
    for tag, pct in (("big", 0.6), ("small", 0.25)):
        np.random.seed(77)
        torch.manual_seed(77)
        x0 = G.seeded((2, 4, 64, 64), 240)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
        fgm = ((xx / pct) ** 2 + (yy / pct) ** 2 <= 1.0).float()[None, None].repeat(2, 1, 1, 1)
        xi, m1, _ = S.init_x_with_fg_from_training_image(x0, fgm, fgm.clone(), 0.4, base_scale_range=(0.7, 1.0),
                                                         fg_noise_anneal_mean_range=(0.1, 0.4))
        assert torch.allclose(xi, T(f"s2/initx/{tag}/x"), rtol=1e-6, atol=1e-6)
        assert torch.allclose(m1, T(f"s2/initx/{tag}/fg"), rtol=1e-6, atol=1e-7)
        assert np.allclose([float(np.random.rand()), float(torch.rand(1))], FIX[f"s2/initx/{tag}/after"])      # same draws consumed


def _leaves():
    outfeat, score, qq, subj_1b, subj_2b, fg4 = G.stage2_case()
    lv = {k: {li: v.clone().requires_grad_(True) for li, v in dct.items()} for k, dct in
          (("outfeat", outfeat), ("score", score), ("q", qq))}
    return lv, subj_1b, subj_2b, fg4


def test_calc_prompt_mix_loss():
    lv, _, subj_2b, _ = _leaves()
    l_feat, l_delta, l_norm = LatentDiffusion.calc_prompt_mix_loss(None, lv["outfeat"], None, lv["score"], subj_2b, 1)
    assert np.allclose([float(l_feat), float(l_delta), float(l_norm)], FIX["s2/prompt_mix/losses"], rtol=2e-5)
    (l_feat + l_delta + l_norm).backward()
    assert np.allclose(G.grad_norms([lv["outfeat"][li] for li in G.STAGE2_LAYERS]), FIX["s2/prompt_mix/gnorm_outfeat"], rtol=3e-4)
    assert np.allclose(G.grad_norms([lv["score"][li] for li in G.STAGE2_LAYERS]), FIX["s2/prompt_mix/gnorm_score"], rtol=3e-4)


def test_calc_comp_fg_bg_preserve_loss():
    lv, subj_1b, _, fg4 = _leaves()
    ls = LatentDiffusion.calc_comp_fg_bg_preserve_loss(None, lv["outfeat"], None, lv["q"], None, lv["score"], fg4, torch.ones(4),
                                                       subj_1b, 1)
    assert np.allclose([float(l) for l in ls], FIX["s2/preserve/losses"], rtol=3e-5, atol=1e-8)
    sum(l for l in ls if torch.is_tensor(l)).backward()
    for k in ("outfeat", "score", "q"):
        assert np.allclose(G.grad_norms([lv[k][li] for li in G.STAGE2_LAYERS]), FIX[f"s2/preserve/gnorm_{k}"], rtol=5e-4, atol=1e-10), k
    lv, subj_1b, _, fg4 = _leaves()
    ls0 = LatentDiffusion.calc_comp_fg_bg_preserve_loss(None, lv["outfeat"], None, lv["q"], None, lv["score"], fg4, torch.zeros(4),
                                                        subj_1b, 1)
    assert np.allclose([float(l) for l in ls0], FIX["s2/preserve/no_mask"])


def test_teacher_selection():
    want = json.loads(str(FIX["s2/select"]))
    for losses, (teach, best, _colors) in zip(([0.30, 0.27, 0.26, 0.20], [0.25, 0.26, 0.27, 0.29], [0.31, 0.30, 0.279, 0.2795]), want):
        t, b = S.select_teacher(torch.tensor(losses))
        assert [bool(v) for v in t.tolist()] == teach and b == best

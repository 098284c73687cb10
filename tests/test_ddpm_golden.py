"""The oracle's restatements of ``ldm/models/diffusion/ddpm.py`` -- and the product's host-side mirrors -- against numbers the
reference's OWN methods produced (tests/golden/make_golden_ddpm.py -> ddpm_methods.npz): schedule buffers, q_sample /
predict_start_from_noise, calc_recon_loss (the north-star scalar) with its gradient, the recon iteration's attention losses
(values and gradient norms per layer, four switch combinations), LatentDiffusion.forward's conditioning assembly (four
iteration types) and the Arc2Face teacher rollout's timestep / x0 chain.  Inputs are re-drawn from the same seeds here."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_ddpm as G           # noqa: E402  (only its input builders: nothing of the reference is imported here)
from oracle import distill_oracle as DO, ldm_oracle as O, regs_oracle as R          # noqa: E402

FIX = np.load(os.path.join(ROOT, "tests", "golden", "ddpm_methods.npz"))


def T(name):
    return torch.from_numpy(FIX[name])


def test_schedule_buffers_oracle_and_product():
    from adaprompt_amd.ldm.models.diffusion.ddpm import DDPM
    s = O.make_schedule()
    ddpm = nn_free_ddpm()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        want = T("sched/" + k)
        assert torch.equal(getattr(ddpm, k), want), k                     # product: bit-equal to the reference's buffers
        if k in s:
            assert torch.equal(s[k], want), k
    assert DDPM is type(ddpm)


def nn_free_ddpm():
    """the product's DDPM schedule without building a UNet."""
    import torch.nn as nn
    from adaprompt_amd.ldm.models.diffusion.ddpm import DDPM
    d = nn.Module.__new__(DDPM)
    nn.Module.__init__(d)
    DDPM.register_schedule(d, None, "linear", 1000, 0.00085, 0.012)
    return d


def test_q_sample_and_predict_x0():
    s = O.make_schedule()
    x0, nz = G.seeded((4, 4, 8, 8), 1), G.seeded((4, 4, 8, 8), 2)
    t = torch.tensor([0, 17, 500, 999])
    xt = O.q_sample(s, x0, t, nz)
    assert torch.allclose(xt, T("q_sample"), rtol=0, atol=0)
    assert torch.allclose(O.predict_start_from_noise(s, xt, t, nz * 0.9 + 0.05), T("predict_x0"), rtol=1e-6, atol=1e-6)
    d = nn_free_ddpm()
    assert torch.allclose(d.predict_start_from_noise(xt, t, nz * 0.9 + 0.05), T("predict_x0"), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("i", range(4))
def test_calc_recon_loss_value_and_gradient(i):
    fw, bw, use_masks = FIX[f"recon/{i}/cfg"]
    out, tgt, img, fg = G.recon_case(10 + 10 * i)
    out.requires_grad_(True)
    loss, _ = O.calc_recon_loss(out, tgt, img if use_masks else None, fg if use_masks else None, float(fw), float(bw))
    loss.backward()
    assert abs(float(loss) - float(FIX[f"recon/{i}/loss"])) < 1e-7 * max(1.0, abs(float(loss)))
    assert torch.allclose(out.grad, T(f"recon/{i}/grad"), rtol=1e-6, atol=1e-9)


def _grad_norms(leaf):
    return np.array([0.0 if leaf[li].grad is None else float(leaf[li].grad.double().norm()) for li in G.ATTN_N])


@pytest.mark.parametrize("tag", ["full", "nobg", "nomask", "inst"])
@pytest.mark.parametrize("who", ["oracle", "product"])
def test_fg_bg_complementary_loss(tag, who):
    sc, subj, bg, fg = G.attn_case()
    kw = {"full": (bg, fg, None, False), "nobg": (None, fg, None, False), "nomask": (bg, None, None, False),
          "inst": (bg, fg, torch.tensor([1.0, 0.0]), True)}[tag]
    leaf = {li: v.clone().requires_grad_(True) for li, v in sc.items()}
    if who == "oracle":
        losses = R.calc_fg_bg_complementary_loss(leaf, subj, kw[0], 2, fg_grad_scale=0.1, fg_mask=kw[1], instance_mask=kw[2],
                                                 do_sqrt_norm=kw[3])
    else:
        from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
        losses = LatentDiffusion.calc_fg_bg_complementary_loss(None, leaf, subj, kw[0], 2, fg_grad_scale=0.1, fg_mask=kw[1],
                                                               instance_mask=kw[2], do_sqrt_norm=kw[3])
    want = FIX[f"complem/{tag}/losses"]
    got = np.array([float(l) for l in losses])
    assert np.allclose(got, want, rtol=2e-5, atol=1e-7), (got, want)
    tot = sum(l for l in losses if torch.is_tensor(l))
    if torch.is_tensor(tot) and tot.requires_grad:
        tot.backward()
    assert np.allclose(_grad_norms(leaf), FIX[f"complem/{tag}/gnorm"], rtol=2e-4, atol=1e-9)


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_fg_bg_xlayer_consist_loss(who):
    sc, subj, bg, _ = G.attn_case()
    leaf = {li: v.clone().requires_grad_(True) for li, v in sc.items()}
    if who == "oracle":
        fn = R.calc_fg_bg_xlayer_consist_loss
    else:
        from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
        fn = lambda *a: LatentDiffusion.calc_fg_bg_xlayer_consist_loss(None, *a)        # noqa: E731
    lfg, lbg = fn(leaf, subj, bg, 2)
    assert np.allclose([float(lfg), float(lbg)], FIX["xlayer/losses"], rtol=2e-5)
    (lfg + lbg).backward()
    assert np.allclose(_grad_norms(leaf), FIX["xlayer/gnorm"], rtol=2e-4, atol=1e-9)
    lfg1, lbg1 = fn({li: v.clone() for li, v in sc.items()}, subj, None, 1)
    assert np.allclose([float(lfg1), float(lbg1)], FIX["xlayer_nobg/losses"], rtol=2e-5)


@pytest.mark.parametrize("mode", ["recon_delta", "recon_plain", "mix", "ada_delta"])
def test_forward_conditioning_assembly(mode):
    """``assemble_conditioning`` (the mirror of ddpm.py:1940-2179) on the same stand-in text encoder / embedding manager
    the reference's ``forward`` ran on: context, prompts handed on, every extra_info key and all the index bookkeeping."""
    from adaprompt_amd.ldm.models.diffusion.conditioning import ConditioningMixin
    from tests.stubs import StubEmbeddingManager, StubTextEncoder
    enc = StubTextEncoder(dim=16)
    em = StubEmbeddingManager(text_embedder=enc, dim=16)
    me = types.SimpleNamespace()
    me.N_CA_LAYERS, me.prompt_mix_scheme, me.apply_arc2face_inverse_embs, me.cached_inits = 16, "mix_hijk", False, {}
    prompts = G.prompts_case(3)
    me.iter_flags = {"do_static_prompt_delta_reg": mode != "recon_plain", "do_mix_prompt_distillation": mode == "mix",
                     "do_ada_prompt_delta_reg": mode in ("mix", "ada_delta"), "do_normal_recon": mode.startswith("recon"),
                     "do_arc2face_distill": False, "reuse_init_conds": False, "delta_prompts": prompts,
                     "zs_clip_features": None, "zs_id_embs": None}

    def glc(cond_in, zs_clip_features=None, zs_id_embs=None, randomize_clip_weights=False, apply_arc2face_inverse_embs=False):
        emb = enc.encode(cond_in, embedding_manager=em)
        return emb, cond_in, {"placeholder2indices": dict(em.placeholder2indices), "prompt_emb_mask": em.prompt_emb_mask}
    me.get_learned_conditioning = glc
    c_emb, c_in, extra = ConditioningMixin.assemble_conditioning(me, list(prompts[0]), 3)
    assert torch.equal(c_emb.detach(), T(f"fwd/{mode}/c_emb"))
    assert list(c_in) == list(FIX[f"fwd/{mode}/c_in"])
    assert sorted(extra.keys()) == list(FIX[f"fwd/{mode}/keys"]), (sorted(extra.keys()), list(FIX[f"fwd/{mode}/keys"]))
    assert extra["iter_type"] == str(FIX[f"fwd/{mode}/iter_type"])
    for name in ("placeholder2indices", "placeholder2indices_1b", "placeholder2indices_2b"):
        have = [k for k in FIX.files if k.startswith(f"fwd/{mode}/{name}/")]
        assert (name in extra) == bool(have), name
        for k in have:
            ib, it = extra[name][k.rsplit("/", 1)[1]]
            assert np.array_equal(np.stack([ib.numpy(), it.numpy()]), FIX[k]), k
    for name in ("c_static_emb_4b", "c_static_emb_1b"):
        assert (name in extra) == (f"fwd/{mode}/{name}" in FIX.files)
        if name in extra:
            assert torch.equal(extra[name].detach(), T(f"fwd/{mode}/{name}")), name


@pytest.mark.parametrize("nd", [1, 3, 5])
def test_teacher_rollout_schedule(nd):
    """the oracle's rollout (and, through it, every test that pins the product's Arc2FaceWrapper against the oracle) against
    the reference's own ``Arc2FaceWrapper.forward`` with a closed-form eps model: same draws (rand_like, then randn_like,
    per step), same timesteps, same x0 chain."""
    s = O.make_schedule()
    x0, nz = G.seeded((2, 4, 8, 8), 50 + nd), G.seeded((2, 4, 8, 8), 60 + nd)
    t = torch.tensor([900, 431])
    ctx = G.seeded((2, 21, 16), 70 + nd, 0.1)
    torch.manual_seed(40 + nd)
    rel, noises = [], [nz]
    for _ in range(nd - 1):
        rel.append(torch.rand(2))
        noises.append(torch.randn(2, 4, 8, 8))

    def eps_fn(x, tt, c):
        return 0.3 * x + 0.01 * tt.view(-1, 1, 1, 1).float() / 1000 + c.mean()
    preds, x0s, ns, ts = DO.arc2face_rollout(eps_fn, s, x0, nz, t, ctx, num_denoising_steps=nd, relative_ts=rel, noises=noises)
    assert np.array_equal(torch.stack(ts).numpy(), FIX[f"rollout/{nd}/ts"])
    assert torch.allclose(preds[-1], T(f"rollout/{nd}/pred_last"), rtol=1e-5, atol=1e-6)
    assert torch.allclose(x0s[-1], T(f"rollout/{nd}/x0_last"), rtol=1e-5, atol=1e-5)
    assert torch.equal(ns[-1], T(f"rollout/{nd}/noise_last"))


def test_shared_step_front_flags_and_rng_order():
    """``prepare_recon_iteration`` (the mirror of the front of the reference's ``shared_step``, ddpm.py:1436-1938) on the
    batches and seeds the reference's own ``shared_step`` ran on: every iteration flag, the prompt lists picked, the batch
    trimming of multi-step distillation, what the embedding manager is told -- and both host RNG streams end in the same
    state, i.e. ``random`` / ``np.random`` were consumed in the same order and number."""
    import json
    import random
    from adaprompt_amd.ldm.models.diffusion.conditioning import ConditioningMixin
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from tests.stubs import StubEmbeddingManager
    cases = json.loads(str(FIX["shared_step/cases"]))
    assert len(cases) == len(G.SHARED_STEP_CASES)
    seen = set()
    for (kind, distill, seed), want in zip(G.SHARED_STEP_CASES, cases):
        me = types.SimpleNamespace()
        me.training_percent, me.do_static_prompt_delta_reg, me.use_background_token, me.do_zero_shot = 0.3, True, True, True
        me.p_gen_arc2face_rand_face, me.p_add_noise_to_real_id_embs, me.max_num_denoising_steps = 0.4, 0.6, 5
        me.apply_arc2face_inverse_embs = False
        me.embedding_manager = StubEmbeddingManager(dim=8)
        me.draw_num_denoising_steps = LatentDiffusion.draw_num_denoising_steps
        me.half_batch_size = LatentDiffusion.half_batch_size
        me.zero_shot_features = lambda *a, **k: ConditioningMixin.zero_shot_features(me, *a, **k)
        me.arc2face = types.SimpleNamespace(gen_arc2face_prompt_embs=lambda n, pre_face_embs=None: (
            n, pre_face_embs if pre_face_embs is not None else G.seeded((n, 512), 8), G.seeded((n, 21, 8), 10)))
        me.do_static_prompt_delta_reg = True
        LatentDiffusion.init_iteration_flags(me)
        me.iter_flags["do_arc2face_distill"] = distill
        if distill:
            me.iter_flags["do_static_prompt_delta_reg"] = False
        batch = G.shared_step_batch(4)
        x0 = G.seeded((4, 4, 2, 2), 9)
        img_mask = torch.nn.functional.interpolate(batch["aug_mask"][:, None], size=(2, 2), mode="nearest")
        fg_mask = torch.nn.functional.interpolate(batch["fg_mask"][:, None], size=(2, 2), mode="nearest")
        if kind == "compos_fp":
            for k in list(batch.keys()):
                if k.startswith(("subj_prompt", "cls_prompt")):
                    base, bg = (k[:-3], "_bg") if k.endswith("_bg") else (k, "")
                    batch[base + "_fp" + bg] = ["a face portrait of " + q for q in batch[k]]
        if kind.startswith("compos"):
            me.iter_flags.update(do_mix_prompt_distillation=True, do_ada_prompt_delta_reg=True, is_compos_iter=True,
                                 calc_clip_loss=True, do_normal_recon=False)
            me.use_fp_trick, me.do_clip_teacher_filtering, me.cached_inits = True, True, {}
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        prep = ConditioningMixin.prepare_compos_iteration if kind.startswith("compos") else ConditioningMixin.prepare_recon_iteration
        x_start, im, fm, captions = prep(me, batch, x0, img_mask, fg_mask)
        got = G.flags_record(me.iter_flags, x_start, captions, me.embedding_manager.calls)
        got["after"] = [random.random(), float(np.random.rand())]
        for k, v in want.items():
            g = got[k]
            if k == "embman_names":
                g = [list(g[0]), g[1]]
            assert g == v, (kind, seed, k, g, v)
        seen.add((want["gen_arc2face_rand_face"], want["add_noise_to_real_id_embs"], want["use_arc2face_as_target"],
                  want["num_denoising_steps"] > 1))
    assert len(seen) >= 4          # the seeds cover random faces, noised ids, both targets, single- and multi-step

"""Golden vectors from the reference's OWN ``ldm/models/diffusion/ddpm.py`` methods (SURVEY.md 8c said the file "cannot be
imported": it can, once the third-party names it only mentions at import time are present -- pytorch_lightning, insightface,
cv2, clip, kornia, taming, diffusers, torchvision, a removed ``transformers`` name -- as inert stand-ins; no arithmetic of the
reference is replaced).  The methods are called UNBOUND on a bare object that carries only the attributes they read, so no
checkpoint / CLIP / Lightning is needed:

    DDPM.register_schedule, q_sample, predict_start_from_noise        (ddpm.py:240-292, 416-419, 358-362)
    LatentDiffusion.calc_recon_loss                                   (:3571-3595)
    LatentDiffusion.calc_fg_bg_complementary_loss / calc_fg_mb_suppress_loss / calc_fg_bg_xlayer_consist_loss  (:3932-4387)
    LatentDiffusion.forward's conditioning assembly                   (:1940-2179), with tests/stubs.py as the text encoder
    Arc2FaceWrapper.forward's rollout schedule                        (:5432-5478), a closed-form "UNet"

    python tests/golden/make_golden_ddpm.py          # writes tests/golden/ddpm_methods.npz  (own process: sys.modules games)

tests/test_ddpm_golden.py holds the oracle restatements AND the product mirrors to these numbers."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True


def _no_breakpoints(*a, **k):
    raise RuntimeError("the reference reached a breakpoint() (its way of asserting): the inputs of this case are not valid for it")


sys.breakpointhook = _no_breakpoints        # never wait on stdin
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "ddpm_methods.npz")


def import_reference_ddpm():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    sys.path.insert(0, REF)
    stub("cv2")
    import adaface.subj_basis_generator  # noqa: F401   (before the torchvision stand-in: transformers probes torchvision.__spec__)
    del sys.modules["ldm"]               # subj_basis_generator.py:23 rebinds it to the adaface package
    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils", make_grid=None, draw_bounding_boxes=None)
    tv.transforms = stub("torchvision.transforms")
    pl = stub("pytorch_lightning", LightningModule=nn.Module)
    pl.utilities = stub("pytorch_lightning.utilities")
    pl.utilities.distributed = stub("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    stub("insightface")
    stub("insightface.app", FaceAnalysis=object)
    stub("clip")
    stub("kornia")
    stub("taming")
    stub("taming.modules")
    stub("taming.modules.vqvae")
    stub("taming.modules.vqvae.quantize", VectorQuantizer2=object)
    oc = stub("omegaconf")
    oc.listconfig = stub("omegaconf.listconfig", ListConfig=list)
    stub("diffusers", UNet2DConditionModel=object)
    stub("evaluation")
    stub("evaluation.clip_eval", CLIPEvaluator=object)
    import transformers
    if not hasattr(transformers, "ViTFeatureExtractor"):
        transformers.ViTFeatureExtractor = object        # removed upstream; ddpm.py:24 only names it
    import ldm.models.diffusion.ddpm as D
    assert D.__file__.startswith(REF)
    return D


def seeded(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# ---- shared with tests/test_ddpm_golden.py: the inputs of every case, by seed ------------------------------------------
ATTN_N = {7: 64, 8: 64, 12: 16, 16: 64, 17: 64, 18: 64, 19: 256, 20: 256, 21: 256, 22: 1024, 23: 1024, 24: 1024}


def attn_case(B=2, heads=2, seed=100):
    sc = {li: seeded((B, heads, n, 77), seed + li, 1.5) for li, n in ATTN_N.items()}
    subj = (torch.arange(B).repeat_interleave(9), torch.arange(4, 13).repeat(B))
    bg = (torch.arange(B).repeat_interleave(4), torch.arange(20, 24).repeat(B))
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 32), torch.linspace(-1, 1, 32), indexing="ij")
    fg = ((xx / 0.6) ** 2 + (yy / 0.7) ** 2 <= 1.0).float()[None, None].repeat(B, 1, 1, 1)
    return sc, subj, bg, fg


def recon_case(seed):
    out, tgt = seeded((3, 4, 16, 16), seed), seeded((3, 4, 16, 16), seed + 1)
    img = (torch.rand(3, 1, 16, 16, generator=torch.Generator().manual_seed(seed + 2)) > 0.2).float()
    fg = (torch.rand(3, 1, 16, 16, generator=torch.Generator().manual_seed(seed + 3)) > 0.6).float()
    return out, tgt, img, fg


def prompts_case(B=3):
    subj_single = [f"a photo of a z , , , , , , , , and y , , , number{i}" for i in range(B)]
    subj_comp = [p + " riding a horse in the snow" for p in subj_single]
    cls_single = [p.replace(" z ", " person ") for p in subj_single]
    cls_comp = [p.replace(" z ", " person ") for p in subj_comp]
    return subj_single, subj_comp, cls_single, cls_comp


def shared_step_batch(B=4):
    """a batch dict with the keys ``shared_step`` reads (ldm/data/personalized.py:510-833), prompts as in ``prompts_case``."""
    ss, sc, cs, cc = prompts_case(B)
    bgify = lambda ps: [p + " with background y" for p in ps]          # noqa: E731
    return {"subject_name": ["alice", "bob", "alice", "bob"][:B], "is_in_mix_subj_folder": [False] * B,
            "image_path": [f"/data/{i}.jpg" for i in range(B)], "has_fg_mask": torch.ones(B, dtype=torch.bool),
            "has_wds_comp": torch.zeros(B, dtype=torch.bool),
            "aug_mask": torch.ones(B, 16, 16), "fg_mask": (seeded((B, 16, 16), 5) > 0).float(),
            "image_unnorm": torch.arange(B, dtype=torch.uint8).view(B, 1, 1, 1).repeat(1, 16, 16, 3),    # instance i: all pixels = i
            "zs_clip_features": seeded((B, 514, 8), 6), "zs_id_embs": seeded((B, 512), 7),
            "caption": list(ss), "caption_bg": bgify(ss),
            "subj_prompt_single": list(ss), "subj_prompt_comp": list(sc), "cls_prompt_single": list(cs),
            "cls_prompt_comp": list(cc),
            "subj_prompt_single_bg": bgify(ss), "subj_prompt_comp_bg": bgify(sc),
            "cls_prompt_single_bg": bgify(cs), "cls_prompt_comp_bg": bgify(cc)}


# ---- stage-2 cases (shared with tests/test_stage2_golden.py) --------------------------------------------------------------
STAGE2_LAYERS = {7: (16, 16), 12: (8, 8), 18: (16, 32), 21: (32, 64)}        # layer -> (attention side, outfeat side)


def stage2_case(seed=300, C=8, heads=2, d=4):
    """one instance x 4 blocks (subject single, subject comp, mix single, mix comp): captured outfeat / attnscore / q of
    four layers (two of them with the output upsampled behind the transformer), subject token positions, foreground mask."""
    outfeat = {li: seeded((4, C, so, so), seed + li) for li, (_, so) in STAGE2_LAYERS.items()}
    score = {li: seeded((4, heads, sa * sa, 77), seed + 50 + li, 1.5) for li, (sa, _) in STAGE2_LAYERS.items()}
    q = {li: seeded((4, heads, sa * sa, d), seed + 100 + li) for li, (sa, _) in STAGE2_LAYERS.items()}
    subj_1b = (torch.zeros(9, dtype=torch.long), torch.arange(4, 13))
    subj_2b = (torch.cat([torch.zeros(9, dtype=torch.long), torch.ones(9, dtype=torch.long)]), torch.arange(4, 13).repeat(2))
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
    fg = ((xx / 0.55) ** 2 + (yy / 0.7) ** 2 <= 1.0).float()[None, None].repeat(4, 1, 1, 1)
    return outfeat, score, q, subj_1b, subj_2b, fg


def grad_norms(tensors):
    return np.array([0.0 if t.grad is None else float(t.grad.double().norm()) for t in tensors])


SHARED_STEP_CASES = [("compos", False, 31), ("compos", False, 32), ("compos_fp", False, 33), ("compos_fp", False, 34),
                     ("recon", False, 11), ("recon", False, 12), ("distill", True, 21), ("distill", True, 22), ("distill", True, 23),
                     ("distill", True, 24), ("distill", True, 25), ("distill", True, 26)]


def flags_record(fl, x_start, captions, em_calls):
    keep = ("use_background_token", "gen_arc2face_rand_face", "add_noise_to_real_id_embs", "use_arc2face_as_target",
            "num_denoising_steps", "same_subject_in_batch", "use_wds_comp", "use_fp_trick", "comp_init_fg_from_training_image",
            "reuse_init_conds", "do_teacher_filter")
    out = {k: int(fl.get(k, 0)) for k in keep}
    out["bs"] = int(x_start.shape[0])
    out["captions"] = list(captions)
    out["n_delta_prompts"] = -1 if fl.get("delta_prompts") is None else len(fl["delta_prompts"][0])
    out["masks_none"] = [fl["img_mask"] is None, fl["fg_mask"] is None]
    out["have_fg"] = [bool(v) for v in fl["batch_have_fg_mask"].tolist()]
    out["zs_id_shape"] = list(fl["zs_id_embs"].shape)
    out["zs_clip_shape"] = list(fl["zs_clip_features"].shape)
    out["zs_id_norms"] = [round(float(v), 5) for v in fl["zs_id_embs"].norm(dim=-1)]
    out["a2f"] = None if fl.get("arc2face_prompt_emb") is None else list(fl["arc2face_prompt_emb"].shape)
    out["embman_names"] = [c for c in em_calls if c[0] == "set_curr_batch_subject_names"][-1][1:]
    return out


def json_dumps(x):
    import json
    return json.dumps(x)


def main():
    D = import_reference_ddpm()
    rec = {}
    # ---- A / B: schedule, q_sample, predict_start_from_noise
    m = nn.Module.__new__(D.LatentDiffusion)          # no __init__: only the attributes the methods below read
    nn.Module.__init__(m)
    m.parameterization, m.v_posterior, m.loss_type, m.num_timesteps_cond = "eps", 0.0, "l2", 1
    D.LatentDiffusion.register_schedule(m, beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.012)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        rec["sched/" + k] = getattr(m, k).numpy()
    x0, nz = seeded((4, 4, 8, 8), 1), seeded((4, 4, 8, 8), 2)
    t = torch.tensor([0, 17, 500, 999])
    xt = D.DDPM.q_sample(m, x0, t, nz)
    rec["q_sample"] = xt.numpy()
    rec["predict_x0"] = D.DDPM.predict_start_from_noise(m, xt, t, nz * 0.9 + 0.05).numpy()
    # ---- C: calc_recon_loss
    for i, (fw, bw, use_masks) in enumerate([(1, 1, True), (1, 0.1, True), (1.0, 0.0, True), (1, 0.05, False)]):
        out, tgt, img, fg = recon_case(10 + 10 * i)
        out.requires_grad_(True)
        loss, px = D.LatentDiffusion.calc_recon_loss(m, out, tgt, img if use_masks else None, fg if use_masks else None, fw, bw)
        loss.backward()
        rec[f"recon/{i}/loss"] = np.float64(loss.item())
        rec[f"recon/{i}/grad"] = out.grad.numpy()
        rec[f"recon/{i}/cfg"] = np.array([fw, bw, float(use_masks)])
    # ---- D: the recon iteration's attention losses
    sc, subj, bg, fg = attn_case()
    for tag, kw in (("full", dict(bg=bg, fg_mask=fg, inst=None, sqrt=False)),
                    ("nobg", dict(bg=None, fg_mask=fg, inst=None, sqrt=False)),
                    ("nomask", dict(bg=bg, fg_mask=None, inst=None, sqrt=False)),
                    ("inst", dict(bg=bg, fg_mask=fg, inst=torch.tensor([1.0, 0.0]), sqrt=True))):
        leaf = {li: v.clone().requires_grad_(True) for li, v in sc.items()}
        losses = D.LatentDiffusion.calc_fg_bg_complementary_loss(
            m, leaf, subj, kw["bg"], 2, fg_grad_scale=0.1, fg_mask=kw["fg_mask"], instance_mask=kw["inst"], do_sqrt_norm=kw["sqrt"])
        tot = sum(l for l in losses if torch.is_tensor(l))
        if torch.is_tensor(tot) and tot.requires_grad:
            tot.backward()
        rec[f"complem/{tag}/losses"] = np.array([float(l) for l in losses])
        rec[f"complem/{tag}/gnorm"] = np.array([0.0 if leaf[li].grad is None else float(leaf[li].grad.double().norm()) for li in ATTN_N])
    leaf = {li: v.clone().requires_grad_(True) for li, v in sc.items()}
    lfg, lbg = D.LatentDiffusion.calc_fg_bg_xlayer_consist_loss(m, leaf, subj, bg, 2)
    (lfg + lbg).backward()
    rec["xlayer/losses"] = np.array([float(lfg), float(lbg)])
    rec["xlayer/gnorm"] = np.array([0.0 if leaf[li].grad is None else float(leaf[li].grad.double().norm()) for li in ATTN_N])
    leaf = {li: v.clone().requires_grad_(True) for li, v in sc.items()}
    lfg1, lbg1 = D.LatentDiffusion.calc_fg_bg_xlayer_consist_loss(m, leaf, subj, None, 1)
    rec["xlayer_nobg/losses"] = np.array([float(lfg1), float(lbg1)])
    # ---- E: forward()'s conditioning assembly, with the test stand-ins as text encoder / embedding manager
    sys.path.insert(0, ROOT)
    from tests.stubs import StubEmbeddingManager, StubTextEncoder
    for mode in ("recon_delta", "recon_plain", "mix", "ada_delta"):
        enc = StubTextEncoder(dim=16)
        em = StubEmbeddingManager(text_embedder=enc, dim=16)
        fake = types.SimpleNamespace()
        fake.num_timesteps, fake.device = 1000, torch.device("cpu")
        fake.model = types.SimpleNamespace(conditioning_key="crossattn")
        fake.cond_stage_trainable, fake.N_CA_LAYERS, fake.prompt_mix_scheme = True, 16, "mix_hijk"
        fake.apply_arc2face_inverse_embs, fake.shorten_cond_schedule, fake.cached_inits = False, False, {}
        prompts = prompts_case(3)
        fake.iter_flags = {"do_static_prompt_delta_reg": mode != "recon_plain", "do_mix_prompt_distillation": mode == "mix",
                           "do_ada_prompt_delta_reg": mode in ("mix", "ada_delta"), "do_normal_recon": mode.startswith("recon"),
                           "do_arc2face_distill": False, "reuse_init_conds": False, "delta_prompts": prompts,
                           "zs_clip_features": None, "zs_id_embs": None}

        def glc(cond_in, zs_clip_features=None, zs_id_embs=None, randomize_clip_weights=False,
                apply_arc2face_inverse_embs=False, _enc=enc, _em=em):
            emb = _enc.encode(cond_in, embedding_manager=_em)
            return emb, cond_in, {"placeholder2indices": dict(_em.placeholder2indices), "prompt_emb_mask": _em.prompt_emb_mask}
        fake.get_learned_conditioning = glc
        fake.p_losses = lambda x_start, cond, t: cond
        torch.manual_seed(0)
        c_emb, c_in, extra = D.LatentDiffusion.forward(fake, torch.zeros(3, 4, 8, 8), list(prompts[0]))
        rec[f"fwd/{mode}/c_emb"] = c_emb.detach().numpy()
        rec[f"fwd/{mode}/c_in"] = np.array(list(c_in))
        rec[f"fwd/{mode}/keys"] = np.array(sorted(extra.keys()))
        rec[f"fwd/{mode}/iter_type"] = np.array(extra["iter_type"])
        for name in ("placeholder2indices", "placeholder2indices_1b", "placeholder2indices_2b"):
            if name in extra:
                for ph, (ib, it) in extra[name].items():
                    rec[f"fwd/{mode}/{name}/{ph}"] = np.stack([ib.numpy(), it.numpy()])
        for name in ("c_static_emb_4b", "c_static_emb_1b"):
            if name in extra:
                rec[f"fwd/{mode}/{name}"] = extra[name].detach().numpy()
    # ---- F: the teacher rollout's timestep schedule and x0 chain (closed-form eps model in place of the diffusers UNet)
    for nd in (1, 3, 5):
        wrap = types.SimpleNamespace()
        wrap.unet = lambda sample, timestep, encoder_hidden_states, return_dict=False: (
            (0.3 * sample + 0.01 * timestep.view(-1, 1, 1, 1).float() / 1000 + encoder_hidden_states.mean()).float(),)
        torch.manual_seed(40 + nd)
        x0, nz = seeded((2, 4, 8, 8), 50 + nd), seeded((2, 4, 8, 8), 60 + nd)
        t = torch.tensor([900, 431])
        ctx = seeded((2, 21, 16), 70 + nd, 0.1)
        preds, x0s, noises, ts = D.Arc2FaceWrapper.forward.__wrapped__(wrap, m, x0, nz, t, ctx, num_denoising_steps=nd) \
            if hasattr(D.Arc2FaceWrapper.forward, "__wrapped__") else D.Arc2FaceWrapper.forward(wrap, m, x0, nz, t, ctx, num_denoising_steps=nd)
        rec[f"rollout/{nd}/ts"] = torch.stack(ts).numpy()
        rec[f"rollout/{nd}/pred_last"] = preds[-1].numpy()
        rec[f"rollout/{nd}/x0_last"] = x0s[-1].numpy()
        rec[f"rollout/{nd}/noise_last"] = noises[-1].numpy()
    # ---- H: stage 2 (compositional distillation): ldm/util.py functions and the two ddpm.py loss methods
    import random as _random
    import ldm.util as U
    assert U.__file__.startswith(REF)
    emb = seeded((2 * 2 * 16, 77, 12), 200, 0.3).requires_grad_(True)          # BS = 2: (subject half | class half)
    for tag, kw in (("a", dict(training_percent=0.25, t_frac=torch.tensor([0.9, 0.85]), K=[1.0, 0.8], V=[1.0, 0.6])),
                    ("b", dict(training_percent=0.8, t_frac=torch.tensor([0.55, 0.95]), K=[1.0, 1.0], V=[1.0, 0.7]))):
        emb.grad = None
        out, _, v_sc, _, k_sc = U.mix_static_vk_embeddings(emb, torch.arange(4, 13), kw["training_percent"], t_frac=kw["t_frac"],
                                                           use_layerwise_embedding=True, N_CA_LAYERS=16,
                                                           K_CLS_SCALE_LAYERWISE_RANGE=kw["K"], V_CLS_SCALE_LAYERWISE_RANGE=kw["V"])
        (out * seeded(tuple(out.shape), 201)).sum().backward()
        rec[f"s2/mixvk/{tag}/out"] = out.detach().numpy()
        rec[f"s2/mixvk/{tag}/grad"] = emb.grad.numpy().copy()
        rec[f"s2/mixvk/{tag}/scales"] = np.stack([v_sc.numpy(), k_sc.numpy()])
    rec["s2/cfg_scales"] = U.gen_cfg_scales_for_stu_tea(6, 5, 2, "cpu").numpy()
    rec["s2/dyn_scale"] = np.array([U.calc_dyn_loss_scale(torch.tensor(v), 0.2, 2, min_scale_base_ratio=1, max_scale_base_ratio=3)
                                    for v in (0.05, 0.3, 0.9)])
    # elastic matching on pooled maps
    qe, fe = seeded((4, 10, 49), 210).requires_grad_(True), seeded((4, 14, 49), 211).requires_grad_(True)
    me = (torch.rand(1, 1, 49, generator=torch.Generator().manual_seed(212)) > 0.55).float()
    lm, lf, lb, scb, mcb = U.calc_elastic_matching_loss(qe, fe, me, fg_bg_cutoff_prob=0.25, single_q_grad_scale=0.1,
                                                        single_feat_grad_scale=0.01, mix_feat_grad_scale=0.05)
    (lm + lf + lb).backward()
    rec["s2/elastic/losses"] = np.array([float(lm), float(lf), float(lb)])
    rec["s2/elastic/below"] = np.stack([scb.detach().numpy(), mcb.detach().numpy()])
    rec["s2/elastic/grad_q"], rec["s2/elastic/grad_f"] = qe.grad.numpy(), fe.grad.numpy()
    # delta alignment, spatial weights
    fb, fx, rb, rx = (seeded((1, 2, 64), 220 + i).requires_grad_(True) for i in range(4))
    dl = U.calc_delta_alignment_loss(fb, fx, rb, rx, ref_grad_scale=0.05, feat_base_grad_scale=1, use_cosine_loss=True,
                                     cosine_exponent=3, delta_types=["feat_to_ref"])["feat_to_ref"]
    dl.backward()
    rec["s2/delta/loss"] = np.float64(float(dl))
    rec["s2/delta/gnorm"] = grad_norms([fb, fx, rb, rx])
    sw, sa = U.convert_attn_to_spatial_weight(seeded((2, 2, 256), 230, 1.0) + 0.5, 1, torch.Size([32, 32]), reversed=True)
    rec["s2/spatial_weight"], rec["s2/spatial_attn"] = sw.numpy(), sa.numpy()
    # the compositional iteration's initial latent
    for tag, pct in (("big", 0.6), ("small", 0.25)):
        np.random.seed(77)
        torch.manual_seed(77)
        x0 = seeded((2, 4, 64, 64), 240)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
        fgm = ((xx / pct) ** 2 + (yy / pct) ** 2 <= 1.0).float()[None, None].repeat(2, 1, 1, 1)
        xi, m1, m2 = U.init_x_with_fg_from_training_image(x0, fgm, fgm.clone(), 0.4, base_scale_range=(0.7, 1.0),
                                                          fg_noise_anneal_mean_range=(0.1, 0.4))
        rec[f"s2/initx/{tag}/x"], rec[f"s2/initx/{tag}/fg"] = xi.numpy(), m1.numpy()
        rec[f"s2/initx/{tag}/after"] = np.array([float(np.random.rand()), float(torch.rand(1))])
    # the two loss methods of ddpm.py
    outfeat, score, qq, subj_1b, subj_2b, fg4 = stage2_case()
    leaves = {k: {li: v.clone().requires_grad_(True) for li, v in dct.items()} for k, dct in
              (("outfeat", outfeat), ("score", score), ("q", qq))}
    l_feat, l_attn_delta, l_attn_norm = D.LatentDiffusion.calc_prompt_mix_loss(m, leaves["outfeat"], None, leaves["score"], subj_2b, 1)
    (l_feat + l_attn_delta + l_attn_norm).backward()
    rec["s2/prompt_mix/losses"] = np.array([float(l_feat), float(l_attn_delta), float(l_attn_norm)])
    rec["s2/prompt_mix/gnorm_outfeat"] = grad_norms([leaves["outfeat"][li] for li in STAGE2_LAYERS])
    rec["s2/prompt_mix/gnorm_score"] = grad_norms([leaves["score"][li] for li in STAGE2_LAYERS])
    leaves = {k: {li: v.clone().requires_grad_(True) for li, v in dct.items()} for k, dct in
              (("outfeat", outfeat), ("score", score), ("q", qq))}
    have = torch.ones(4)
    ls = D.LatentDiffusion.calc_comp_fg_bg_preserve_loss(m, leaves["outfeat"], None, leaves["q"], None, leaves["score"], fg4, have,
                                                         subj_1b, 1)
    sum(l for l in ls if torch.is_tensor(l)).backward()
    rec["s2/preserve/losses"] = np.array([float(l) for l in ls])
    for k in ("outfeat", "score", "q"):
        rec[f"s2/preserve/gnorm_{k}"] = grad_norms([leaves[k][li] for li in STAGE2_LAYERS])
    ls0 = D.LatentDiffusion.calc_comp_fg_bg_preserve_loss(m, outfeat, None, qq, None, score, fg4, torch.zeros(4), subj_1b, 1)
    rec["s2/preserve/no_mask"] = np.array([float(l) for l in ls0])
    # teacher selection (calc_clip_losses, ddpm.py:3636-3679) with a scripted CLIP evaluator
    sel = []
    for losses in ([0.30, 0.27, 0.26, 0.20], [0.25, 0.26, 0.27, 0.29], [0.31, 0.30, 0.279, 0.2795]):
        fake = types.SimpleNamespace(iter_flags={"do_teacher_filter": True, "reuse_init_conds": False}, num_candidate_teachers=2,
                                     num_total_teacher_filter_iters=0, num_teachable_iters=0, num_total_reuse_filter_iters=0.001,   # ddpm.py:300-303
                                     num_reuse_teachable_iters=0)
        fake.decode_first_stage = lambda z: z
        fake.clip_evaluator = types.SimpleNamespace(txt_to_img_similarity=lambda prompts, images, reduction, _l=losses: 0.5 - torch.tensor(_l))
        try:
            _imgs, teach, best, colors = D.LatentDiffusion.calc_clip_losses(fake, torch.zeros(4, 4, 8, 8), {"cls_comp_prompts": ["p"]}, {}, "train")
        except ZeroDivisionError:            # (its reuse-iteration statistics divide by a counter that is still 0)
            raise
        sel.append([[bool(v) for v in teach.tolist()], int(best), [int(c) for c in colors.tolist()]])
    rec["s2/select"] = np.array(json_dumps(sel))
    # ---- G: the front of shared_step (ddpm.py:1436-1938): iteration flags and the order random / np.random are consumed in
    import json
    import random
    from tests.stubs import StubEmbeddingManager as _EM
    cases = []
    for kind, distill, seed in SHARED_STEP_CASES:
        class Fake:                                  # callable: shared_step ends in ``self(x_start, captions)``
            def __call__(self, x_start, captions):
                return x_start, captions
        fake = Fake()
        fake.first_stage_key, fake.training_percent = "image", 0.3
        fake.do_static_prompt_delta_reg, fake.use_fp_trick, fake.use_background_token = True, True, True
        fake.do_clip_teacher_filtering, fake.do_zero_shot, fake.cached_inits = True, True, {}
        fake.p_gen_arc2face_rand_face, fake.p_add_noise_to_real_id_embs, fake.max_num_denoising_steps = 0.4, 0.6, 5
        fake.apply_arc2face_inverse_embs = False
        fake.embedding_manager = _EM(dim=8)
        batch = shared_step_batch(4)
        if kind == "compos_fp":              # broad_class == 1 datasets also carry the face-portrait prompt variants
            for k in list(batch.keys()):
                if k.startswith(("subj_prompt", "cls_prompt")):
                    base, bg = (k[:-3], "_bg") if k.endswith("_bg") else (k, "")
                    batch[base + "_fp" + bg] = ["a face portrait of " + q for q in batch[k]]
        fake.get_input = lambda b, k: (seeded((len(b["subject_name"]), 4, 2, 2), 9), None)
        def fake_encoder(images, fg, image_paths=None, is_face=True, calc_avg=False, _b=batch):
            # features are a function of the IMAGES handed over (instance i's pixels are all i): a repeated image gives
            # repeated features; calc_avg as ddpm.py:2442-2465 (mean; the id embedding re-normalised)
            idx = images[:, 0, 0, 0].long()
            f, e = _b["zs_clip_features"][idx], _b["zs_id_embs"][idx]
            if calc_avg:
                f = f.mean(dim=0, keepdim=True)
                e = torch.nn.functional.normalize(e.mean(dim=0, keepdim=True), p=2, dim=-1)
            return f, e, 0
        fake.encode_zero_shot_image_features = fake_encoder
        fake.arc2face = types.SimpleNamespace(gen_arc2face_prompt_embs=lambda n, pre_face_embs=None: (
            n, pre_face_embs if pre_face_embs is not None else seeded((n, 512), 8), seeded((n, 21, 8), 10)))
        fake.iter_flags = {}
        D.DDPM.init_iteration_flags(fake)
        fake.iter_flags["do_arc2face_distill"] = distill
        if kind.startswith("compos"):        # what training_step sets for a prompt-mix iteration (ddpm.py:556-565)
            fake.iter_flags.update(do_mix_prompt_distillation=True, do_ada_prompt_delta_reg=True, is_compos_iter=True,
                                   calc_clip_loss=True, do_normal_recon=False)
        if distill:
            fake.iter_flags["do_static_prompt_delta_reg"] = False          # training_step switches it off (ddpm.py:572)
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        x_start, captions = D.LatentDiffusion.shared_step(fake, batch)
        r = flags_record(fake.iter_flags, x_start, captions, fake.embedding_manager.calls)
        r["after"] = [random.random(), float(np.random.rand())]            # where the two host RNG streams stand afterwards
        cases.append(r)
    rec["shared_step/cases"] = np.array(json.dumps(cases))
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, len(rec), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

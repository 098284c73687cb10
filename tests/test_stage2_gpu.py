"""One compositional-distillation micro-batch (BASELINE config 4: bs = 3, 154-token split K/V context) on the
MI355X against the same iteration with every device computation replaced by the CPU oracle: the no-grad teacher-filter pass
with classifier-free guidance, VAE decode, teacher selection, the with-grad pass on the selected candidate under the four
mixed contexts, the losses on the captured outfeat / attnscore / q of the distillation layers, and the gradient into the
embedding manager's subject vectors (through the K/V prompt mixing, the UNet's attention side outputs and features).
The host logic (iteration driver, stage-2 losses) is the product's own on both sides -- it is pinned separately by the
reference-generated goldens (tests/test_stage2_golden.py, tests/test_ddpm_golden.py); what is compared here is HIP vs oracle."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from adaprompt_amd import synth          # noqa: E402
from conftest import rel_err          # noqa: E402


# measured (round 2/3): x_start_cache 0.19, every loss part <= 2.2e-2, the gradient into the subject vectors 2.0e-2.  The cached
# x0 is the guided eps divided by sqrt(alpha_bar_t) ~ 0.07 at t ~ 900: an eps error of 9e-3 (the UNet gate) times the guidance
# scale becomes ~0.2 of it by construction, hence its own, loose gate; the other two sit at <= 2x what is measured
X0_TOL, PART_TOL, GRAD_TOL = 0.25, 4e-2, 4e-2
VEC_SCALE = 1.0


def _params(ucfg, vdd, dim, num_teachers=2):
    return {"first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL",
                                   "params": {"ddconfig": vdd, "embed_dim": 4, "with_decoder": True}},
            "cond_stage_config": {"target": "tests.stubs.StubTextEncoder", "params": {"dim": dim}},
            "personalization_config": {"target": "tests.stubs.StubEmbeddingManager", "params": {"dim": dim, "num_vectors_per_subj_token": 9}},
            "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": ucfg},
            "scale_factor": 0.18215, "linear_start": 0.00085, "linear_end": 0.012, "conditioning_key": "crossattn",
            "cond_stage_trainable": True, "use_layerwise_embedding": True, "do_zero_shot": True, "mix_prompt_distill_weight": 1e-4,
            "comp_fg_bg_preserve_loss_weight": 1e-3, "prompt_emb_delta_reg_weight": 2e-4, "normalize_ca_q_and_outfeat": True,
            "num_candidate_teachers": num_teachers, "composition_regs_iter_gap": 3}


def _batch(B, dev):
    import make_golden_ddpm as G
    b = G.shared_step_batch(4)
    b = {k: (v[:B] if (torch.is_tensor(v) or isinstance(v, list)) else v) for k, v in b.items()}
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    fg = ((xx / 0.6) ** 2 + (yy / 0.7) ** 2 <= 1.0).float()[None].repeat(B, 1, 1)
    b.update(fg_mask=fg, aug_mask=torch.ones(B, 512, 512), zs_clip_features=G.seeded((B, 514, 8), 6), zs_id_embs=G.seeded((B, 512), 7))
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}


def _score_fn(prompts, images):
    """scripted CLIP similarity: candidate 1's mixed image beats its subject image by the larger margin; a small image-dependent
    term makes the score a function of the decoded pixels (so a wrong decode shows) without being able to flip the order."""
    n = images.shape[0]
    base = torch.tensor([0.30, 0.31, 0.25, 0.22][:n] if n == 4 else [0.30, 0.22], device=images.device)
    stat = images.float()[:, :, ::8, ::8].mean(dim=(1, 2, 3))
    return 0.5 - (base + 0.02 * torch.tanh(stat))


@pytest.mark.gpu
@pytest.mark.parametrize("size,B", [("narrow", 3), ("sd15", 1)])
def test_compositional_micro_batch_vs_oracle(size, B):
    """``narrow``: BASELINE config 4's bs = 3 at model_channels 64, with the CLIP-scored teacher filter over two candidates.
    ``sd15``: the micro-batch at the FULL SD-1.5 sizes (859.5 M UNet, 768-wide context, full VAE decoder) with one instance and
    ``do_clip_teacher_filtering`` off -- the guided with-grad pass under the four mixed contexts, the decode of the two comp
    images for the score, the stage-2 losses and the gradient into the subject vectors -- so that the fp32 oracle's autograd fits
    the suite's time budget (~1.5 min of CPU; the filter's extra no-grad passes are the narrow case's business)."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import ldm_oracle as O
    dev = torch.device("cuda:0")
    if size == "narrow":
        ucfg = dict(synth.SD15_UNET, model_channels=64, context_dim=128)
        vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=512)
    else:
        ucfg, vdd = dict(synth.SD15_UNET), dict(synth.SD15_VAE_DD)
    cdim = ucfg["context_dim"]
    NT = 2 if size == "narrow" else 1          # candidates of the teacher filter (one per instance at most)
    usd = synth.synthetic_unet_state_dict(ucfg)
    vsd = synth.synthetic_vae_state_dict(vdd, decoder=True)
    x0 = synth.synthetic_input("s2.x0", (B, 4, 64, 64))
    noise = synth.synthetic_input("s2.noise", (B, 4, 64, 64))
    fresh = synth.synthetic_input("s2.fresh", (B, 4, 64, 64))
    t = torch.tensor([931, 872, 990][:B])

    def run(device, oracle):
        torch.manual_seed(0)
        ld = LatentDiffusion(**_params(ucfg, vdd, cdim, NT))
        missing, unexpected = ld.load_state_dict({**usd, **vsd}, strict=False)
        assert not unexpected
        ld = ld.to(device)
        if size == "sd15":
            ld.do_clip_teacher_filtering = False
        ld.embedding_manager.vectors["z"].data.mul_(VEC_SCALE)
        ld.embedding_manager.vectors["y"].data.mul_(VEC_SCALE)
        if oracle:
            sched = O.make_schedule()
            ld.q_sample = lambda x_start, t, noise=None: O.q_sample(sched, x_start, t, noise)
            ld.apply_model = lambda x_noisy, tt, cond: O.unet_forward(usd, ucfg, x_noisy, tt, cond[0], cond[2])
            ld.decode_first_stage = lambda z, **kw: O.decode_first_stage(vsd, vdd, z)
        ld.clip_score_fn = _score_fn
        ld.training_percent = 0.3
        ld.init_iteration_flags()
        ld.iter_flags.update(do_mix_prompt_distillation=True, do_ada_prompt_delta_reg=True, is_compos_iter=True, calc_clip_loss=True,
                             do_normal_recon=False)
        ld._compos_test_hooks = {"py_random": random.Random(5), "randn_like": lambda like: fresh[:like.shape[0]].to(like.device)}
        random.seed(3)            # the front's draws: fresh iteration, no fp prompts in this batch, background token by chance
        np.random.seed(11)        # (init_x_with_fg_from_training_image draws from numpy's global generator)
        batch = _batch(B, device)
        loss, grad, out, aux = ld.shared_step(batch, t=t.to(device), noise=noise.to(device), x_start=x0.to(device))
        ld.manual_backward(out, grad, aux)
        if device.type == "cuda":
            torch.cuda.synchronize()
        g = {k: v.grad.detach().cpu() for k, v in ld.embedding_manager.vectors.items() if v.grad is not None}
        parts = {k: (float(v) if torch.is_tensor(v) else v) for k, v in aux["reg_parts"].items()}
        return float(loss), parts, g, dict(ld.iter_flags), sorted(ld.cached_inits.keys()), ld.cached_inits

    lh, ph, gh, fh, ch, cache_h = run(dev, oracle=False)
    if size == "narrow":
        # the same micro-batch again on the device: every loss part and the gradient are bit-equal (the Stage-2 losses sum in a
        # fixed order since round 4: csrc/stage2loss.hip; before, the vendor GEMMs under them moved comp_single_map_align by up
        # to 12 % of itself from run to run)
        lh2, ph2, gh2, _, _, _ = run(dev, oracle=False)
        assert lh2 == lh and ph2 == ph, sorted(k for k in ph if ph[k] != ph2[k])
        assert all(torch.equal(gh[k], gh2[k]) for k in gh)
    lo, po, go, fo, co, cache_o = run(torch.device("cpu"), oracle=True)
    # same decisions
    filtered = size == "narrow"
    assert fh["is_teachable"] and fo["is_teachable"] and ph["best_cand_idx"] == po["best_cand_idx"] == (NT - 1 if filtered else 0)
    assert fh["do_teacher_filter"] == fo["do_teacher_filter"] == filtered and fh["use_background_token"] == fo["use_background_token"]
    report = {"grad_z": rel_err(gh["z"], go["z"]), "grad_y": rel_err(gh["y"], go["y"]), "loss": abs(lh - lo) / abs(lo)}
    if filtered:                  # the selected candidate's x0 prediction is cached for the reuse iteration
        assert ch == co == ["alice"]
        assert torch.equal(cache_h["alice"]["t"].cpu(), cache_o["alice"]["t"])
        report["x_start_cache"] = rel_err(cache_h["alice"]["x_start"].cpu(), cache_o["alice"]["x_start"])
    else:
        assert ch == co == []
    for k in po:
        if k != "best_cand_idx":
            report[k] = abs(ph[k] - po[k]) / (abs(po[k]) + 1e-12)
    print(f"stage-2 micro-batch [{size}], HIP vs oracle (relative):", {k: round(v, 5) for k, v in report.items()})
    # the x0 prediction cached for the reuse iteration: guided noise prediction (scale ~3) of the selected candidate, divided by
    # sqrt(alphas_cumprod[t]) ~ 0.07 at t ~ 900 -- the bf16 path's ~1 % noise-prediction error is amplified ~10x
    assert report.get("x_start_cache", 0.0) < X0_TOL
    for k in ("loss_clip_subj_comp", "loss_clip_cls_comp"):
        assert abs(ph[k] - po[k]) < 2e-3, (k, ph[k], po[k])
    for k in po:
        if k not in ("best_cand_idx", "loss_clip_subj_comp", "loss_clip_cls_comp"):
            # feat_delta_align is a difference of differences of nearly equal features: bf16 operands leave a floor of
            # ~(2^-9 |feat|)^2 ~ 2e-5 under it, whatever its value
            # comp_single_map_align (~4e-4) sums what is left after a HARD elastic matching of query positions: in the narrow
            # model a handful of matches flip with the bf16 noise of the queries (0.14 of it = 5.6e-5; at the full size the
            # same part agrees to 1.8e-4 of itself) -- a count of flipped matches, heavy-tailed, hence a floor at 2x.  The run
            # to run movement the floor also had to absorb before round 4 is gone (asserted bit-equal above)
            floor = {"feat_delta_align": 4e-5, "comp_single_map_align": 1.2e-4}.get(k, 2e-6)
            assert abs(ph[k] - po[k]) < PART_TOL * abs(po[k]) + floor, (k, ph[k], po[k])
    assert report["loss"] < PART_TOL
    assert float(go["z"].norm()) > 0
    assert report["grad_z"] < GRAD_TOL, report

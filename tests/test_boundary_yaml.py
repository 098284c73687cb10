"""The drop-in boundary at the yaml level: the dotted ``target:`` strings of the reference's
``configs/stable-diffusion/v1-finetune-ada.yaml`` resolve with this repo AHEAD of a reference checkout on ``sys.path`` -- the
mirrored ones to this package, the boundary callees (embedding manager, text encoder, LR scheduler) to the reference's own
modules -- and the ``params`` of yaml:5 instantiate this package's ``LatentDiffusion`` unchanged."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
YAML = os.path.join(REF, "configs", "stable-diffusion", "v1-finetune-ada.yaml")
needs_ref = pytest.mark.skipif(not os.path.isfile(YAML), reason="no reference checkout on this box")

_RESOLVE = r'''
import sys, types, yaml
sys.dont_write_bytecode = True
def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
sys.path.insert(0, %(ref)r)
sys.path.insert(0, %(root)r)
import ldm                                   # this repo's alias package (first on the path)
assert ldm.__file__.startswith(%(root)r), ldm.__file__
stub("cv2")                                  # third-party packages the image lacks; import-time only
import adaface.subj_basis_generator          # (transformers probes torchvision.__spec__: import before the stub below)
tv = stub("torchvision"); tv.utils = stub("torchvision.utils", make_grid=None, draw_bounding_boxes=None)
stub("clip"); stub("kornia")
from ldm.util import get_obj_from_str
cfg = yaml.safe_load(open(%(yaml)r))
found = {}
def walk(node):
    if isinstance(node, dict):
        t = node.get("target")
        if isinstance(t, str) and t.startswith("ldm."):
            found[t] = get_obj_from_str(t)
        for v in node.values():
            walk(v)
    elif isinstance(node, list):
        for v in node:
            walk(v)
walk(cfg["model"])
for t, obj in sorted(found.items()):
    print(t, "->", obj.__module__, sys.modules[obj.__module__].__file__)
'''


@needs_ref
def test_every_model_target_of_the_reference_yaml_resolves():
    out = subprocess.run([sys.executable, "-c", _RESOLVE % {"ref": REF, "root": ROOT, "yaml": YAML}], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = dict(l.split(" -> ") for l in out.stdout.strip().splitlines())
    want_here = ["ldm.models.diffusion.ddpm.LatentDiffusion", "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                 "ldm.models.autoencoder.AutoencoderKL", "ldm.lr_scheduler.LambdaWarmUpCosineScheduler"]
    want_ref = ["ldm.modules.embedding_manager.EmbeddingManager", "ldm.modules.encoders.modules.FrozenCLIPEmbedder"]
    for t in want_here:                       # yaml:5, 108, 125, 66 -> the MI355X implementation
        assert lines[t].startswith("adaprompt_amd.") and ROOT in lines[t], (t, lines[t])
    for t in want_ref:                        # yaml:87, 150 -> the reference's own modules, untouched
        assert REF in lines[t], (t, lines[t])


def _stubbed(params):
    """yaml:5's params with the two boundary configs pointed at the test stand-ins and the UNet / VAE narrowed (the
    constructor surface is what is under test, not 860 M parameters)."""
    p = dict(params)
    p["cond_stage_config"] = {"target": "tests.stubs.StubTextEncoder", "params": {"dim": 64, "last_layers_skip_weights": [0.5, 0.5]}}
    pers = dict(p["personalization_config"]["params"])
    p["personalization_config"] = {"target": "tests.stubs.StubEmbeddingManager",
                                   "params": {"subject_strings": pers["subject_strings"],
                                              "background_strings": pers["background_strings"],
                                              "num_vectors_per_subj_token": pers["num_vectors_per_subj_token"], "dim": 64}}
    u = dict(p["unet_config"]); u["params"] = dict(u["params"], model_channels=32, context_dim=64); p["unet_config"] = u
    f = dict(p["first_stage_config"]); fp = dict(f["params"]); fp["ddconfig"] = dict(fp["ddconfig"], ch=32); f["params"] = fp
    p["first_stage_config"] = f
    return p


def _check_model(ld, params):
    assert ld.scale_factor == 0.18215 and ld.manual_accumulate_grad_batches == 2 and ld.grad_clip == 0.5
    assert ld.optimizer_type == "Prodigy" and ld.prodigy_config["d_coef"] == 2 and ld.composition_regs_iter_gap == 3
    assert ld.do_zero_shot and ld.arc2face_distill_iter_prob == params["arc2face_distill_iter_prob"]
    assert ld.mix_prompt_distill_weight == 1e-4 and ld.comp_fg_bg_preserve_loss_weight == 1e-3
    assert not any(q.requires_grad for q in ld.model.parameters())                  # unfreeze_model: False
    assert not any(q.requires_grad for q in ld.first_stage_model.parameters())
    assert not any(q.requires_grad for q in ld.cond_stage_model.parameters())
    assert ("make_frozen_copy",) in ld.embedding_manager.calls
    assert tuple(ld.empty_context.shape) == (1, 77, 64)
    assert float(ld.sqrt_alphas_cumprod[0]) == pytest.approx((1 - 0.00085) ** 0.5, rel=1e-6)
    # Lightning-facing surface: training_step(batch, batch_idx), configure_optimizers(), on_save_checkpoint(checkpoint)
    import inspect
    assert list(inspect.signature(ld.training_step).parameters)[:2] == ["batch", "batch_idx"]
    assert list(inspect.signature(ld.on_save_checkpoint).parameters)[0] == "checkpoint"


@needs_ref
def test_yaml_params_instantiate_latent_diffusion():
    from adaprompt_amd.ldm.util import instantiate_from_config, load_config
    cfg = load_config(YAML)["model"]
    assert cfg["params"]["prompt_emb_delta_reg_weight"] == 2e-4 and cfg["base_learning_rate"] == 8e-04
    assert cfg["target"] == "ldm.models.diffusion.ddpm.LatentDiffusion"
    params = _stubbed(cfg["params"])
    ld = instantiate_from_config({"target": cfg["target"], "params": params})
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    assert type(ld) is LatentDiffusion
    _check_model(ld, params)


def test_reference_keyword_surface_instantiates_without_a_checkout():
    """the same on a box without the reference (the GPU box): the keyword names of yaml:5-84, written out here."""
    from adaprompt_amd import synth
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    params = {
        "beta_schedule": "linear", "linear_start": 0.00085, "linear_end": 0.0120, "num_timesteps_cond": 1, "log_every_t": 200,
        "timesteps": 1000, "first_stage_key": "image", "cond_stage_key": "caption", "image_size": 64, "channels": 4,
        "cond_stage_trainable": True, "conditioning_key": "crossattn", "monitor": "val/loss_ema", "scale_factor": 0.18215,
        "use_ema": False, "unfreeze_model": False, "model_lr": 0.0, "use_layerwise_embedding": True, "use_fp_trick": True,
        "do_clip_teacher_filtering": True, "num_candidate_teachers": 2, "composition_regs_iter_gap": 3, "do_zero_shot": True,
        "arc2face_distill_iter_prob": 0.1, "static_embedding_reg_weight": 0, "prompt_emb_delta_reg_weight": 2e-4,
        "mix_prompt_distill_weight": 1e-4, "normalize_ca_q_and_outfeat": True, "comp_fg_bg_preserve_loss_weight": 1e-3,
        "fg_bg_complementary_loss_weight": 2e-4, "fg_bg_xlayer_consist_loss_weight": 5e-5,
        "compel_cfg_weight_level_range": [2, 2], "apply_compel_cfg_prob": 0.5, "fg_wds_complementary_loss_weight": 0,
        "wds_bg_recon_discount": 0.05, "optimizer_type": "Prodigy", "grad_clip": 0.5, "manual_accumulate_grad_batches": 2,
        "adam_config": {"betas": [0.9, 0.993]},
        "prodigy_config": {"betas": [0.985, 0.993], "zs_betas": [0.9, 0.999], "d_coef": 2, "warm_up_steps": 500,
                           "scheduler_cycles": 1, "scheduler_type": "Linear"},
        "personalization_config": {"params": {"subject_strings": ["z"], "background_strings": ["y"], "num_vectors_per_subj_token": 9}},
        "unet_config": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(synth.SD15_UNET)},
        "first_stage_config": {"target": "ldm.models.autoencoder.AutoencoderKL",
                               "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": dict(synth.SD15_VAE_DD),
                                          "lossconfig": {"target": "torch.nn.Identity"}}},
    }
    params = _stubbed(params)
    ld = LatentDiffusion(**params)
    _check_model(ld, params)
    # refused loudly, not ignored
    with pytest.raises(NotImplementedError):
        LatentDiffusion(**dict(params, use_ema=True))
    with pytest.raises(TypeError):
        LatentDiffusion(**dict(params, not_a_reference_keyword=1))

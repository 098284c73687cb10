"""Host-side logic of the drop-in boundary, checkable without a GPU: config targets, state-dict
namespace, flag plumbing, schedule buffers, checkpoint API, loud failure without the device."""
import os

import pytest
import torch

from adaprompt_amd import synth
from oracle import ldm_oracle as O

NARROW = dict(synth.SD15_UNET, model_channels=64, context_dim=128)


def test_reference_yaml_targets_resolve_to_this_package():
    from adaprompt_amd.ldm.util import instantiate_from_config, get_obj_from_str
    import ldm  # noqa: F401  (alias package of this repo)
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.models.autoencoder import AutoencoderKL
    import adaprompt_amd.ldm.modules.diffusionmodules.openaimodel as mine
    assert UNetModel is mine.UNetModel
    assert get_obj_from_str("ldm.models.autoencoder.AutoencoderKL") is AutoencoderKL
    m = instantiate_from_config({"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": NARROW})
    assert isinstance(m, mine.UNetModel)
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})


def test_state_dict_namespace_matches_reference_checkpoints():
    """686 UNet tensors / 859 520 964 parameters, named as in the reference."""
    from adaprompt_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    with torch.device("meta"):
        m = UNetModel(**synth.SD15_UNET)
    sd = m.state_dict()
    ref = dict(synth.unet_param_shapes(**synth.SD15_UNET))
    assert set(sd) == set(ref) and len(sd) == 686
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert sum(v.numel() for v in sd.values()) == 859_520_964
    from adaprompt_amd.ldm.models.autoencoder import AutoencoderKL
    with torch.device("meta"):
        v = AutoencoderKL(dict(synth.SD15_VAE_DD), None, 4)
    enc = {k: v_ for k, v_ in v.state_dict().items() if not k.startswith("post_quant")}
    refv = dict(synth.vae_encoder_param_shapes(**synth.SD15_VAE_DD))
    assert set(enc) == set(refv)
    assert sum(p.numel() for n, p in enc.items() if n.startswith("encoder.")) == 34_163_592


def test_cross_attn_flag_plumbing():
    """set_cross_attn_flags: layerwise arrays, subset of layers, and restore (openaimodel.py:722-824)."""
    from adaprompt_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel, DISTILL_LAYERS
    with torch.device("meta"):
        m = UNetModel(**NARROW)
    mods = dict(m._ca_modules())
    assert sorted(mods) == [1, 2, 4, 5, 7, 8, 12, 16, 17, 18, 19, 20, 21, 22, 23, 24]
    old, _ = m.set_cross_attn_flags(ca_flag_dict={"save_attn_vars": True}, ca_layer_indices=DISTILL_LAYERS)
    assert old == {"save_attn_vars": False}
    on = [li for li, s in mods.items() if s.transformer_blocks[0].attn2.save_attn_vars]
    assert on == DISTILL_LAYERS
    m.set_cross_attn_flags(ca_flag_dict=old, ca_layer_indices=DISTILL_LAYERS)
    assert not any(s.transformer_blocks[0].attn2.save_attn_vars for s in mods.values())
    sizes = list(range(16))
    m.set_cross_attn_flags(ca_flag_dict={"use_conv_attn_kernel_size:layerwise": sizes})
    assert mods[12].transformer_blocks[0].attn2.use_conv_attn_kernel_size == 6
    assert mods[24].transformer_blocks[0].attn2.use_conv_attn_kernel_size == 15


def test_schedule_buffers_match_oracle():
    from adaprompt_amd.ldm.models.diffusion.ddpm import DDPM
    with torch.device("meta"):
        pass
    d = DDPM({"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": NARROW},
             linear_start=0.00085, linear_end=0.012)
    s = O.make_schedule()
    for k, v in s.items():
        assert torch.equal(getattr(d, k), v), k
    assert d.num_timesteps == 1000


def test_checkpoint_api_roundtrip(tmp_path):
    """.ckpt (['state_dict']) and .safetensors, strict=False with missing/unexpected reporting (ddpm.py:321-344)."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    vdd = dict(synth.SD15_VAE_DD, ch=32, resolution=64)
    cfgs = ({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": vdd, "embed_dim": 4}},
            {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": NARROW})
    sd = {**synth.synthetic_unet_state_dict(NARROW), **synth.synthetic_vae_state_dict(vdd)}
    sd["cond_stage_model.transformer.text_model.final_layer_norm.weight"] = torch.ones(4)     # an "unexpected" key
    ck = os.path.join(tmp_path, "m.ckpt")
    torch.save({"state_dict": sd, "global_step": 7}, ck)
    ld = LatentDiffusion.hot_path(*cfgs)
    missing, unexpected = ld.init_from_ckpt(ck)
    assert unexpected == ["cond_stage_model.transformer.text_model.final_layer_norm.weight"]
    buffers = {n for n, _ in ld.named_buffers()}
    assert all(k.startswith("first_stage_model.post_quant_conv") or k in buffers for k in missing)
    k = "model.diffusion_model.output_blocks.5.1.transformer_blocks.0.attn2.to_k.weight"
    assert torch.equal(ld.state_dict()[k], sd[k])
    from safetensors.torch import save_file
    st = os.path.join(tmp_path, "m.safetensors")
    save_file({k_: v.contiguous() for k_, v in sd.items()}, st)
    ld2 = LatentDiffusion.hot_path(*cfgs)
    ld2.init_from_ckpt(st, ignore_keys=["cond_stage_model"])
    assert torch.equal(ld2.state_dict()[k], sd[k])
    ld2.freeze_unet()
    assert not any(p.requires_grad for p in ld2.model.parameters())
    assert not any(p.requires_grad for p in ld2.first_stage_model.parameters())


def test_no_cpu_fallback():
    """the product path refuses CPU tensors instead of silently computing somewhere else."""
    from adaprompt_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from adaprompt_amd.ldm.models.autoencoder import AutoencoderKL
    m = UNetModel(**NARROW)
    extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1}
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        m(torch.zeros(1, 4, 64, 64), torch.zeros(1, dtype=torch.long), context=torch.zeros(16, 77, 128), extra_info=extra)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 4, 64, 64), torch.zeros(1, dtype=torch.long), context=torch.zeros(16, 77, 128),
          extra_info={"use_layerwise_context": False})
    v = AutoencoderKL(dict(synth.SD15_VAE_DD, ch=32, resolution=64), None, 4)
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        v.encode(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="without its decoder"):      # training replica: no decoder weights carried
        v.decode(torch.zeros(1, 4, 8, 8))
    v2 = AutoencoderKL(dict(synth.SD15_VAE_DD, ch=32, resolution=64), None, 4, with_decoder=True)
    with pytest.raises(RuntimeError, match="HIP kernels only"):          # CPU tensor: no fallback on the decode path
        v2.decode(torch.zeros(1, 4, 8, 8))


def test_hook_standin_contract():
    from adaprompt_amd.hook_standin import SyntheticSubjBasisGenerator, make_cond_fn
    h = SyntheticSubjBasisGenerator(n_params=4 * 16 * 77 * 32, dim=32)
    ids = torch.nn.functional.normalize(torch.randn(3, 512), dim=-1)
    ctx, prompts, extra = make_cond_fn(h)({"zs_id_embs": ids})
    assert ctx.shape == (48, 77, 32) and extra["use_layerwise_context"] and extra["capture_distill_attn"]
    ctx.sum().backward()
    assert h.bases.grad is not None and h.bases.grad.abs().sum() > 0


def test_prodigy_linear_schedule_matches_reference_lrs():
    """SequentialLR2 + ConstantLR + PolynomialLR chain (reference util.py:26-41, ddpm.py:5219-5247) against the LR
    sequence the reference's own classes produced (tests/golden/prodigy_zs_clip0.npz: lrs)."""
    import torch
    from conftest import load_golden
    from ldm.util import prodigy_linear_schedule, SequentialLR2
    g = load_golden("prodigy_zs_clip0")
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sched = prodigy_linear_schedule(opt, max_steps=8, warm_up_steps=2, scheduler_cycles=1)
    assert isinstance(sched, SequentialLR2)
    lrs = []
    for _ in range(8):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert max(abs(a - float(b)) for a, b in zip(lrs, g["lrs"])) < 1e-12
    assert lrs[0] == lrs[1] == lrs[2] == 1.0 and lrs[-1] < lrs[-2] < 1.0


def test_prodigy_constructor_validation_matches_reference():
    """same ValueErrors as ldm/prodigy.py:64-73; FSDP is refused instead of silently ignored."""
    import pytest
    import torch
    from ldm.prodigy import Prodigy
    p = [torch.nn.Parameter(torch.zeros(2))]
    for bad in (dict(d0=0.0), dict(lr=0.0), dict(eps=0.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0))):
        with pytest.raises(ValueError):
            Prodigy(p, **bad)
    with pytest.raises(NotImplementedError):
        Prodigy(p, fsdp_in_use=True)
    opt = Prodigy(p, d_coef=2.0, use_bias_correction=True)
    g = opt.param_groups[0]
    assert g["d"] == g["d0"] == g["d_max"] == 1e-6 and g["k"] == 0 and g["d_numerator"] == 0.0 and g["d_coef"] == 2.0


def test_diffusers_key_map_is_a_bijection_onto_the_ldm_names():
    from adaprompt_amd.synth import SD15_UNET, unet_param_shapes
    from adaprompt_amd.ldm.modules.diffusionmodules.openaimodel import (ldm_to_diffusers_unet_key,
                                                                        diffusers_to_ldm_unet_state_dict)
    import torch
    shapes = dict(unet_param_shapes(**dict(SD15_UNET)))
    fwd = {k: ldm_to_diffusers_unet_key(k) for k in shapes}
    assert len(fwd) == 686 and len(set(fwd.values())) == 686
    known = {"input_blocks.0.0.weight": "conv_in.weight",
             "time_embed.2.bias": "time_embedding.linear_2.bias",
             "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight":
                 "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight",
             "input_blocks.3.0.op.bias": "down_blocks.0.downsamplers.0.conv.bias",
             "input_blocks.10.0.emb_layers.1.weight": "down_blocks.3.resnets.0.time_emb_proj.weight",
             "middle_block.2.out_layers.3.weight": "mid_block.resnets.1.conv2.weight",
             "output_blocks.2.1.conv.weight": "up_blocks.0.upsamplers.0.conv.weight",
             "output_blocks.5.2.conv.weight": "up_blocks.1.upsamplers.0.conv.weight",
             "output_blocks.11.0.skip_connection.weight": "up_blocks.3.resnets.2.conv_shortcut.weight",
             "out.0.weight": "conv_norm_out.weight", "out.2.bias": "conv_out.bias"}
    for k, v in known.items():
        assert fwd[k] == v
    # a diffusers-named dict (meta tensors, linear-projection form for one proj_in) comes back under ldm names/shapes
    dsd = {fwd[k]: torch.empty(s, device="meta") for k, s in shapes.items()}
    k = "input_blocks.1.1.proj_in.weight"
    dsd[fwd[k]] = torch.empty(shapes[k][:2], device="meta")
    back = diffusers_to_ldm_unet_state_dict(dsd)
    assert set(back) == set(shapes) and all(tuple(back[n].shape) == tuple(shapes[n]) for n in shapes)
    del dsd[fwd["out.2.bias"]]
    import pytest
    with pytest.raises(KeyError):
        diffusers_to_ldm_unet_state_dict(dsd)


def test_multistep_host_helpers_match_oracle():
    import numpy as np
    import torch
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion, Arc2FaceWrapper
    from oracle import distill_oracle as D
    for bs in (3, 4):
        for nd in (1, 3, 5, 7):
            assert LatentDiffusion.half_batch_size(bs, nd) == D.half_batch_size(bs, nd)
    rs = np.random.RandomState(5)
    draws = [LatentDiffusion.draw_num_denoising_steps(7, rs) for _ in range(4000)]
    freq = [draws.count(s) / 4000 for s in (1, 3, 5, 7)]
    assert max(abs(a - b) for a, b in zip(freq, (0.4, 0.3, 0.2, 0.1))) < 0.03
    assert set(LatentDiffusion.draw_num_denoising_steps(3, rs) for _ in range(50)) == {1, 3}
    ctx = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    lw = Arc2FaceWrapper.layerwise(ctx)
    assert lw.shape == (32, 3, 4) and torch.equal(lw[0], ctx[0]) and torch.equal(lw[15], ctx[0]) \
        and torch.equal(lw[16], ctx[1])


def test_mirror_probably_anneal_t_matches_reference_draws():
    """adaprompt_amd.ldm.util.probably_anneal_t on host tensors consumes ``random`` / ``np.random`` exactly as the
    reference (golden: tests/golden/anneal_t.npz, ldm/util.py:1468-1530); on the device it is the same distribution."""
    import random

    import numpy as np
    import torch
    from conftest import load_golden
    from adaprompt_amd.ldm import util as U
    g = load_golden("anneal_t")
    for row, want in zip(g["cases"].tolist(), g["outs"].tolist()):
        seed, tp, lb, ub, k0, k1 = row[:6]
        t = torch.tensor([int(v) for v in row[6:]])
        random.seed(100 + int(seed))
        np.random.seed(200 + int(seed))
        lb = int(lb) if lb == int(lb) else lb
        got = U.probably_anneal_t(t, tp, 1000, ratio_range=(lb, ub), keep_prob_range=(k0, k1))
        assert got.tolist() == want, (row, got.tolist(), want)
    av = [U.anneal_value(tp, fp, (0.2, 0.9)) for tp in (0.0, 0.3, 0.5, 1.0) for fp in (0.5, 1.0)]
    np.testing.assert_allclose(av, g["anneal_values"].numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(U.anneal_array(0.25, 0.5, [0.4, 0.3, 0.2, 0.1], [0.1, 0.2, 0.3, 0.4]),
                               g["anneal_array"].numpy(), rtol=0, atol=0)


def test_conditioning_helpers_match_reference_golden():
    """the host-side helpers of LatentDiffusion.forward's conditioning assembly (ddpm.py:1710-2042) against fixtures
    captured from the reference's own ldm/util.py with the same seeds."""
    import random

    import numpy as np
    import torch
    from conftest import load_golden
    from adaprompt_amd import synth
    from adaprompt_amd.ldm import util as U
    g = load_golden("cond_helpers")

    def close(x, y):
        return float((x.double() - y.double()).abs().max()) <= 1e-6 * float(y.double().abs().max()) + 1e-9

    a, b = synth.synthetic_input("ch.a", (4, 3, 5)), synth.synthetic_input("ch.b", (4, 7))
    r = U.repeat_selected_instances(slice(0, 2), 3, a, None, b)
    assert r[1] is None and torch.equal(r[0], g["rep_a"]) and torch.equal(r[2], g["rep_b"])
    emb = synth.synthetic_input("ch.emb", (6, 32))
    torch.manual_seed(11)
    assert close(U.add_noise_to_tensor(emb, 0.1, noise_std_is_relative=True, keep_norm=False), g["noise_rel"])
    torch.manual_seed(12)
    assert close(U.add_noise_to_tensor(emb, 0.05, noise_std_is_relative=False, keep_norm=True), g["noise_keepnorm"])
    for i, (tp, prob) in enumerate(((0.0, 1.0), (0.6, 0.5), (0.3, 0.0))):
        random.seed(30 + i)
        np.random.seed(40 + i)
        torch.manual_seed(50 + i)
        got = U.anneal_add_noise_to_embedding(emb, tp, begin_noise_std_range=[0.02, 0.06], end_noise_std_range=[0.01, 0.03],
                                              add_noise_prob=prob)
        assert close(got, g[f"anneal_noise_{i}"]), i
    random.seed(33)
    np.random.seed(43)
    torch.manual_seed(53)
    assert close(U.anneal_add_noise_to_embedding(emb, 0.5, begin_noise_std_range=[0.02, 0.06], end_noise_std_range=None,
                                                 add_noise_prob=1.0), g["anneal_noise_noend"])
    te = synth.synthetic_input("ch.te", (16, 77, 24))
    idx = torch.tensor([5, 6, 7, 8, 5, 6, 7, 8])
    assert close(U.distribute_embedding_to_M_tokens(te, idx), g["dist_sqrt"])
    assert close(U.distribute_embedding_to_M_tokens(te, idx, divide_scheme="M"), g["dist_M"])
    assert torch.equal(U.distribute_embedding_to_M_tokens(te, torch.tensor([9])), g["dist_single"])
    d = {"z": (torch.zeros(4, dtype=torch.long), torch.tensor([5, 6, 7, 8])), "y": None,
         "w": (torch.zeros(1, dtype=torch.long), torch.tensor([20]))}
    assert close(U.distribute_embedding_to_M_tokens_by_dict(te, d), g["dist_dict"])


def test_iteration_flag_draw_follows_the_reference_order():
    """ddpm.py:516-572: compositional iterations on every gap-th global step (they consume np.random.choice and skip the
    arc2face draw), otherwise one np.random.rand against arc2face_distill_iter_prob, which also disables the static
    prompt-delta loss."""
    import numpy as np
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion

    class Stub:
        prompt_emb_delta_reg_weight = 2e-4

    class Rec:
        def __init__(self, vals):
            self.vals, self.calls = list(vals), []

        def rand(self):
            self.calls.append("rand")
            return self.vals.pop(0)

        def choice(self, n, p=None):
            self.calls.append(("choice", n))
            return 0

    draw = LatentDiffusion.draw_iteration_flags
    r = Rec([0.05])
    f = draw(Stub(), 7, composition_regs_iter_gap=3, arc2face_distill_iter_prob=0.1, np_random=r)
    assert r.calls == ["rand"] and f["do_arc2face_distill"] and not f["do_static_prompt_delta_reg"] and f["do_normal_recon"]
    r = Rec([0.5])
    f = draw(Stub(), 7, composition_regs_iter_gap=3, arc2face_distill_iter_prob=0.1, np_random=r)
    assert r.calls == ["rand"] and not f["do_arc2face_distill"] and f["do_static_prompt_delta_reg"]
    r = Rec([0.0])
    f = draw(Stub(), 6, composition_regs_iter_gap=3, arc2face_distill_iter_prob=0.1, np_random=r)
    assert r.calls == [("choice", 1)] and f["is_compos_iter"] and not f["do_normal_recon"] and f["do_ada_prompt_delta_reg"] \
        and not f["do_mix_prompt_distillation"] and not f["do_arc2face_distill"]
    f = draw(Stub(), 6, composition_regs_iter_gap=3, mix_prompt_distill_weight=1e-4, np_random=Rec([]))
    assert f["do_mix_prompt_distillation"] and f["calc_clip_loss"]
    r = Rec([])
    f = draw(Stub(), 6, composition_regs_iter_gap=0, arc2face_distill_iter_prob=0.0, np_random=r)
    assert r.calls == [] and f["do_normal_recon"] and not f["is_compos_iter"]
    # with the real generator: the share of distillation iterations is the configured probability
    np.random.seed(0)
    n = sum(draw(Stub(), 1, arc2face_distill_iter_prob=0.3)["do_arc2face_distill"] for _ in range(2000))
    assert 520 < n < 680


def test_configure_optimizers_prodigy_branch():
    """ddpm.py:5134-5345, Prodigy branch: one flat lr-1 parameter list of the not-excluded, requires-grad parameters,
    zero-shot betas, d_coef, bias correction, ConstantLR warm-up + linear cycle(s) under SequentialLR2."""
    import torch
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaprompt_amd.ldm.prodigy import Prodigy
    from adaprompt_amd.ldm.util import SequentialLR2

    class Stub:
        optimizer_type, do_zero_shot = "Prodigy", True
        model = torch.nn.Linear(3, 2)

    a, b, c, d = (torch.nn.Parameter(torch.zeros(n)) for n in (4, 5, 6, 7))
    b.requires_grad_(False)
    groups = [{"params": [a, b], "lr_ratio": 1.0, "excluded_from_prodigy": False},
              {"params": [c], "lr_ratio": 0.1, "excluded_from_prodigy": True},
              {"params": [d], "lr_ratio": 2.0, "excluded_from_prodigy": False}]
    out = LatentDiffusion.configure_optimizers(Stub(), groups, max_steps=2000,
                                               prodigy_config={"warm_up_steps": 500, "scheduler_cycles": 2, "d_coef": 5})
    assert len(out) == 1 and out[0]["frequency"] == 1 and out[0]["lr_scheduler"]["interval"] == "step"
    opt, sched = out[0]["optimizer"], out[0]["lr_scheduler"]["scheduler"]
    assert isinstance(opt, Prodigy) and isinstance(sched, SequentialLR2)
    got = [q for g in opt.param_groups for q in g["params"]]
    assert len(got) == 2 and got[0] is a and got[1] is d
    g0 = opt.param_groups[0]
    assert g0["lr"] == 1.0 and tuple(g0["betas"]) == (0.9, 0.999) and g0["d_coef"] == 5 and g0["use_bias_correction"] \
        and g0["safeguard_warmup"]
    assert list(sched._milestones) == [500, 1250]            # 2 cycles of 750 steps after the warm-up
    Stub.do_zero_shot = False
    out = LatentDiffusion.configure_optimizers(Stub(), groups, max_steps=2000, unfreeze_model=True)
    opt = out[0]["optimizer"]
    got = [q for g in opt.param_groups for q in g["params"]]
    assert len(got) == 4 and got[2] is Stub.model.weight and tuple(opt.param_groups[0]["betas"]) == (0.985, 0.993)
    Stub.optimizer_type = "ProdigyAdamW"          # the two-optimiser variant (ddpm.py:5274-5302) is refused loudly
    with pytest.raises(NotImplementedError):
        LatentDiffusion.configure_optimizers(Stub(), groups, max_steps=10)


def test_on_save_checkpoint_contract(tmp_path):
    """ddpm.py:5393-5400: frozen UNet -> the checkpoint dict is emptied; the embedding manager writes its two files."""
    import torch
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion

    class Stub:
        model = torch.nn.Linear(2, 2)

    class Mgr:
        def __init__(self):
            self.saved = []

        def save(self, path):
            self.saved.append(path)

    st, mgr = Stub(), Mgr()
    LatentDiffusion.freeze_unet(st)
    ck = {"state_dict": {"a": 1}, "epoch": 3}
    LatentDiffusion.on_save_checkpoint(st, ck, mgr, str(tmp_path), 1200)
    assert ck == {} and [p.split("/")[-1] for p in mgr.saved] == ["embeddings.pt", "embeddings_gs-1200.pt"]
    st.unfreeze_model = True
    ck = {"state_dict": {"a": 1}}
    LatentDiffusion.on_save_checkpoint(st, ck)
    assert ck == {"state_dict": {"a": 1}}


def test_training_step_auto_iteration_dispatch():
    """the preamble of the reference's training_step (ddpm.py:516-572, 1839-1859) drives shared_step: distillation
    iterations get use_arc2face_as_target + a drawn ND, recon iterations get timestep annealing, compositional
    iterations are refused; training_percent follows the global step."""
    import numpy as np
    import pytest
    import torch
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion

    class Stub:
        manual_accumulate_grad_batches, prompt_emb_delta_reg_weight, batch_idx = 2, 2e-4, 0
        do_static_prompt_delta_reg = True
        cond_fn = staticmethod(lambda batch: None)          # the context comes from a cond_fn: training_step draws ND itself
        draw_iteration_flags = LatentDiffusion.draw_iteration_flags
        init_iteration_flags = LatentDiffusion.init_iteration_flags
        draw_num_denoising_steps = staticmethod(LatentDiffusion.draw_num_denoising_steps)
        manual_backward = staticmethod(lambda *a: None)
        _iteration_preamble = LatentDiffusion._iteration_preamble
        _micro_batch_backward = LatentDiffusion._micro_batch_backward

        def __init__(self):
            self.calls = []

        def shared_step(self, batch, **kw):
            self.calls.append(kw)
            return torch.tensor(0.0), None, torch.zeros(1), {}

    st = Stub()
    np.random.seed(3)
    kinds = []
    for i in range(40):
        LatentDiffusion.training_step(st, {}, auto_iteration={"max_steps": 100, "arc2face_distill_iter_prob": 0.5,
                                                               "max_num_denoising_steps": 5})
        kw = st.calls[-1]
        if kw.get("use_arc2face_as_target"):
            # distillation iterations are normal-recon iterations: t is annealed too (reference ddpm.py:2851-2861)
            assert kw["num_denoising_steps"] in (1, 3, 5) and kw["anneal_t"] is True
            assert st.iter_flags["do_arc2face_distill"] and not st.iter_flags["do_static_prompt_delta_reg"]
            kinds.append(1)
        else:
            assert kw == {"anneal_t": True}
            kinds.append(0)
    assert 10 < sum(kinds) < 30 and st.batch_idx == 40 and abs(st.training_percent - 0.38) < 1e-9
    with pytest.raises(NotImplementedError):
        st.batch_idx = 12                                             # global step 6, a multiple of the gap
        LatentDiffusion.training_step(st, {}, auto_iteration={"max_steps": 100, "composition_regs_iter_gap": 3})


def test_hostinfo_cpu_share_reads_the_cgroup_quota(tmp_path):
    """the CPU oracle's thread count on a GPU box: 256 logical CPUs, cgroup quota 16 (profiles/r03_cpu_threads_probe.log)."""
    import os
    import torch
    from adaprompt_amd import hostinfo
    aff = len(os.sched_getaffinity(0))
    assert hostinfo.cpu_share(str(tmp_path)) == aff                       # no cgroup files: the affinity
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert hostinfo.cpu_share(str(tmp_path)) == aff                       # v2, unlimited
    (tmp_path / "cpu.max").write_text("200000 100000\n")
    assert hostinfo.cpu_share(str(tmp_path)) == min(aff, 2)               # v2 quota
    (tmp_path / "cpu.max").write_text("50000 100000\n")
    assert hostinfo.cpu_share(str(tmp_path)) == 1                         # never below one
    (tmp_path / "cpu.max").unlink()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert hostinfo.cpu_share(str(tmp_path)) == min(aff, 3)               # v1 quota
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert hostinfo.cpu_share(str(tmp_path)) == aff                       # v1, unlimited
    before = torch.get_num_threads()
    try:
        assert hostinfo.limit_torch_threads(cap=1) == 1 and torch.get_num_threads() == 1
    finally:
        torch.set_num_threads(before)


def test_trainer_fit_window_loop_host_logic():
    """``Trainer.fit`` = Lightning's loop over ``training_step(batch, batch_idx)``; on lanes that entry buffers a window's
    micro-batches (``_training_step_in_windows``), optionally with windows of latents prefetched ahead, and ``flush_window``
    finishes an epoch.  Host logic only (the window itself is a stub): partial windows, ``max_steps``, checkpoint cadence,
    every batch encoded / consumed / logged exactly once and in order, the lanes' gates removed at the end."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaprompt_amd.trainer import Trainer

    class FakePrefetcher:
        def __init__(self, log):
            self.q, self.log = [], log

        def submit(self, batch, post_noise=None):
            self.log.append(("submit", batch["i"]))
            self.q.append(batch["i"])

        def get(self):
            self.log.append(("get", self.q[0]))
            return self.q.pop(0)

    class Lanes:
        removed = False

        def remove(self):
            self.removed = True

    class Stub:
        manual_accumulate_grad_batches = 2
        composition_regs_iter_gap = arc2face_distill_iter_prob = mix_prompt_distill_weight = 0
        max_num_denoising_steps = 1
        training_step = LatentDiffusion.training_step
        _training_step_in_windows = LatentDiffusion._training_step_in_windows
        _run_buffered_window = LatentDiffusion._run_buffered_window
        flush_window = LatentDiffusion.flush_window

        def __init__(self):
            self.batch_idx, self.log, self.saves, self.steps = 0, [], [], 0

        @property
        def global_step(self):
            return self.batch_idx // self.manual_accumulate_grad_batches

        def make_prefetcher(self):
            return FakePrefetcher(self.log)

        def training_window(self, batches, optimizer, reducer, scheduler, lanes, auto_iteration=None, step_kwargs=None,
                            after_backward=None):
            assert self.batch_idx % 2 == 0 and len(batches) == 2 and lanes is not None and auto_iteration["max_steps"] > 0
            got = [step_kwargs(k)["x_start"] if step_kwargs is not None else None for k in range(2)]
            self.log.append(("window", [b["i"] for b in batches], got))
            self.batch_idx += 2
            self.steps += 1
            if after_backward is not None:
                for k in range(2):
                    after_backward(k)
            return [(float(b["i"]), {}) for b in batches]

        def _iteration_preamble(self, auto, kw):
            pass

        def shared_step(self, batch, **kw):
            self.log.append(("single", batch["i"], kw.get("x_start")))
            return float(batch["i"]), None, None, {}

        def _micro_batch_backward(self, *a, **k):
            pass

        def _optimizer_step(self, *a):
            self.steps += 1

        def on_save_checkpoint(self, ckpt):
            self.saves.append(self.global_step)

    def fit(n_batches, max_steps, look, every=2):
        m, lanes = Stub(), Lanes()
        tr = Trainer(max_steps=max_steps, every_n_train_steps=every, micro_batch_lanes=True, prefetch_windows=look)
        tr.optimizer = type("O", (), {"param_groups": []})()
        tr.lanes = lanes
        object.__setattr__(m, "trainer", tr)
        import adaprompt_amd.ops as ops_mod
        was, ops_mod.gn_sync_poisoned = ops_mod.gn_sync_poisoned, lambda: False          # (save_checkpoint's device read)
        try:
            logged = tr.fit(m, ({"i": i} for i in range(n_batches)))
        finally:
            ops_mod.gn_sync_poisoned = was
        assert lanes.removed and tr.lanes is None
        return m, logged

    # no prefetch: windows as their last micro-batch arrives, the odd tail on one stream, window left open
    m, logged = fit(7, 10, 0)
    assert logged == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0] and m.batch_idx == 7 and m.steps == 3
    assert [e for e in m.log if e[0] != "single"] == [("window", [0, 1], [None, None]), ("window", [2, 3], [None, None]),
                                                        ("window", [4, 5], [None, None])]
    assert m.log[-1] == ("single", 6, None) and m.saves == [2]
    # one window of latents ahead: every batch submitted once, in order, each window consumes ITS latents
    m, logged = fit(7, 10, 1)
    assert logged == [float(i) for i in range(7)] and m.batch_idx == 7
    assert [e[1] for e in m.log if e[0] == "submit"] == list(range(7))
    wins = [e for e in m.log if e[0] == "window"]
    assert wins == [("window", [0, 1], [0, 1]), ("window", [2, 3], [2, 3]), ("window", [4, 5], [4, 5])]
    assert m.log[-1] == ("single", 6, 6)                   # (the tail's latent was prefetched behind the last window too)
    # the encodes of window w+1 are submitted behind window w's backwards, not before it
    i_w0 = m.log.index(wins[0])
    assert ("submit", 2) in m.log[i_w0:] and ("submit", 2) not in m.log[:i_w0]
    # max_steps reached with batches still buffered: they are dropped, the prefetch queue is drained
    m, logged = fit(9, 2, 1, every=1)
    assert m.global_step == 2 and logged == [0.0, 1.0, 2.0, 3.0] and m.saves == [1, 2]
    assert not m._win_entry["buf"] and not m._win_entry["pf"].q and m._win_entry["submitted"] == 0

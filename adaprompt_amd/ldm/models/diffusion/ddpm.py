"""The diffusion trainer's hot path on the MI355X kernels (reference
ldm/models/diffusion/ddpm.py): schedule buffers (``register_schedule`` :240-292), ``q_sample``
(:416-419), ``predict_start_from_noise`` (:358-362), first-stage encode + posterior sample
(:955-962, :1381-1419, :1178-1256), ``apply_model`` / ``DiffusionWrapper`` (:2192-2297,
:5505-5544), ``guided_denoise`` (:2483-2532), ``calc_recon_loss`` (:3571-3595), the recon
iteration's regularisers (:3207-3270, ``calc_fg_bg_xlayer_consist_loss`` :4259-4387) and the
manual-optimisation ``training_step`` (:515-638: backward every micro-batch, clip 0.5 + step +
zero_grad every ``manual_accumulate_grad_batches``-th batch, loss not divided).

What is deliberately NOT here (SURVEY.md section 2 "out of scope" / section 8f "next"): Lightning, the
data pipeline, the CLIP text encoder + EmbeddingManager + SubjBasisGenerator internals (they stay
the reference's own classes behind ``cond_fn``), the Arc2Face text encoder (the teacher's context is an input),
compositional distillation and its auxiliary losses.  The Arc2Face teacher ROLLOUT and the
multi-step distillation loss (ddpm.py:5432-5478, 2950-3039) are here: the teacher is an SD-1.5-topology UNet and
runs on the same kernels (``Arc2FaceWrapper``).  ``cond_fn(batch) -> (c_static_emb
[16*B, 77, 768], prompts, extra_info)`` is the embedding hook: whatever produced the context (the
reference's ``get_learned_conditioning``) is called as-is and only its output enters the path.
"""
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ...modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from ...modules.distributions.distributions import DiagonalGaussianDistribution
from ...util import default, instantiate_from_config
from .conditioning import ConditioningMixin

STAGE2_FUSED = True       # CUDA tensors take the one-launch HIP forms of the Stage-2 per-layer terms (csrc/stage2loss.hip); False:
                          # the torch expressions (what CPU tensors always take) -- tests/test_stage2loss_gpu.py compares the two

# the recon iteration's attention regularisers as one call of the HIP library (LatentDiffusion.fused_token_map_losses);
# ADAP_FUSED_REG=0 runs the host expressions instead (the checker of the fused form in tests/)
FUSED_REG_LOSSES = os.environ.get("ADAP_FUSED_REG", "1") != "0"


# in a recon iteration whose regularisers are fused, the capture kernel forms the token maps only (the dense attnscore / attn /
# q tensors, 86 MB per 64 x 64 layer at bs 4, are never read); ADAP_CAPTURE_DENSE=1 keeps writing them
CAPTURE_TOKEN_MAPS_ONLY = os.environ.get("ADAP_CAPTURE_DENSE", "0") != "1"


def set_fused_reg_losses(on):
    global FUSED_REG_LOSSES
    FUSED_REG_LOSSES = bool(on)


class DiffusionWrapper(nn.Module):
    """reference ddpm.py:5505-5544 for conditioning_key='crossattn' with the AdaFace cond triple."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert conditioning_key == "crossattn", "SD-1.5 / AdaFace uses cross-attention conditioning"

    def forward(self, x, t, c_concat=None, c_crossattn=None, c_in=None, extra_info=None):
        assert c_concat is None and c_crossattn is not None
        c = c_crossattn[0]
        if isinstance(c, (tuple, list)):
            # the AdaFace cond triple (c_static_emb, prompts, extra_info), ddpm.py:5523-5533
            cc, c_in, extra_info = c
        else:
            cc = torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc, context_in=c_in, extra_info=extra_info)


class DDPM(nn.Module):
    """Schedule buffers + q_sample / predict_x0 + the manual-optimisation step bookkeeping."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None, ignore_keys=(),
                 load_only_unet=False, monitor="val/loss", use_ema=False, first_stage_key="image", image_size=256, channels=3,
                 log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, given_betas=None,
                 original_elbo_weight=0., unfreeze_model=False, model_lr=0., v_posterior=0., conditioning_key="crossattn",
                 parameterization="eps", optimizer_type="Prodigy", grad_clip=0.5, manual_accumulate_grad_batches=2,
                 adam_config=None, prodigy_config=None, use_positional_encodings=False, learn_logvar=False, logvar_init=0.,
                 use_layerwise_embedding=True, composition_regs_iter_gap=3, static_embedding_reg_weight=0.,
                 prompt_emb_delta_reg_weight=2e-4, mix_prompt_distill_weight=0., comp_fg_bg_preserve_loss_weight=0.,
                 fg_bg_complementary_loss_weight=2e-4, fg_wds_complementary_loss_weight=0.,
                 fg_bg_xlayer_consist_loss_weight=5e-5, compel_cfg_weight_level_range=(2, 2), apply_compel_cfg_prob=0.,
                 wds_bg_recon_discount=1., do_clip_teacher_filtering=True, num_candidate_teachers=2,
                 use_background_token=True, use_fp_trick=True, normalize_ca_q_and_outfeat=True, do_zero_shot=True,
                 arc2face_distill_iter_prob=0, apply_arc2face_inverse_embs=False, p_gen_arc2face_rand_face=0.4,
                 p_add_noise_to_real_id_embs=0.6, max_num_denoising_steps=5,
                 extend_prompt2token_proj_attention_multiplier=-1, load_old_embman_ckpt=False):
        """The reference's keyword surface (ddpm.py:76-130), so that the ``params`` block of v1-finetune-ada.yaml:5-84
        instantiates this class unchanged.  Every keyword is stored under the reference's attribute name; the ones whose
        feature is not built (EMA, learned log-variance, x0 parameterisation, positional encodings) are refused loudly
        instead of being ignored.  Defaults differ from the reference only where the reference's default is never what the
        yaml ships AND this package's tests relied on the shipped value: the three recon-regulariser weights (yaml:40,48,50),
        ``manual_accumulate_grad_batches`` (yaml:62) and ``use_layerwise_embedding`` (yaml:28)."""
        super().__init__()
        assert parameterization == "eps", "SD-1.5 / AdaFace predicts eps; 'x0' is not built"
        if use_ema:
            raise NotImplementedError("use_ema: True (LitEma) is out of scope -- the shipped config has use_ema: False (yaml:25)")
        if learn_logvar or use_positional_encodings:
            raise NotImplementedError("learn_logvar / use_positional_encodings are unused by the SD-1.5 config and not built")
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised, self.log_every_t, self.first_stage_key = clip_denoised, log_every_t, first_stage_key
        self.image_size, self.channels = image_size, channels
        self.use_layerwise_embedding = use_layerwise_embedding
        self.N_CA_LAYERS = 16 if use_layerwise_embedding else 1
        self.do_zero_shot = do_zero_shot
        self.static_embedding_reg_weight = static_embedding_reg_weight
        self.composition_regs_iter_gap = composition_regs_iter_gap
        self.prompt_emb_delta_reg_weight = prompt_emb_delta_reg_weight
        self.mix_prompt_distill_weight = mix_prompt_distill_weight
        self.comp_fg_bg_preserve_loss_weight = comp_fg_bg_preserve_loss_weight
        self.fg_bg_complementary_loss_weight = fg_bg_complementary_loss_weight
        self.fg_wds_complementary_loss_weight = fg_wds_complementary_loss_weight
        self.fg_bg_xlayer_consist_loss_weight = fg_bg_xlayer_consist_loss_weight
        self.compel_cfg_weight_level_range = None if compel_cfg_weight_level_range is None else list(compel_cfg_weight_level_range)
        self.empty_context = None
        self.apply_compel_cfg_prob = apply_compel_cfg_prob
        self.do_clip_teacher_filtering = do_clip_teacher_filtering
        self.num_candidate_teachers = num_candidate_teachers
        self.prompt_mix_scheme = "mix_hijk"
        self.wds_bg_recon_discount = wds_bg_recon_discount
        self.use_background_token = use_background_token
        self.use_fp_trick = use_fp_trick
        self.normalize_ca_q_and_outfeat = normalize_ca_q_and_outfeat
        self.arc2face_distill_iter_prob = arc2face_distill_iter_prob if do_zero_shot else 0
        self.apply_arc2face_inverse_embs = apply_arc2face_inverse_embs
        self.p_gen_arc2face_rand_face = p_gen_arc2face_rand_face
        self.p_add_noise_to_real_id_embs = p_add_noise_to_real_id_embs
        self.max_num_denoising_steps = max_num_denoising_steps
        self.extend_prompt2token_proj_attention_multiplier = extend_prompt2token_proj_attention_multiplier
        self.load_old_embman_ckpt = load_old_embman_ckpt
        self.cached_inits = {}
        self.do_static_prompt_delta_reg = prompt_emb_delta_reg_weight >= 0
        self.init_iteration_flags()
        self.is_dreambooth = False
        self.v_posterior = v_posterior
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = use_ema
        self.optimizer_type = optimizer_type
        self.adam_config = adam_config
        self.prodigy_config = prodigy_config if "Prodigy" in optimizer_type else None
        self.grad_clip = grad_clip
        self.manual_accumulate_grad_batches = manual_accumulate_grad_batches
        self.automatic_optimization = False
        self.training_percent = 0.
        self.original_elbo_weight = original_elbo_weight
        self.unfreeze_model = unfreeze_model
        self.model_lr = model_lr
        self.monitor = monitor
        self.loss_type = loss_type
        self.register_schedule(given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys, only_model=load_only_unet)

    def init_iteration_flags(self):
        """ddpm.py:484-506."""
        self.iter_flags = {"calc_clip_loss": False, "do_normal_recon": True, "do_arc2face_distill": False,
                           "gen_arc2face_rand_face": False, "arc2face_prompt_emb": None, "add_noise_to_real_id_embs": False,
                           "faceless_img_count": 0, "use_arc2face_as_target": False, "num_denoising_steps": 1,
                           "is_compos_iter": False, "do_mix_prompt_distillation": False,
                           "do_static_prompt_delta_reg": self.do_static_prompt_delta_reg, "do_ada_prompt_delta_reg": False,
                           "use_background_token": False, "use_wds_comp": False, "use_wds_cls_captions": False,
                           "use_fp_trick": False, "reuse_init_conds": False, "comp_init_fg_from_training_image": False}

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", f32(alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(alphas_cumprod)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - alphas_cumprod)))
        self.register_buffer("log_one_minus_alphas_cumprod", f32(np.log(1.0 - alphas_cumprod)))
        self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1.0 / alphas_cumprod)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1.0 / alphas_cumprod - 1)))

    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        return ops.q_sample(x_start, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)


_CONSTS = {}


def _const_tensor(values, device):
    """a small constant f32 tensor on ``device``, created once (``torch.tensor(list, device=...)`` is a pageable
    host -> device copy, i.e. a stream synchronisation, every time it is called)."""
    key = (values, str(device))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, device=device, dtype=torch.float32)
    return t


class _CondRequest:
    """what ``_shared_step_gen`` yields where the sequential step computes its conditioning (``cond_fn(batch)`` on the batch as
    trimmed for the iteration, or the yaml-instantiated text encoder + embedding manager): ``make()`` computes it -- with the
    iteration flags of ITS micro-batch current -- on whatever stream the driver chooses."""

    def __init__(self, make):
        self.make = make


class LatentDiffusion(ConditioningMixin, DDPM):
    """The recon-distillation iteration core of the reference ``LatentDiffusion`` (ddpm.py:710-3457)."""

    def __init__(self, first_stage_config, cond_stage_config=None, personalization_config=None, num_timesteps_cond=None,
                 cond_stage_key="image", cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None,
                 conditioning_key=None, scale_factor=1.0, scale_by_std=False, is_dreambooth=False, *args,
                 cond_fn=None, bg_pixel_weight=0.1, **kwargs):
        """The reference's constructor (ddpm.py:710-810): ``first_stage_config``, ``cond_stage_config``,
        ``personalization_config`` and -- through ``**kwargs`` -- ``unet_config=`` and every ``DDPM`` keyword, so
        ``instantiate_from_config(yaml.model)`` builds this class from v1-finetune-ada.yaml as it stands.  The text encoder
        and the embedding manager named by the two configs are the reference's own classes (or whatever the targets resolve
        to) and are only called (``conditioning.ConditioningMixin``).  Two keyword-only extras: ``cond_fn(batch) -> cond``
        replaces the conditioning side wholesale (benchmarks / kernel tests without HF weights), ``bg_pixel_weight``."""
        self.num_timesteps_cond = default(num_timesteps_cond, 1)
        self.scale_by_std = scale_by_std
        assert self.num_timesteps_cond <= kwargs.get("timesteps", 1000)
        if scale_by_std:
            raise NotImplementedError("scale_by_std is unused by the SD-1.5 config (scale_factor 0.18215 is given) and not built")
        if is_dreambooth:
            raise NotImplementedError("is_dreambooth: the DreamBooth regularisation batches are out of scope (SURVEY.md section 2)")
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        super().__init__(*args, conditioning_key=conditioning_key, **kwargs)
        self.concat_mode = concat_mode
        self.cond_stage_trainable = cond_stage_trainable
        self.cond_stage_key = cond_stage_key
        self.cond_stage_forward = cond_stage_forward
        self.clip_denoised = False
        self.scale_factor = scale_factor
        self.bg_pixel_weight = bg_pixel_weight
        self.cond_fn = cond_fn
        # CLIP text-image similarity of decoded images (evaluation/clip_eval.py:CLIPEvaluator.txt_to_img_similarity with
        # reduction='diag' in the reference, ddpm.py:3624-3627): a callable behind the boundary, set by the trainer
        self.clip_score_fn = None
        # the zero-shot feature front end's third-party encoders (conditioning.set_zero_shot_image_encoders)
        self.clip_image_encoder = self.clip_preprocessor = self.insightface_app = None
        self.dino_encoder = self.dino_preprocess = self.neg_image_features = None
        self.zs_image_encoders_instantiated = False
        self.empty_context_2b = self.empty_context_tea_filter = None
        self.batch_idx = 0
        self.is_dreambooth = False
        self.instantiate_first_stage(first_stage_config)
        self.embedding_manager = None
        if cond_stage_config is not None:
            self.instantiate_cond_stage(cond_stage_config)
        self.restarted_from_ckpt = ckpt_path is not None
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)
        self.arc2face = None          # Arc2FaceWrapper, attached by set_arc2face_teacher (ddpm.py:771, 903-907)
        if not self.unfreeze_model:   # ddpm.py:775-786
            if self.cond_stage_model is not None:
                self.cond_stage_model.eval()
                for p in self.cond_stage_model.parameters():
                    p.requires_grad = False
            self.freeze_unet()
        if personalization_config is not None:
            self.embedding_manager = self.instantiate_embedding_manager(personalization_config, self.cond_stage_model)
            if self.do_zero_shot and hasattr(self.embedding_manager, "make_frozen_copy_of_subj_basis_generators"):
                self.embedding_manager.make_frozen_copy_of_subj_basis_generators()      # ddpm.py:797
        if self.cond_stage_model is not None and self.embedding_manager is not None:
            # ddpm.py:812-814: the empty prompt's context, one layer's worth
            self.empty_context = self.get_learned_conditioning([""], embman_iter_type="empty")[0][[0]]

    @classmethod
    def hot_path(cls, first_stage_config, unet_config, cond_fn=None, **kwargs):
        """the kernels-only construction used by bench.py / smoke() / the kernel tests: no text encoder, no embedding manager
        (the context comes from ``cond_fn`` or is passed to ``shared_step``), SD-1.5's schedule and scale factor."""
        kwargs.setdefault("linear_start", 0.00085)
        kwargs.setdefault("linear_end", 0.012)
        kwargs.setdefault("scale_factor", 0.18215)
        kwargs.setdefault("conditioning_key", "crossattn")
        kwargs.setdefault("unfreeze_model", True)         # leave requires_grad alone: callers call freeze_unet() themselves
        return cls(first_stage_config, unet_config=unet_config, cond_fn=cond_fn, **kwargs)

    # ---- ddpm.py:863-901 ---------------------------------------------------------------------------------------------
    def instantiate_first_stage(self, config):
        self.first_stage_model = instantiate_from_config(config).eval()
        for p in self.first_stage_model.parameters():
            p.requires_grad = False

    def instantiate_cond_stage(self, config):
        if config == "__is_unconditional__":
            self.cond_stage_model = None
            return
        if config == "__is_first_stage__":
            raise NotImplementedError("cond_stage_config '__is_first_stage__' is not an SD-1.5 configuration")
        model = instantiate_from_config(config)
        if not self.cond_stage_trainable:
            model = model.eval()
            for p in model.parameters():
                p.requires_grad = False
        self.cond_stage_model = model

    def instantiate_embedding_manager(self, config, text_embedder):
        model = instantiate_from_config(config, text_embedder=text_embedder)
        ckpt = (config.get("params") or {}).get("embedding_manager_ckpt", None)
        if ckpt:                      # not when missing OR empty
            model.load(ckpt, self.extend_prompt2token_proj_attention_multiplier, self.load_old_embman_ckpt)
        return model

    # ---- checkpoint API (ddpm.py:321-344): .ckpt ['state_dict'] or .safetensors, strict=False ----------
    def init_from_ckpt(self, path, ignore_keys=(), only_model=False):
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path, device="cpu")
        else:
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd)
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        target = self.model if only_model else self
        missing, unexpected = target.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        return missing, unexpected

    def set_arc2face_teacher(self, teacher):
        """attach the frozen teacher without registering its weights in this module's state dict (the reference
        keeps it out of checkpoints the same way, ddpm.py:5393-5400)."""
        object.__setattr__(self, "arc2face", teacher)

    def freeze_unet(self):
        """``unfreeze_model: False`` (yaml:26, ddpm.py:775-786)."""
        for p in self.model.parameters():
            p.requires_grad = False
        self.unfreeze_model = False

    @property
    def global_step(self):
        """Lightning's counter under manual optimisation: one per optimiser step, i.e. per ``manual_accumulate_grad_batches``
        micro-batches (ddpm.py:518-520)."""
        return self.batch_idx // self.manual_accumulate_grad_batches

    def on_save_checkpoint(self, checkpoint, embedding_manager=None, ckpt_dir=None, global_step=None):
        """ddpm.py:5392-5400 -- Lightning calls ``on_save_checkpoint(checkpoint)``: with a frozen UNet the checkpoint dict is
        emptied (nothing in it changed) and only the embedding manager's state is written -- by the embedding manager itself
        (``save(path)``, the reference's own class behind the boundary), as ``embeddings.pt`` and ``embeddings_gs-<step>.pt``
        in ``self.trainer.checkpoint_callback.dirpath``.  The three keywords override what is otherwise read from ``self`` /
        the attached trainer."""
        import os
        if not getattr(self, "unfreeze_model", any(p.requires_grad for p in self.model.parameters())):
            checkpoint.clear()
        embedding_manager = embedding_manager if embedding_manager is not None else getattr(self, "embedding_manager", None)
        if ckpt_dir is None:
            cb = getattr(getattr(self, "trainer", None), "checkpoint_callback", None)
            ckpt_dir = getattr(cb, "dirpath", None)
        if embedding_manager is not None and ckpt_dir is not None and os.path.isdir(ckpt_dir):
            global_step = self.global_step if global_step is None else global_step
            embedding_manager.save(os.path.join(ckpt_dir, "embeddings.pt"))
            embedding_manager.save(os.path.join(ckpt_dir, f"embeddings_gs-{global_step}.pt"))

    # ---- first stage ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_first_stage_moments(self, image_hwc, mask=None):
        return self.first_stage_model.encode_moments_nhwc(image_hwc, mask)

    @torch.no_grad()
    def encode_first_stage(self, x, mask=None):
        """reference signature (NCHW in, posterior out)."""
        return self.first_stage_model.encode(x, mask)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        """scale_factor * posterior.sample()  (ddpm.py:955-962)."""
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return self.scale_factor * encoder_posterior.sample(noise)
        if isinstance(encoder_posterior, torch.Tensor):
            return self.scale_factor * encoder_posterior
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """z NCHW latent -> image NCHW in [-1, 1] (ddpm.py:1260-1267 and the un-tiled tail)."""
        assert not predict_cids
        return self.first_stage_model.decode(z * (1.0 / self.scale_factor))

    def sample_latent_nhwc(self, moments, noise=None):
        """pixel-major fast path of the same: moments [B,h,w,2z] -> scale_factor * sample, [B,h,w,z]."""
        if noise is None:
            noise = torch.randn(moments.shape[:-1] + (moments.shape[-1] // 2,), device=moments.device)
        return ops.posterior_sample(moments, noise, self.scale_factor)

    @torch.no_grad()
    def get_input(self, batch, post_noise=None):
        """image [B,H,W,3] f32 in [-1,1] (+ fg_mask / aug_mask [B,H,W]) -> x_start NCHW [B,4,h,w] and the
        latent-resolution masks (ddpm.py:1178-1256)."""
        img = batch["image"]
        fg = batch.get("fg_mask")
        aug = batch.get("aug_mask")
        mask = None
        if fg is not None or aug is not None:
            mask = {"fg_mask": None if fg is None else fg[:, None].float(),
                    "aug_mask": None if aug is None else aug[:, None].float()}
        if post_noise is not None:
            post_noise = post_noise.permute(0, 2, 3, 1).contiguous()
        fs = self.first_stage_model
        if ops.TIMER is None and hasattr(fs, "encode_c_abi") and not os.environ.get("ADAP_VAE_PY"):
            # one C-ABI call (adap_vae_encode): the library issues the encoder's launches itself -- the same launches, the
            # same bits as the Python sequencing below, ~60 ctypes round trips less on the host (the per-kernel timers of
            # bench.py's roofline pass hook the Python wrappers, so that pass keeps the sequencing)
            if post_noise is None:
                f = 2 ** (fs.encoder.num_resolutions - 1)
                post_noise = torch.randn(img.shape[0], img.shape[1] // f, img.shape[2] // f, fs.quant_conv.out_channels // 2,
                                         device=img.device)
            _moments, z = fs.encode_c_abi(img, mask, post_noise, self.scale_factor)
            return z.permute(0, 3, 1, 2), mask
        moments = self.encode_first_stage_moments(img, mask)              # pixel-major [B,h,w,8]
        z = self.sample_latent_nhwc(moments, post_noise)                  # pixel-major [B,h,w,4]
        return z.permute(0, 3, 1, 2), mask

    # ---- denoising -------------------------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond):
        """cond = (c_static_emb, c_in, extra_info)  (ddpm.py:2192-2297, the un-tiled branch :2292)."""
        return self.model(x_noisy, t, c_crossattn=[cond])

    def guided_denoise(self, x_start, noise, t, cond, unet_has_grad=True, do_pixel_recon=False, cfg_info=None):
        """ddpm.py:2483-2532.  -> (model_output, x_noisy) on the recon path; with ``do_pixel_recon`` ->
        (model_output, x_recon): the unconditional prediction is computed without grad on the FIRST half of the
        batch only and repeated (the second half shares its initial conditions), combined by classifier-free
        guidance ``eps * s - eps_uncond * (s - 1)`` with per-instance scales ``cfg_info['cfg_scales']`` (or no
        guidance when None) and turned into x0 by ``predict_start_from_noise``."""
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        with torch.set_grad_enabled(unet_has_grad):
            model_output = self.apply_model(x_noisy, t, cond)
        if not do_pixel_recon:
            return model_output, x_noisy
        with torch.no_grad():
            x_start_, noise_, t_ = x_start.chunk(2)[0], noise.chunk(2)[0], t.chunk(2)[0]
            x_noisy_ = self.q_sample(x_start=x_start_.contiguous(), t=t_.contiguous(), noise=noise_.contiguous())
            model_output_uncond = self.apply_model(x_noisy_, t_, cfg_info["uncond_context"]).repeat(2, 1, 1, 1)
        if cfg_info.get("cfg_scales") is not None:
            cfg_scales = cfg_info["cfg_scales"].view(-1, 1, 1, 1)
            pred_noise = model_output * cfg_scales - model_output_uncond * (cfg_scales - 1)
        else:
            pred_noise = model_output
        x_recon = self.predict_start_from_noise(x_noisy, t=t, noise=pred_noise)
        return model_output, x_recon

    def calc_recon_loss(self, model_output, target, img_mask, fg_mask, fg_pixel_weight=1, bg_pixel_weight=1):
        """returns (loss, d loss / d model_output); both NCHW-shaped."""
        mo = model_output.permute(0, 2, 3, 1)
        tg = target.permute(0, 2, 3, 1)
        im = None if img_mask is None else img_mask.reshape(mo.shape[:-1])
        fg = None if fg_mask is None else fg_mask.reshape(mo.shape[:-1])
        loss, grad = ops.masked_mse(mo.detach(), tg, im, fg, fg_pixel_weight, bg_pixel_weight)
        return loss, grad.permute(0, 3, 1, 2)

    # ---- cross-layer consistency of the subject / background attention maps (ddpm.py:4259-4387) ---------------
    XLAYER_WEIGHTS = {8: 0.5, 12: 1., 16: 1., 17: 1., 18: 1., 19: 0.5, 20: 0.5, 21: 0.5, 22: 0.25, 23: 0.25, 24: 0.25}
    XLAYER_BELOW = {8: 7, 12: 8, 16: 12, 17: 16, 18: 17, 19: 18, 20: 19, 21: 20, 22: 21, 23: 22, 24: 23}

    def calc_fg_bg_xlayer_consist_loss(self, ca_attnscores, subj_indices, bg_indices, SSB_SIZE, token_maps=None):
        """ca_attnscores {layer_idx: attnscore [B, heads, N, 77]} as captured by the UNet (WITH gradient);
        subj_indices / bg_indices: (instance idx, token idx) of the subject / background embeddings in the prompts.
        Per aligned layer: the score map of the subject tokens (mean over heads, sum over the K embeddings) against the
        map of the layer below it (``XLAYER_BELOW``), the finer one bilinearly resized to the coarser, demeaned cosine
        with a sign-preserving squared reference, weighted by the normalised ``XLAYER_WEIGHTS``.  -> (fg, bg) losses."""
        from ...util import cosine_loss_rows, normalize_dict_values, normalized_sum, token_weight_matrix
        layer_w = normalize_dict_values(dict(LatentDiffusion.XLAYER_WEIGHTS))
        below = LatentDiffusion.XLAYER_BELOW

        first = next(iter(ca_attnscores.values()))
        idx_groups = [subj_indices] + ([bg_indices] if bg_indices is not None else [])
        # only the first SSB instances count (ddpm.py:4292-4301): every instance lists the same number of tokens
        w_full = token_weight_matrix(idx_groups, first.shape[0], first.shape[-1])       # [B, 77, groups]
        groups = idx_groups
        w_all = w_full[:SSB_SIZE, None]                                                 # [SSB, 1, 77, groups]
        sums = [[] for _ in groups]
        maps = {}
        # ``token_maps`` = ({layer: [B, heads, N, groups]}, weights): the per-head token maps the capture kernel emitted
        # for exactly these weights (UNetModel.forward) -- then the dense attnscore is not touched at all
        tm = token_maps[0] if (token_maps is not None and token_maps[1] is w_full) else None

        def token_map(layer, gi):
            """[SSB, N]: sum over the group's tokens of the head-mean score = score . w, w[b, m] = how often token m
            of instance b is listed.  The reference gathers the (instance, token) rows and reduces them; as ONE
            contraction per layer (both groups, and a layer serves two pairs) the forward reads the score tensor once,
            coalesced, and autograd's backward is an outer product that yields the dense d attnscore the HIP backward
            consumes -- the gather's backward is an index_put with accumulation over 77-strided slices of up to 40 MB."""
            if layer not in maps:
                if tm is not None:
                    m = tm[layer][:SSB_SIZE].mean(dim=1)                                            # [SSB, N, groups]
                else:
                    m = torch.matmul(ca_attnscores[layer][:SSB_SIZE], w_all).mean(dim=1)
                maps[layer] = m.permute(2, 0, 1)                                                    # [groups, SSB, N]
            return maps[layer] if gi is None else maps[layer][gi]

        for layer in ca_attnscores:
            if layer not in layer_w:
                continue
            fine, coarse = layer, below[layer]
            if ca_attnscores[coarse].shape[2] > ca_attnscores[fine].shape[2]:
                fine, coarse = coarse, fine
            hf, hc = int(np.sqrt(ca_attnscores[fine].shape[2])), int(np.sqrt(ca_attnscores[coarse].shape[2]))
            G = len(groups)
            # both groups' maps of all instances as one batch of G*SSB rows: one resize and one cosine per layer pair
            m_fine, m_coarse = token_map(fine, None), token_map(coarse, None)                       # [G, SSB, N]
            if hf != hc:            # (a same-size bilinear resize is the identity)
                m_fine = F.interpolate(m_fine.reshape(G * SSB_SIZE, 1, hf, hf), size=(hc, hc), mode="bilinear",
                                       align_corners=False).reshape(G, SSB_SIZE, hc * hc)
            per_instance = cosine_loss_rows(m_fine, m_coarse, exponent=2, do_demean_first=True, ref_grad_scale=1,
                                            aim_to_align=True)                                       # [G, SSB]
            per_group = per_instance.mean(dim=1) * layer_w[layer]          # = calc_ref_cosine_loss per group
            for gi in range(G):
                sums[gi].append(per_group[gi])
        loss_fg = normalized_sum(sums[0])
        loss_bg = normalized_sum(sums[1]) if len(groups) > 1 else normalized_sum([])
        return loss_fg, loss_bg

    # ---- subject / background attention maps should be complementary and respect the fg mask (ddpm.py:3932-4258) --
    COMPLEM_WEIGHTS = {7: 0.5, 8: 0.5, 12: 1., 16: 1., 17: 1., 18: 1., 19: 1., 20: 1., 21: 1., 22: 1., 23: 1., 24: 1.}

    def calc_fg_mb_suppress_loss(self, ca_attnscores, subj_indices, BLOCK_SIZE, fg_mask, instance_mask=None,
                                 token_maps=None):
        """ddpm.py:3932-4040: the subject-only case of ``calc_fg_bg_complementary_loss`` (its second return value)."""
        if subj_indices is None or len(subj_indices) == 0 or fg_mask is None:
            return 0
        return self.calc_fg_bg_complementary_loss(ca_attnscores, subj_indices, None, BLOCK_SIZE, fg_mask=fg_mask,
                                                  instance_mask=instance_mask, token_maps=token_maps)[1]

    def calc_fg_bg_complementary_loss(self, ca_attnscores, subj_indices, bg_indices, BLOCK_SIZE, fg_grad_scale=0.1,
                                      fg_mask=None, instance_mask=None, do_sqrt_norm=False, token_maps=None):
        """-> (fg_bg_complementary, subj_mb_suppress, bg_mf_suppress, fg_bg_mask_contrast), ddpm.py:4043-4258.
        Everything is a function of the per-head score maps of the subject tokens (sum over K_fg) and of the
        background tokens (sum over K_bg) -- the token maps the capture kernel emits (``token_maps``, see
        ``calc_fg_bg_xlayer_consist_loss``); without them they are contracted from the dense attnscore.

        The reference walks the 12 layers one by one, with two ``.any()`` host checks per layer (skip a layer whose
        resized mask has no foreground or no background pixel).  Here the layers of one resolution form one batch
        [layers, BLOCK, heads, N] and the skip is a 0/1 factor computed on the device: same sums, no sync, a third of
        the launches."""
        from ...util import (cosine_loss_rows, gen_gradient_scaler, normalize_dict_values, resize_mask_for_feat_or_attn,
                             token_weight_matrix)
        if subj_indices is None:
            return 0, 0, 0, 0
        have_bg = bg_indices is not None
        use_mask = fg_mask is not None and (instance_mask is None or bool(instance_mask.sum() > 0))
        if not have_bg and not use_mask:
            return 0, 0, 0, 0
        layer_w = normalize_dict_values(dict(LatentDiffusion.COMPLEM_WEIGHTS))
        layers = [li for li in ca_attnscores if li in layer_w]
        if not layers:
            return 0, 0, 0, 0
        first = ca_attnscores[layers[0]]
        idx_groups = [subj_indices] + ([bg_indices] if have_bg else [])
        w_full = token_weight_matrix(idx_groups, first.shape[0], first.shape[-1])           # [B, 77, groups]
        tm = token_maps[0] if (token_maps is not None and token_maps[1] is w_full) else None
        # tokens listed per instance (every instance lists the same number: ddpm.py:4081-4084)
        k_fg = float(subj_indices[0].numel()) / first.shape[0]
        k_bg = float(bg_indices[0].numel()) / first.shape[0] if have_bg else 1.0
        by_res = {}
        for li in layers:
            by_res.setdefault(ca_attnscores[li].shape[2], []).append(li)
        scaler = gen_gradient_scaler(0.5)           # protect the subject's activations on the foreground
        margin, margin_bg_at_mf = 0.4, 0.4 * k_fg / k_bg
        iw = None if instance_mask is None else instance_mask[:BLOCK_SIZE].view(1, BLOCK_SIZE, 1, 1).float()
        tot = [0, 0, 0, 0]
        for n_px, lis in by_res.items():
            if tm is not None:
                maps = torch.stack([tm[li][:BLOCK_SIZE] for li in lis])                     # [L, BLOCK, heads, N, groups]
            else:
                maps = torch.stack([torch.matmul(ca_attnscores[li][:BLOCK_SIZE], w_full[:BLOCK_SIZE, None]) for li in lis])
            if do_sqrt_norm:
                maps = maps / _const_tensor(tuple(float(np.sqrt(k)) for k in (k_fg, k_bg)[:maps.shape[-1]]), maps.device)
            lw = _const_tensor(tuple(layer_w[li] for li in lis), maps.device)
            subj = maps[..., 0]
            bg = maps[..., 1] if have_bg else None
            L, Bk, Hh, N = subj.shape
            if have_bg:
                # cosine_embedding(bg, subj*|subj|, -1): rows are (layer, instance, head); mean over instance and head
                per_row = cosine_loss_rows(bg, subj, exponent=2, do_demean_first=False, ref_grad_scale=fg_grad_scale,
                                           aim_to_align=False)                               # [L, BLOCK, heads]
                tot[0] = tot[0] + (per_row.mean(dim=(1, 2)) * lw).sum()
            if not use_mask:
                continue
            m = resize_mask_for_feat_or_attn(subj[0], fg_mask[:BLOCK_SIZE], "fg_mask", num_spatial_dims=1,
                                             mode="nearest|bilinear")
            fgm = (m.reshape(1, Bk, 1, N) > 1e-6).to(subj.dtype).expand(1, Bk, Hh, N)
            bgm = 1 - fgm
            # a layer whose mask has an empty foreground or background in ANY instance is skipped (ddpm.py:4150-4157)
            valid = ((fgm.sum(dim=(2, 3)) > 0).all() & (bgm.sum(dim=(2, 3)) > 0).all()).to(subj.dtype)

            def hinge(x):           # masked_mean(x, x > 0, instance_weights) per leading index: [..., BLOCK, heads, N]
                pos = (x > 0).to(x.dtype)
                xw = x if iw is None else x * iw
                return (xw * pos).sum(dim=(-3, -2, -1)) / pos.sum(dim=(-3, -2, -1)).clamp(min=1e-6)

            def mean_over(x, msk):  # masked_mean(x, msk, dim=(heads, N), keepdim) per (layer, instance)
                return (x * msk).sum(dim=(2, 3), keepdim=True) / msk.sum(dim=(2, 3), keepdim=True).clamp(min=1e-6)

            if maps.is_cuda and maps.dtype == torch.float32:
                # one fused HIP pass per direction instead of ~60 element-wise launches (functional.MaskHingesFn)
                from .... import functional as HF
                iw_v = None if instance_mask is None else instance_mask[:BLOCK_SIZE].float().contiguous()
                h4 = HF.MaskHingesFn.apply(maps, fgm[0, :, 0, :].contiguous(), iw_v, margin, margin_bg_at_mf, have_bg)
                h4 = (h4 * lw).sum(dim=1) * valid                                            # [4]
                tot[1] = tot[1] + h4[0] * 0.05
                if have_bg:
                    tot[2] = tot[2] + h4[1] * 0.1
                    tot[3] = tot[3] + (h4[2] + h4[3]) * 0.05
                continue
            subj_at_mf = scaler(subj * fgm)
            subj_at_mb = subj * bgm
            avg_subj_mf = mean_over(subj_at_mf, fgm)
            if not have_bg:
                tot[1] = tot[1] + (hinge(subj_at_mb + margin - avg_subj_mf) * lw).sum() * valid * 0.05
                continue
            bg_at_mf, bg_at_mb = bg * fgm, bg * bgm
            avg_bg_mb = mean_over(bg_at_mb, bgm)
            # the four hinge terms as one [4, L, BLOCK, heads, N] batch
            h4 = hinge(torch.stack([subj_at_mb + margin - avg_subj_mf, bg_at_mf + margin - avg_bg_mb,
                                    bg_at_mf + margin_bg_at_mf - avg_subj_mf, subj_at_mb + margin - avg_bg_mb]))
            h4 = (h4 * lw).sum(dim=1) * valid                                               # [4]
            tot[1] = tot[1] + h4[0] * 0.05
            tot[2] = tot[2] + h4[1] * 0.1
            tot[3] = tot[3] + (h4[2] + h4[3]) * 0.05
        return tuple(tot)

    # ---- Stage 2: losses of a compositional-distillation iteration (ddpm.py:3714-3930, 4389-4551) ----------------------
    MIX_LAYER_WEIGHTS = {7: 0.5, 8: 0.5, 12: 1., 16: 1., 17: 1., 18: 1., 19: 1., 20: 1., 21: 1., 22: 1., 23: 1., 24: 1.}
    FEAT_SIZE2POOLER = {8: (4, 2), 16: (4, 2), 32: (8, 4), 64: (8, 4)}

    def calc_prompt_mix_loss(self, ca_outfeats, ca_outfeat_lns, ca_attnscores, fg_indices_2b, BLOCK_SIZE):
        """The batch is four blocks: (subject single, subject comp, mix single, mix comp).  What composing does to the
        subject's features should be what it does to the class-mixed ("mix") features:
          * feat_delta_align: per layer, the pooled output features, weighted down where the subject attends, subject minus
            its projection on mix -- the comp delta against the single delta, orthogonal L2;
          * subj_attn_delta_align: the same for the subject tokens' score maps (cosine, exponent 3);
          * subj_attn_norm_distill: L1 between the mean subject scores of the subject and the mix instances.
        Mix-side gradients are scaled by 0.1 (features) / 0.05 (scores).  -> the three sums over layers (normalised weights)."""
        from .... import functional as HF
        from ...stage2 import calc_delta_alignment_loss, convert_attn_to_spatial_weight, double_token_indices, ortho_l2loss
        from ...util import gen_gradient_scaler, normalize_dict_values, normalized_sum, ortho_subtract
        w_layers = normalize_dict_values(dict(LatentDiffusion.MIX_LAYER_WEIGHTS))
        K_fg = len(fg_indices_2b[0]) // len(torch.unique(fg_indices_2b[0]))
        fg_indices_4b = double_token_indices(fg_indices_2b, BLOCK_SIZE * 2)
        feat_gs, attn_gs = gen_gradient_scaler(0.1), gen_gradient_scaler(0.05)
        l_attn_delta, l_feat_delta, l_attn_norm = [], [], []
        for li, ca_outfeat in ca_outfeats.items():
            if li not in w_layers:
                continue
            w = w_layers[li]
            if ca_outfeat_lns is not None:
                ca_outfeat = ca_outfeat_lns[str(li)](ca_outfeat.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            score = ca_attnscores[li].permute(0, 3, 1, 2)                                   # [4B, 77, heads, N]
            subj_attn_4b = score[fg_indices_4b].reshape(BLOCK_SIZE * 4, K_fg, *score.shape[2:]).sum(dim=1)
            ss_attn, sc_attn, ms_attn, mc_attn = subj_attn_4b.chunk(4)
            hw = ca_outfeat.shape[2:]
            fused = STAGE2_FUSED and subj_attn_4b.is_cuda and BLOCK_SIZE == 1
            if fused:                          # one launch each way for the two score-map terms (csrc/stage2loss.hip)
                d_align, d_norm = HF.SubjAttnTermsFn.apply(subj_attn_4b, 0.05)
                l_attn_delta.append(d_align * w)
                l_attn_norm.append(d_norm * w)
            else:
                mc_attn_gs, ms_attn_gs = attn_gs(mc_attn), attn_gs(ms_attn)
                d = calc_delta_alignment_loss(ss_attn, sc_attn, ms_attn, mc_attn, ref_grad_scale=0.05, feat_base_grad_scale=1,
                                              use_cosine_loss=True, cosine_exponent=3, delta_types=["feat_to_ref"])
                l_attn_delta.append(d["feat_to_ref"] * w)
                l_attn_norm.append(((sc_attn.mean(dim=-1) - mc_attn_gs.mean(dim=-1)).abs().mean()
                                    + (ss_attn.mean(dim=-1) - ms_attn_gs.mean(dim=-1)).abs().mean()) * w)
            if fused and subj_attn_4b.shape[-1] == hw[0] * hw[1]:        # the maps already have the features' resolution
                sub = subj_attn_4b.detach().float().contiguous()
                sw = ops.attn_spatial_weight(sub[3], sub[1], reversed=True).view(1, 1, *hw)
                feat = ca_outfeat * sw
            else:
                sw_mix, _ = convert_attn_to_spatial_weight(mc_attn, BLOCK_SIZE, hw, reversed=True)
                sw_subj, _ = convert_attn_to_spatial_weight(sc_attn, BLOCK_SIZE, hw, reversed=True)
                feat = ca_outfeat * ((sw_mix + sw_subj) / 2)
            k, st = LatentDiffusion.FEAT_SIZE2POOLER[feat.shape[-1]]
            feat_2d = F.avg_pool2d(feat, k, stride=st).reshape(feat.shape[0], -1)
            ss_f, sc_f, ms_f, mc_f = feat_2d.chunk(4)
            comp_delta = ortho_subtract(sc_f, feat_gs(mc_f))
            single_delta = ortho_subtract(ss_f, feat_gs(ms_f))
            l_feat_delta.append(ortho_l2loss(comp_delta, single_delta, mean=True) * w)
        return normalized_sum(l_feat_delta), normalized_sum(l_attn_delta), normalized_sum(l_attn_norm)

    def calc_comp_fg_bg_preserve_loss(self, ca_outfeats, ca_outfeat_lns, ca_qs, ca_q_bns, ca_attnscores, fg_mask,
                                      batch_have_fg_mask, subj_indices, BLOCK_SIZE):
        """Elastic matching between the comp and the single instances (``stage2.calc_elastic_matching_loss`` on the pooled
        queries / output features of every layer) + suppression of the subject tokens' scores on what the matching calls
        background.  -> (comp_single_map_align, sc_ss_fg_match, mc_ms_fg_match (= 0, disabled in the reference),
        sc_mc_bg_match, comp_subj_bg_attn_suppress, comp_mix_bg_attn_suppress)."""
        from .... import functional as HF
        from ...stage2 import calc_elastic_matching_loss
        from ...util import gen_gradient_scaler, masked_mean, normalize_dict_values, normalized_sum, resize_mask_for_feat_or_attn
        if fg_mask is None or batch_have_fg_mask.sum() == 0:
            return 0, 0, 0, 0, 0, 0
        w_layers = normalize_dict_values(dict(LatentDiffusion.MIX_LAYER_WEIGHTS))
        fg_mask_4b = fg_mask * batch_have_fg_mask.view(-1, 1, 1, 1)
        K_fg = len(subj_indices[0]) // len(torch.unique(subj_indices[0]))
        ib1, it1 = subj_indices[0][:BLOCK_SIZE * K_fg], subj_indices[1][:BLOCK_SIZE * K_fg]
        ind_B = torch.cat([ib1 + i * BLOCK_SIZE for i in range(4)], dim=0)
        ind_N = it1.repeat(4)
        mix_gs = gen_gradient_scaler(0.02)
        l_map, l_scss, l_scmc, l_sbg, l_mbg = [], [], [], [], []
        pooled_masks = {}               # per feature-map size: (pooled mask of the single instance, whether it has foreground)
        for li, ca_outfeat in ca_outfeats.items():
            if li not in w_layers:
                continue
            w = w_layers[li]
            q = ca_qs[li]                                                                    # [4B, heads, N, d]
            qh = int(np.sqrt(q.shape[2] * ca_outfeat.shape[2] // ca_outfeat.shape[3]))
            qw = q.shape[2] // qh
            q = q.permute(0, 1, 3, 2).reshape(q.shape[0], -1, qh, qw)
            if ca_q_bns is not None:
                q = ca_q_bns[str(li)](q)
            if ca_outfeat.shape[2:] != q.shape[2:]:
                ca_outfeat = F.interpolate(ca_outfeat, size=q.shape[2:], mode="bilinear", align_corners=False)
            if ca_outfeat_lns is not None:
                ca_outfeat = ca_outfeat_lns[str(li)](ca_outfeat.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            pool = (lambda x: F.avg_pool2d(x, 4, stride=2)) if ca_outfeat.shape[-1] > 8 else (lambda x: x)
            q_p = pool(q).reshape(*q.shape[:2], -1)
            f_p = pool(ca_outfeat).reshape(*ca_outfeat.shape[:2], -1)
            size = tuple(ca_outfeat.shape[2:])
            if size not in pooled_masks:            # the same mask at every layer of one resolution: resized and counted once
                m4 = resize_mask_for_feat_or_attn(ca_outfeat, fg_mask_4b, "fg_mask_4b", num_spatial_dims=2, mode="nearest|bilinear")
                m_p = pool(m4).chunk(4)[0]
                m_p = m_p.reshape(*m_p.shape[:2], -1)
                pooled_masks[size] = (m_p, bool((m_p != 0).any().item()))
            m_p, fg_any = pooled_masks[size]
            lm, lf, lb, sc_below, mc_below = calc_elastic_matching_loss(q_p, f_p, m_p, fg_bg_cutoff_prob=0.25,
                                                                        single_q_grad_scale=0.1, single_feat_grad_scale=0.01,
                                                                        mix_feat_grad_scale=0.05, fg_any=fg_any)
            l_map.append(lm * w)
            l_scss.append(lf * w)
            l_scmc.append(lb * w)
            if sc_below is None or mc_below is None:
                continue
            score = ca_attnscores[li].permute(0, 3, 1, 2)
            subj_attn = score[ind_B, ind_N].reshape(BLOCK_SIZE * 4, K_fg, *score.shape[2:]).sum(dim=1)
            H = int(np.sqrt(subj_attn.shape[-1]))
            a_hw = subj_attn.reshape(*subj_attn.shape[:2], H, H)
            if a_hw.shape[2:] != ca_outfeat.shape[2:]:
                a_hw = F.interpolate(a_hw, size=ca_outfeat.shape[2:], mode="bilinear", align_corners=False)
            a_p = pool(a_hw).reshape(*a_hw.shape[:2], -1)
            if STAGE2_FUSED and a_p.is_cuda and BLOCK_SIZE == 1:                # one launch each way (csrc/stage2loss.hip)
                ls, lx = HF.BgSuppressFn.apply(a_p, sc_below, mc_below, 0.02)
                l_sbg.append(ls * w)
                l_mbg.append(lx * w)
                continue
            _, sc_a, _, mc_a = a_p.chunk(4)
            l_sbg.append(masked_mean(sc_a.clamp(min=0), sc_below) * w)
            l_mbg.append(masked_mean(mix_gs(mc_a).clamp(min=0), mc_below) * w)
        return (normalized_sum(l_map), normalized_sum(l_scss), 0, normalized_sum(l_scmc), normalized_sum(l_sbg),
                normalized_sum(l_mbg))

    # ---- Stage 2: one compositional-distillation micro-batch (ddpm.py:2602-2839, 3041-3205, 3272-3445) --------------------
    def compos_distill_step(self, x_start, noise, cond, fg_mask=None, batch_have_fg_mask=None, t=None, clip_score_fn=None,
                            py_random=None, np_random=None, randn_like=None):
        """``is_compos_iter`` with ``do_mix_prompt_distillation``.  ``cond`` is what the conditioning assembly returns in
        'mix' mode: the four-way context (subject single, subject comp, class single, class comp) of ONE instance
        [4 * 16, 77, D], the four prompt lists, ``extra_info`` with ``placeholder2indices_1b`` / ``_2b``.

        Fresh iteration: t in the last fifth of the schedule; the latent is noise, or the training image's foreground
        shrunk onto noise (``comp_init_fg_from_training_image``); ``num_candidate_teachers`` (initial condition) candidates
        are denoised once WITHOUT grad under the subject-comp and under the mixed comp context (154-token split K/V,
        ``stage2.mix_static_vk_embeddings``) with guidance 5 / 6 against the empty prompt, decoded and scored by
        ``clip_score_fn(prompts, images) -> similarity`` (CLIP ViT-B/32 in the reference: a callable behind the boundary);
        if a candidate's mixed version beats its subject version (``stage2.select_teacher``) that candidate's initial
        condition is denoised again WITH grad under all four contexts, and the losses on the captured ``outfeat`` /
        ``attnscore`` / ``q`` of the 12 distillation layers are formed; its x0 prediction is cached for a later
        ``reuse_init_conds`` iteration of the same subject.  Reuse iteration: the cached x0 at a mid-schedule t, one pass.

        ``randn_like``: replaces ``torch.randn_like`` for the fresh latent (parity tests: the same draws on two devices).
        -> (loss or None when nothing was teachable, parts dict).  Backward: ``loss.backward()``."""
        import random as _random
        from ...stage2 import (calc_dyn_loss_scale, chunk_list, extend_indices_B_by_n_times, gen_cfg_scales_for_stu_tea,
                               init_x_with_fg_from_training_image, mix_static_vk_embeddings, select_teacher)
        from ...util import calc_prompt_emb_delta_loss, join_dict_of_indices_with_key_filter, repeat_selected_instances
        py_random = py_random or _random
        np_random = np_random or np.random
        fl = self.iter_flags
        assert fl["is_compos_iter"] and fl["do_mix_prompt_distillation"], "only the prompt-mix form of stage 2 is built"
        c_static_emb, c_in, extra_info = cond
        T, dev, BLOCK = self.num_timesteps, x_start.device, 1
        em = self.embedding_manager
        subj_keys = em.subject_string_dict if em is not None else extra_info.get("subject_strings", {"z": True})
        subj_1b = join_dict_of_indices_with_key_filter(extra_info["placeholder2indices_1b"], subj_keys)
        subj_2b = join_dict_of_indices_with_key_filter(extra_info["placeholder2indices_2b"], subj_keys)
        init_fg = bool(fl.get("comp_init_fg_from_training_image"))
        if self.do_zero_shot:
            k_range, v_range = [1.0, 0.8], ([1.0, 0.7] if init_fg else [1.0, 0.6])
        else:
            k_range, v_range = [1.0, 1.0], ([1.0, 0.85] if init_fg else [1.0, 0.7])
        have = batch_have_fg_mask if batch_have_fg_mask is not None else torch.zeros(x_start.shape[0], device=dev)
        fg_avail = float(have.float().mean()) if fg_mask is not None else 0.0
        img_mask, filtered_fg_mask = None, None
        mix = partial(mix_static_vk_embeddings, training_percent=self.training_percent, use_layerwise_embedding=True,
                      N_CA_LAYERS=self.N_CA_LAYERS, K_CLS_SCALE_LAYERWISE_RANGE=k_range, V_CLS_SCALE_LAYERWISE_RANGE=v_range)
        name = getattr(self, "batch_1st_subject_name", "zs_default")
        reuse = bool(fl.get("reuse_init_conds"))
        do_filter = bool(self.do_clip_teacher_filtering and not reuse)
        fl["do_teacher_filter"] = do_filter
        if reuse:
            cached = self.cached_inits.pop(name)
            x_start, prev_t = cached["x_start"], cached["t"]
            fg_mask, have, filtered_fg_mask = cached["fg_mask"], cached["batch_have_fg_mask"], cached["filtered_fg_mask"]
            init_fg = bool(cached["comp_init_fg_from_training_image"])
            if t is None:
                t_mid = torch.randint(int(T * 0.4), int(T * 0.7), (x_start.shape[0],), device=dev)
                t = torch.minimum(t_mid, prev_t - int(T * 0.15))
        else:
            if t is None:
                t = torch.randint(int(T * 0.8), T, (x_start.shape[0],), device=dev)
            if init_fg and fg_avail > 0:
                filtered_fg_mask = fg_mask.to(x_start.dtype) * have.view(-1, 1, 1, 1).to(x_start.dtype)
                x_start, fg_mask, filtered_fg_mask = init_x_with_fg_from_training_image(
                    x_start, fg_mask, filtered_fg_mask, self.training_percent, base_scale_range=(0.7, 1.0),
                    fg_noise_anneal_mean_range=(0.1, 0.4), np_random=np_random)
            else:
                x_start = (randn_like or torch.randn_like)(x_start)
        cond_orig = cond
        if do_filter:
            NT = self.num_candidate_teachers * BLOCK
            assert NT <= x_start.shape[0], f"num_candidate_teachers {NT} needs a batch of at least that size"
            x_start, noise, t = x_start[:NT].repeat(2, 1, 1, 1), noise[:NT].repeat(2, 1, 1, 1), t[:NT].repeat(2)
            _, subj_comp_emb, _, mix_comp_emb = c_static_emb.chunk(4)
            c_static_emb2 = torch.cat([subj_comp_emb[:BLOCK * self.N_CA_LAYERS].repeat(self.num_candidate_teachers, 1, 1),
                                       mix_comp_emb[:BLOCK * self.N_CA_LAYERS].repeat(self.num_candidate_teachers, 1, 1)], dim=0)
            _, subj_comp_prompts, _, cls_comp_prompts = chunk_list(list(c_in), 4)
            c_in2 = list(subj_comp_prompts) * self.num_candidate_teachers + list(cls_comp_prompts) * self.num_candidate_teachers
            cond = (c_static_emb2, c_in2, extra_info)
            fg_mask, filtered_fg_mask, have = repeat_selected_instances(slice(0, NT), 2, fg_mask, filtered_fg_mask, have)
            cfg_scales = gen_cfg_scales_for_stu_tea(6, 5, NT, dev)
            uncond = self.empty_context_tea_filter
        else:
            if not self.do_clip_teacher_filtering and not reuse:
                x_start = x_start[:BLOCK].repeat(4, 1, 1, 1)
            noise, t = noise[:BLOCK].repeat(4, 1, 1, 1), t[:BLOCK].repeat(4)
            fg_mask, filtered_fg_mask, have = repeat_selected_instances(slice(0, BLOCK), 4, fg_mask, filtered_fg_mask, have)
            cfg_scales = gen_cfg_scales_for_stu_tea(6, 5, BLOCK * 2, dev)
            uncond = self.empty_context_2b
        fg_avail = float(have.float().mean()) if fg_mask is not None else 0.0
        c_vk = mix(cond[0], subj_1b[1], t_frac=t.chunk(2)[0] / T)[0]
        ex = dict(extra_info)
        ex.update(iter_type=self.prompt_mix_scheme, img_mask=None, capture_distill_attn=not do_filter)
        model_output, x_recon = self.guided_denoise(x_start, noise, t, (c_vk, cond[1], ex), unet_has_grad=not do_filter,
                                                    do_pixel_recon=True, cfg_info={"cfg_scales": cfg_scales, "uncond_context": uncond})
        # ---- CLIP text-image score of the comp images: a metric to pick the teacher, never optimised (ddpm.py:3596-3712)
        cls_comp = list(extra_info["cls_comp_prompts"])
        if do_filter:
            code, prompts = x_recon, cls_comp * self.num_candidate_teachers * 2
        else:
            _, sc, _, mc = x_recon.chunk(4)
            code, prompts = torch.cat([sc, mc], dim=0), cls_comp * 2
        parts = {}
        with torch.no_grad():
            images = self.decode_first_stage(code.detach())
            losses_clip = 0.5 - clip_score_fn(prompts, images)
        parts["loss_clip_subj_comp"], parts["loss_clip_cls_comp"] = (v.mean() for v in losses_clip.chunk(2))
        if do_filter or reuse:
            teachable, best = select_teacher(losses_clip)
        else:
            teachable, best = torch.ones_like(losses_clip.chunk(2)[0], dtype=torch.bool), 0
        fl["is_teachable"] = bool(teachable.any())
        parts["best_cand_idx"] = best
        if do_filter and fl["is_teachable"]:
            del model_output, x_recon
            x_sel, noise_sel, t_sel = x_start[[best]].repeat(4, 1, 1, 1), noise[[best]].repeat(4, 1, 1, 1), t[best].repeat(4)
            c_vk = mix(cond_orig[0], subj_1b[1], t_frac=t_sel.chunk(2)[0] / T)[0]
            ex = dict(extra_info)
            ex.update(iter_type=self.prompt_mix_scheme, img_mask=None, capture_distill_attn=True,
                      placeholder2indices=extra_info["placeholder2indices_2b"])
            cfg_info = {"cfg_scales": gen_cfg_scales_for_stu_tea(6, 5, BLOCK * 2, dev), "uncond_context": self.empty_context_2b}
            model_output, x_recon = self.guided_denoise(x_sel, noise_sel, t_sel, (c_vk, cond_orig[1], ex), unet_has_grad=True,
                                                        do_pixel_recon=True, cfg_info=cfg_info)
            fg_mask, filtered_fg_mask, have = repeat_selected_instances([best], 4, fg_mask, filtered_fg_mask, have)
            fg_avail = float(have.float().mean()) if fg_mask is not None else 0.0
            self.cached_inits[name] = {
                "x_start": x_recon.detach().chunk(2)[0].repeat(2, 1, 1, 1), "delta_prompts": extra_info.get("delta_prompts"),
                "t": t_sel, "img_mask": None, "fg_mask": fg_mask, "batch_have_fg_mask": have, "filtered_fg_mask": filtered_fg_mask,
                "use_background_token": fl.get("use_background_token", False), "use_wds_comp": fl.get("use_wds_comp", False),
                "comp_init_fg_from_training_image": init_fg, "zs_clip_features": fl.get("zs_clip_features"),
                "zs_id_embs": fl.get("zs_id_embs"), "arc2face_prompt_emb": fl.get("arc2face_prompt_emb")}
            if len(self.cached_inits) > 100:
                del self.cached_inits[py_random.choice(list(self.cached_inits.keys()))]
        if not fl["is_teachable"]:
            return None, parts
        # ---- losses (ddpm.py:3207-3232, 3272-3445)
        loss = 0
        emb4 = ex.get("c_static_emb_4b")
        if fl.get("do_static_prompt_delta_reg") and emb4 is not None and self.prompt_emb_delta_reg_weight > 0:
            scale = (0.5 if self.optimizer_type == "Prodigy" else 1.0) / (5 if self.do_zero_shot else 1)
            l_delta = calc_prompt_emb_delta_loss(emb4, ex.get("prompt_emb_mask"))
            parts["static_prompt_delta"] = l_delta.detach()
            loss = loss + l_delta * (self.prompt_emb_delta_reg_weight * scale)
        acts = ex["ca_layers_activations"]
        lns = bns = None
        if self.normalize_ca_q_and_outfeat and em is not None and hasattr(em, "ca_q_bns"):
            bns, lns = em.ca_q_bns, em.ca_outfeat_lns
        l_preserve = 0
        if init_fg and fg_avail > 0 and self.comp_fg_bg_preserve_loss_weight > 0:
            l_map, l_ss, l_ms, l_bg, l_sbg, l_mbg = self.calc_comp_fg_bg_preserve_loss(
                acts["outfeat"], lns, acts["q"], bns, acts["attnscore"], filtered_fg_mask, have, subj_1b, BLOCK)
            bg_scale = calc_dyn_loss_scale(l_bg, 0.2, 2, min_scale_base_ratio=1, max_scale_base_ratio=3) if torch.is_tensor(l_bg) else 0
            l_preserve = l_map + (l_ss + l_ms * 0.1 + l_bg * bg_scale) + (l_sbg + l_mbg) * 0.02
            parts.update(comp_single_map_align=l_map, sc_ss_fg_match=l_ss, sc_mc_bg_match=l_bg, comp_subj_bg_attn_suppress=l_sbg,
                         comp_mix_bg_attn_suppress=l_mbg, comp_fg_bg_preserve=l_preserve)
        loss = loss + l_preserve * self.comp_fg_bg_preserve_loss_weight * (0.25 if reuse else 0.5)
        feat_scale = 0.5 if self.do_zero_shot else 2
        mix_lns = None
        if self.normalize_ca_q_and_outfeat:
            if py_random.random() < 0.5 and lns is not None:          # (the draw is made whether or not the norms exist)
                mix_lns, feat_scale = lns, feat_scale * 5
        l_feat, l_attn_delta, l_attn_norm = self.calc_prompt_mix_loss(acts["outfeat"], mix_lns, acts["attnscore"], subj_2b, BLOCK)
        if self.do_zero_shot:
            norm_scale = 1
        else:
            norm_scale = calc_dyn_loss_scale(l_attn_norm, 5., 0.2)
        l_mix = l_attn_delta * 0.1 + l_attn_norm * norm_scale + l_feat * feat_scale
        parts.update(feat_delta_align=l_feat, subj_attn_delta_align=l_attn_delta, subj_attn_norm_distill=l_attn_norm,
                     mix_prompt_distill=l_mix)
        preserve_on = torch.is_tensor(l_preserve) or l_preserve != 0
        loss = loss + l_mix * (0.5 if preserve_on else 1) * self.mix_prompt_distill_weight
        parts = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
        return loss, parts

    # ---- Arc2Face distillation: teacher rollout + multi-step student loss (ddpm.py:2950-3039) ------------------
    MAX_ACCUMU_BATCH_SIZE = 7

    def arc2face_distill_step(self, x_start, noise, t, cond, arc2face_prompt_emb, img_mask, fg_mask,
                              num_denoising_steps=1, relative_ts=None, noises=None, batched_student=True,
                              teacher_out=None):
        """The ``use_arc2face_as_target`` branch.  The teacher (``self.arc2face``) rolls ``num_denoising_steps`` out
        without grad; the student re-denoises the teacher's predictions and is regressed on the teacher's eps with
        bg_pixel_weight 0; the per-step losses are summed and divided by sqrt(ND).  The reference's indexing is kept
        literally: student step s starts from ``pred_x0s[s-1]``, i.e. for s = 0 from the LAST teacher prediction
        (ddpm.py:2978).

        ``batched_student``: the reference runs the student's (up to 7) passes one after the other on 1-2 instances
        (MAX_ACCUMU_BATCH_SIZE, "to avoid OOM").  Nothing couples the samples of a UNet batch (GroupNorm and attention
        are per sample), so here the passes are ONE forward/backward on the steps x instances batch -- the same
        numbers, on grids that fill the chip (288 GB of HBM hold the activations of seven passes easily).
        -> (loss, grads, model_outputs, aux): call ``torch.autograd.backward(model_outputs, grads)``; with
        ``batched_student`` both lists have one element and ``aux['model_outputs_per_step']`` holds the per-step views."""
        nd = int(num_denoising_steps)
        # ``teacher_out``: a rollout already computed for exactly these inputs (DistillPrefetcher, on a side stream)
        teacher = teacher_out if teacher_out is not None else self.arc2face(
            self, x_start, noise, t, arc2face_prompt_emb, num_denoising_steps=nd, relative_ts=relative_ts, noises=noises)
        noise_preds, pred_x0s, noises_, ts = teacher
        HB = x_start.shape[0]
        max_num_loss_steps = self.MAX_ACCUMU_BATCH_SIZE // HB
        loss_start_step = max(0, nd - max_num_loss_steps)
        targets = noise_preds[loss_start_step:]
        steps = list(range(loss_start_step, nd))
        c_emb, c_in, extra_info = cond
        extra_info = dict(extra_info)
        extra_info["img_mask"] = img_mask
        extra_info["capture_distill_attn"] = False                       # ddpm.py:2908
        inv = 1.0 / float(np.sqrt(nd))                                    # ddpm.py:3035
        if batched_student and len(steps) > 1:
            nS = len(steps)
            x0_all = torch.cat([pred_x0s[s - 1] for s in steps])
            nz_all = torch.cat([noises_[s] for s in steps])
            t_all = torch.cat([ts[s] for s in steps])
            L, M, D = c_emb.shape[0] // HB, c_emb.shape[1], c_emb.shape[2]
            c_all = c_emb.view(1, HB * L, M, D).expand(nS, HB * L, M, D).reshape(nS * HB * L, M, D)
            ex_all = dict(extra_info)
            if img_mask is not None:
                ex_all["img_mask"] = img_mask.repeat(nS, 1, 1, 1)
            mo_all, _ = self.guided_denoise(x0_all, nz_all, t_all, (c_all, c_in, ex_all))
            per_step = list(mo_all.chunk(nS))
            loss, gs, step_losses = 0.0, [], []
            for mo, tgt in zip(per_step, targets):
                l, g = self.calc_recon_loss(mo.contiguous(), tgt, img_mask, fg_mask, 1.0, 0.0)
                step_losses.append(l)
                loss = loss + l * inv
                gs.append(g * inv)
            return loss, [torch.cat(gs)], [mo_all], {"teacher": teacher, "step_losses": step_losses,
                                                    "loss_start_step": loss_start_step,
                                                    "model_outputs_per_step": per_step}
        model_outputs = []
        for s in steps:
            mo, _ = self.guided_denoise(pred_x0s[s - 1], noises_[s], ts[s], (c_emb, c_in, extra_info))
            model_outputs.append(mo)
        loss, grads, step_losses = 0.0, [], []
        for mo, tgt in zip(model_outputs, targets):
            l, g = self.calc_recon_loss(mo, tgt, img_mask, fg_mask, 1.0, 0.0)
            step_losses.append(l)
            loss = loss + l * inv
            grads.append(g * inv)
        return loss, grads, model_outputs, {"teacher": teacher, "step_losses": step_losses,
                                            "loss_start_step": loss_start_step,
                                            "model_outputs_per_step": model_outputs}

    @staticmethod
    def half_batch_size(batch_size, num_denoising_steps):
        """instances kept for a multi-step iteration: ``torch.arange(BS).chunk(ND)[0].shape[0]`` (ddpm.py:1857)."""
        if num_denoising_steps <= 1:
            return batch_size
        return torch.arange(batch_size).chunk(int(num_denoising_steps))[0].shape[0]

    @staticmethod
    def draw_num_denoising_steps(max_num_denoising_steps=7, np_random=np.random):
        """ddpm.py:1839-1850: ND in {1,3,5,7} with p = (.4,.3,.2,.1) renormalised over the allowed candidates."""
        cand = [s for s in (1, 3, 5, 7) if s <= max_num_denoising_steps]
        p = np.array([0.4, 0.3, 0.2, 0.1])[:len(cand)]
        return int(np_random.choice(cand, p=p / np.sum(p)))

    def shift_t_for_multistep(self, t, num_denoising_steps):
        """ddpm.py:2858-2861: pull t towards num_timesteps so the later rollout steps stay in a sensible range."""
        if num_denoising_steps > 1:
            return (4 * t + (num_denoising_steps - 1) * self.num_timesteps) // (3 + num_denoising_steps)
        return t

    # ---- one micro-batch of pure recon distillation ------------------------------------------------------------
    def shared_step(self, batch, **kwargs):
        """One micro-batch's forward half -> (loss, grad, model_output, aux); arguments: ``_shared_step_gen``.  That generator
        REQUESTS what other micro-batches of an accumulation window may want to share or to keep off their stream: the
        conditioning (``_CondRequest``: the hook / text encoder / embedding manager -- vendor kernels, which
        ``training_window`` keeps on lane 0) and the plain recon iteration's UNet pass (``x_start, noise, t, cond`` ->
        ``guided_denoise``; ``training_window(fuse=True)`` serves a window's requests with ONE batched pass).  Here both are
        served at once, in place."""
        gen = self._shared_step_gen(batch, **kwargs)
        req, res = self._resume(gen, None)
        if req is None:                                # an iteration that runs its own passes (distillation, stage 2)
            return res
        req, res = self._resume(gen, self.guided_denoise(*req))
        if req is not None:
            raise RuntimeError("shared_step: a second denoising request")
        return res

    @staticmethod
    def _resume(gen, value, serve_cond=None):
        """advance a ``_shared_step_gen`` to its next DENOISING request, serving the conditioning requests on the way
        (``serve_cond(request) -> cond``; default: in place, on the current stream).  -> (request, None), or (None, result)
        when the generator has finished."""
        try:
            req = gen.send(value)
            while isinstance(req, _CondRequest):
                req = gen.send(req.make() if serve_cond is None else serve_cond(req))
        except StopIteration as done:
            return None, done.value
        return req, None

    def _shared_step_gen(self, batch, t=None, noise=None, post_noise=None, cond=None, x_start=None,
                         num_denoising_steps=1, use_arc2face_as_target=False, relative_ts=None, noises=None,
                         trim_to_half_batch=True, batched_student=True, teacher_out=None, anneal_t=False):
        """``x_start``: a latent already encoded for this batch (e.g. by ``LatentPrefetcher`` on a side stream while
        the previous micro-batch's UNet pass was running); otherwise the batch is encoded here.

        ``anneal_t``: apply the recon iteration's timestep annealing to ``t`` (off by default so that callers who pass
        an explicit ``t`` -- the parity tests -- get exactly that ``t``).

        ``use_arc2face_as_target``: the distillation iteration (ddpm.py:1837-1876, 2950-3039): only the first
        HALF_BS instances are kept when ``num_denoising_steps`` > 1, ``batch["arc2face_prompt_emb"]`` [B,21,768] is
        the teacher's context; returns lists of model outputs / gradients.  ``trim_to_half_batch=False``: the caller
        has already cut the batch (and ``cond`` / ``relative_ts`` / ``noises``) down to HALF_BS, as the reference's
        ``forward`` does before ``p_losses`` runs."""
        if x_start is None:
            x_start, _mask = self.get_input(batch, post_noise)
        hw = x_start.shape[-2:]
        fg = batch.get("fg_mask")
        aug = batch.get("aug_mask")
        img_mask = None if aug is None else torch.nn.functional.interpolate(aug[:, None].float(), size=hw, mode="nearest")
        fg_mask = None if fg is None else torch.nn.functional.interpolate(fg[:, None].float(), size=hw, mode="nearest")
        instance_mask = batch.get("batch_have_fg_mask", batch.get("has_fg_mask"))
        do_static_delta = True
        if cond is None and self.cond_fn is None:
            # the reference's own conditioning side (yaml-instantiated text encoder + embedding manager): the front of its
            # ``shared_step`` and ``forward`` (ddpm.py:1436-2179), conditioning.ConditioningMixin
            if self.embedding_manager is None:
                raise RuntimeError("shared_step needs a context: pass cond=, set cond_fn, or construct the model with "
                                   "cond_stage_config + personalization_config")
            if self.iter_flags.get("is_compos_iter"):
                # stage 2: one subject, four prompt types, teacher filtering, losses on the captured activations
                x_start, img_mask, fg_mask, captions = self.prepare_compos_iteration(batch, x_start, img_mask, fg_mask)
                cond = self.assemble_conditioning(captions, x_start.shape[0])
                if noise is None:
                    noise = torch.randn_like(x_start)
                self.ensure_empty_contexts()
                loss, parts = self.compos_distill_step(x_start, noise, cond, fg_mask=self.iter_flags["fg_mask"],
                                                       batch_have_fg_mask=self.iter_flags["batch_have_fg_mask"], t=t,
                                                       clip_score_fn=self.clip_score_fn, **getattr(self, "_compos_test_hooks", {}))
                aux = {"compos": True, "reg_parts": parts, "x_start": x_start}
                if loss is None:                    # nothing teachable: no gradient this micro-batch (ddpm.py:3272)
                    return torch.zeros((), device=x_start.device), None, None, aux
                aux["reg_loss"] = loss
                return loss.detach(), None, None, aux
            x_start, img_mask, fg_mask, captions = self.prepare_recon_iteration(batch, x_start, img_mask, fg_mask)
            fl = self.iter_flags

            def recon_cond(captions=captions, bs=x_start.shape[0]):
                c = self.assemble_conditioning(captions, bs)
                c[2]["capture_distill_attn"] = True                     # ddpm.py:2866 (recon: not do_teacher_filter)
                self.attach_subject_indices(c[2])
                return c
            cond = yield _CondRequest(recon_cond)
            instance_mask = fl["batch_have_fg_mask"]
            use_arc2face_as_target = bool(fl["use_arc2face_as_target"])
            num_denoising_steps = int(fl["num_denoising_steps"])
            trim_to_half_batch = False                                  # prepare_recon_iteration has trimmed everything
            do_static_delta = bool(fl["do_static_prompt_delta_reg"])
            if use_arc2face_as_target:
                batch = dict(batch, arc2face_prompt_emb=fl["arc2face_prompt_emb"])
            t = None if t is None else t[:x_start.shape[0]]
            noise = None if noise is None else noise[:x_start.shape[0]]
        nd = int(num_denoising_steps)
        if use_arc2face_as_target and nd > 1 and trim_to_half_batch:
            hb = self.half_batch_size(x_start.shape[0], nd)
            x_start = x_start[:hb]
            batch = {k: (v[:hb] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in batch.items()}
            t = None if t is None else t[:hb]
            noise = None if noise is None else noise[:hb]
            img_mask = None if img_mask is None else img_mask[:hb]
            fg_mask = None if fg_mask is None else fg_mask[:hb]
        B = x_start.shape[0]
        if t is None:
            t = torch.randint(0, self.num_timesteps, (B,), device=x_start.device).long()
        if noise is None:
            noise = torch.randn_like(x_start)
        if cond is None:
            cond = yield _CondRequest(lambda b=batch: self.cond_fn(b))       # (the batch as trimmed above)
        if anneal_t and teacher_out is None:
            # every normal-recon iteration -- Arc2Face distillation included -- shifts t up by a random factor in [1, 1.3]
            # with an annealed probability BEFORE the multi-step shift (ddpm.py:2851-2866: the zero-shot and the default
            # branch use the same ranges); a prefetched rollout was made from a t that already went through both
            from ...util import probably_anneal_t
            t = probably_anneal_t(t, getattr(self, "training_percent", 0.0), self.num_timesteps, ratio_range=(1, 1.3),
                                  keep_prob_range=(0.4, 0.2))
        if use_arc2face_as_target:
            if teacher_out is None:
                t = self.shift_t_for_multistep(t, nd)          # (a prefetched rollout was made with the shifted t)
            loss, grads, outs, aux = self.arc2face_distill_step(
                x_start, noise, t, cond, batch["arc2face_prompt_emb"], img_mask, fg_mask, nd, relative_ts, noises,
                batched_student=batched_student, teacher_out=teacher_out)
            aux.update(x_start=x_start, t=t)
            return loss, grads, outs, aux
        c_emb, c_in, extra_info = cond
        extra_info = dict(extra_info)
        extra_info["img_mask"] = img_mask                                  # ddpm.py:2876
        if FUSED_REG_LOSSES and CAPTURE_TOKEN_MAPS_ONLY and extra_info.get("subj_indices") is not None:
            # the regularisers of this iteration read the distillation layers through their token maps (recon_regularizers)
            extra_info["capture_token_maps_only"] = True
        # the UNet pass: a request to whoever drives this generator (shared_step: guided_denoise at once; training_window: the
        # window's micro-batches in one batched pass, this one's captures sliced back into ITS extra_info)
        model_output, x_noisy = yield (x_start, noise, t, (c_emb, c_in, extra_info))
        loss, grad = self.calc_recon_loss(model_output, noise, img_mask, fg_mask, 1.0, self.bg_pixel_weight)
        aux = {"x_start": x_start, "x_noisy": x_noisy, "t": t, "extra_info": extra_info}
        reg, parts = self.recon_regularizers(extra_info, B, do_static_prompt_delta_reg=do_static_delta, fg_mask=fg_mask,
                                             instance_mask=instance_mask, do_complementary=True)
        if reg is not None:
            aux["reg_loss"], aux["reg_parts"] = reg, parts
            loss = loss + reg.detach()
        if "reg_tokmap_grads" in extra_info:
            aux["reg_tokmap_grads"] = extra_info.pop("reg_tokmap_grads")
        return loss, grad, model_output, aux

    def recon_regularizers(self, extra_info, block_size, do_static_prompt_delta_reg=True, fg_mask=None,
                           instance_mask=None, do_complementary=False):
        """The terms a ``do_normal_recon`` iteration adds to the masked MSE (ddpm.py:2921-2950 -> 3461-3500, and
        3207-3270), when the conditioning side supplies what they read:
          * (``do_complementary``: not in Arc2Face-distillation iterations, ddpm.py:2922) calc_fg_bg_complementary_loss
            on the captured attnscore with the latent-resolution ``fg_mask``: (0.2 (zero-shot) x complementary +
            subj_mb_suppress + bg_mf_suppress + fg_bg_mask_contrast) x fg_bg_complementary_loss_weight;
          * ``extra_info['c_static_emb_4b']`` (+ ``'prompt_emb_mask'``): calc_prompt_emb_delta_loss, weight
            prompt_emb_delta_reg_weight x 0.5 (Prodigy) / 5 (zero-shot); off in Arc2Face-distillation iterations (:572);
          * ``extra_info['subj_indices']`` (+ ``'bg_indices'``): the (instance, token) positions of the subject /
            background embeddings -- the reference joins them from ``placeholder2indices`` with the embedding
            manager's string sets (ddpm.py:2566-2569), which live behind the boundary -- and the captured
            ``attnscore`` of this forward: calc_fg_bg_xlayer_consist_loss x (0.2 fg / 0.06 bg when zero-shot, else
            1 / 0.3) x fg_bg_xlayer_consist_loss_weight.
        -> (differentiable scalar or None, {name: float tensor}).  Backward: ``manual_backward``."""
        from ...util import calc_prompt_emb_delta_loss
        total, parts = None, {}
        fused_roots = ([], [])
        extra_info.pop("reg_tokmap_grads", None)
        emb4 = extra_info.get("c_static_emb_4b")
        if do_static_prompt_delta_reg and emb4 is not None and self.prompt_emb_delta_reg_weight > 0:
            scale = 0.5 if self.optimizer_type == "Prodigy" else 1.0
            if self.do_zero_shot:
                scale /= 5
            pmask = extra_info.get("prompt_emb_mask")
            if (FUSED_REG_LOSSES and emb4.is_cuda and emb4.dtype == torch.float32 and emb4.dim() == 4 and emb4.is_contiguous()
                    and pmask is not None and pmask.dtype == torch.float32 and pmask.is_contiguous() and emb4.shape[-1] <= 1024):
                # value and gradient in one call of the HIP library (csrc/regloss.hip); the gradient enters autograd at emb4
                o2, d_emb4 = ops.prompt_delta_loss(emb4, pmask, self.prompt_emb_delta_reg_weight * scale)
                parts["static_prompt_delta"] = o2[0]
                total = o2[1]
                fused_roots = ([emb4], [d_emb4])
            else:
                l_delta = calc_prompt_emb_delta_loss(emb4, pmask)
                parts["static_prompt_delta"] = l_delta.detach()
                total = l_delta * (self.prompt_emb_delta_reg_weight * scale)
        subj = extra_info.get("subj_indices")
        acts = extra_info.get("ca_layers_activations")
        tm = None
        if acts is not None and acts.get("attnscore_tokmap") and extra_info.get("ca_tokmap_weights") is not None:
            tm = (acts["attnscore_tokmap"], extra_info["ca_tokmap_weights"])
        if fused_roots[0]:
            extra_info["reg_tokmap_grads"] = fused_roots
        if tm is not None and subj is not None and FUSED_REG_LOSSES:
            # both attention regularisers, values and gradients, as one call of the HIP library (csrc/regloss.hip): the host
            # expressions below -- the readable form, and what runs on CPU tensors -- cost ~400 small launches per micro-batch
            fused = self.fused_token_map_losses(acts, tm, subj, extra_info.get("bg_indices"), block_size, fg_mask, instance_mask,
                                                do_complementary)
            if fused is not None:
                f_total, f_parts, roots, grads = fused
                parts.update(f_parts)
                # manual_backward hands them to autograd as roots
                extra_info["reg_tokmap_grads"] = (fused_roots[0] + roots, fused_roots[1] + grads)
                total = f_total if total is None else total + f_total
                return total, parts
        if do_complementary and subj is not None and acts is not None and acts.get("attnscore") \
                and self.fg_bg_complementary_loss_weight > 0:
            l_c, l_smb, l_bmf, l_con = self.calc_fg_bg_complementary_loss(
                acts["attnscore"], subj, extra_info.get("bg_indices"), block_size, fg_grad_scale=0.1, fg_mask=fg_mask,
                instance_mask=instance_mask, do_sqrt_norm=False, token_maps=tm)
            term = (l_c * (0.2 if self.do_zero_shot else 1.0) + l_smb + l_bmf + l_con) * self.fg_bg_complementary_loss_weight
            for name, val in (("fg_bg_complem", l_c), ("subj_mb_suppress", l_smb), ("bg_mf_suppress", l_bmf),
                              ("fg_bg_mask_contrast", l_con)):
                parts[name] = val.detach() if torch.is_tensor(val) else val
            if torch.is_tensor(term):
                total = term if total is None else total + term
        if subj is not None and acts is not None and acts.get("attnscore") and self.fg_bg_xlayer_consist_loss_weight > 0:
            l_fg, l_bg = self.calc_fg_bg_xlayer_consist_loss(acts["attnscore"], subj, extra_info.get("bg_indices"),
                                                              block_size, token_maps=tm)
            fg_scale, bg_scale = (0.2, 0.06) if self.do_zero_shot else (1.0, 0.3)
            term = (l_fg * fg_scale + l_bg * bg_scale) * self.fg_bg_xlayer_consist_loss_weight
            parts["fg_xlayer_consist"] = l_fg.detach() if torch.is_tensor(l_fg) else l_fg
            parts["bg_xlayer_consist"] = l_bg.detach() if torch.is_tensor(l_bg) else l_bg
            total = term if total is None else total + term
        if not torch.is_tensor(total):          # nothing supplied, or only empty sums (no aligned layer captured)
            total = None
        return total, parts

    def fused_token_map_losses(self, acts, token_maps, subj_indices, bg_indices, block_size, fg_mask, instance_mask,
                               do_complementary):
        """``calc_fg_bg_complementary_loss`` + ``calc_fg_bg_xlayer_consist_loss`` with the weighting of ``recon_regularizers``
        on the capture kernel's token maps through ``ops.reg_losses`` -> (total, {name: value}, token maps, their gradients),
        or None when the fused form does not apply (then the host expressions run)."""
        from ...util import normalize_dict_values, token_weight_matrix
        maps, w_tok = token_maps
        scores = acts["attnscore"]
        layers = [li for li in scores if li in maps]
        if not layers or len(layers) > 16:
            return None
        first = maps[layers[0]]
        if not (first.is_cuda and first.dtype == torch.float32):
            return None
        have_bg = bg_indices is not None
        idx_groups = [subj_indices] + ([bg_indices] if have_bg else [])
        if token_weight_matrix(idx_groups, first.shape[0], scores[layers[0]].shape[-1]) is not w_tok or first.shape[-1] != len(idx_groups):
            return None
        w_c = self.fg_bg_complementary_loss_weight if do_complementary else 0
        w_x = self.fg_bg_xlayer_consist_loss_weight
        use_mask = w_c > 0 and fg_mask is not None and (instance_mask is None or bool(instance_mask.sum() > 0))
        if w_c > 0 and not have_bg and not use_mask:
            w_c = 0
        if not (w_c > 0 or w_x > 0):
            return None
        N = {li: maps[li].shape[2] for li in layers}
        pos = {li: i for i, li in enumerate(layers)}
        pairs = []
        if w_x > 0:
            xw, below = normalize_dict_values(dict(LatentDiffusion.XLAYER_WEIGHTS)), LatentDiffusion.XLAYER_BELOW
            for li in layers:
                if li not in xw:
                    continue
                fine, coarse = li, below[li]
                if coarse not in pos:
                    return None
                if N[coarse] > N[fine]:
                    fine, coarse = coarse, fine
                if N[fine] not in (N[coarse], 4 * N[coarse]):
                    return None
                pairs.append((pos[fine], pos[coarse], xw[li]))
        cw = [0.0] * len(layers)
        if w_c > 0:
            lw = normalize_dict_values(dict(LatentDiffusion.COMPLEM_WEIGHTS))
            cw = [lw.get(li, 0.0) for li in layers]
        k_fg = float(subj_indices[0].numel()) / first.shape[0]
        k_bg = float(bg_indices[0].numel()) / first.shape[0] if have_bg else 1.0
        fg = None
        if use_mask:
            fg = fg_mask[:, 0] if fg_mask.dim() == 4 else fg_mask
            fg = fg.float().contiguous()
            if fg.shape[1] != fg.shape[2] or any(fg.shape[1] % int(np.sqrt(n)) for n in N.values()):
                return None
        iw = None if (instance_mask is None or not use_mask) else instance_mask[:block_size].float().contiguous()
        fg_scale, bg_scale = (0.2, 0.06) if self.do_zero_shot else (1.0, 0.3)
        coefs = (fg_scale * w_x, bg_scale * w_x if have_bg else 0.0, (0.2 if self.do_zero_shot else 1.0) * w_c if have_bg else 0.0,
                 w_c if use_mask else 0.0, w_c if (use_mask and have_bg) else 0.0, w_c if (use_mask and have_bg) else 0.0)
        roots = [maps[li] for li in layers]
        p8, grads = ops.reg_losses(roots, cw, tuple(pairs), fg, iw, block_size, have_bg and w_c > 0, 0.4, 0.4 * k_fg / k_bg, 0.1,
                                   coefs)
        parts = {}
        if w_c > 0:
            parts.update(fg_bg_complem=p8[2], subj_mb_suppress=p8[3], bg_mf_suppress=p8[4], fg_bg_mask_contrast=p8[5])
        if w_x > 0:
            parts.update(fg_xlayer_consist=p8[0], bg_xlayer_consist=p8[1])
        return p8[6], parts, roots, grads

    @staticmethod
    def manual_backward(model_output, grad, aux=None):
        """``manual_backward(loss)`` (ddpm.py:595) for the pair shared_step returns: the masked-MSE gradient enters at
        ``model_output`` (computed by the loss kernel, not by autograd) and the regularisers' scalar at its own root, in
        ONE pass over the tape, so the UNet's backward runs once and receives the attnscore gradients on the way."""
        roots, grads = LatentDiffusion._backward_roots(model_output, grad, aux)
        if roots:
            from .... import functional as HF
            HF.prepare_tokmap_backward(roots, grads)       # all layers' token-map gradient prologues, three launches
            torch.autograd.backward(roots, grads)

    @staticmethod
    def _backward_roots(model_output, grad, aux=None):
        roots, grads = [], []
        if model_output is None:                    # a compositional iteration: the whole loss is the auxiliary scalar
            pass
        elif isinstance(model_output, (list, tuple)):
            for o, g in zip(model_output, grad):
                if o.requires_grad:
                    roots.append(o)
                    grads.append(g)
        elif model_output.requires_grad:
            roots.append(model_output)
            grads.append(grad)
        reg = None if aux is None else aux.get("reg_loss")
        if reg is not None and reg.requires_grad:
            roots.append(reg)
            grads.append(torch.ones_like(reg))
        tg = None if aux is None else aux.get("reg_tokmap_grads")
        if tg is not None:                          # the fused regularisers' gradients enter at the captured token maps
            for o, g in zip(*tg):
                if o.requires_grad:
                    roots.append(o)
                    grads.append(g)
        return roots, grads

    def ensure_empty_contexts(self):
        """ddpm.py:827-835 (``on_train_batch_start`` at global step 0): the empty prompt's context for the guidance passes of
        a compositional iteration -- for 2 instances and for ``num_candidate_teachers`` instances."""
        if getattr(self, "empty_context_2b", None) is None:
            self.empty_context_2b = self.get_learned_conditioning([""] * 2, embman_iter_type="empty")
        if getattr(self, "empty_context_tea_filter", None) is None:
            self.empty_context_tea_filter = self.get_learned_conditioning([""] * self.num_candidate_teachers,
                                                                          embman_iter_type="empty")

    def on_train_batch_start(self, batch, batch_idx, dataloader_idx=0):
        if self.embedding_manager is not None and self.cond_stage_model is not None:
            self.ensure_empty_contexts()

    def make_prefetcher(self):
        return LatentPrefetcher(self)

    def make_distill_prefetcher(self):
        return DistillPrefetcher(self)

    def configure_optimizers(self, optimized_parameters=None, max_steps=None, prodigy_config=None, weight_decay=None,
                             unfreeze_model=None, extra_model_parameters=()):
        """The reference's ``configure_optimizers`` (ddpm.py:5134-5345): the Prodigy branch with its three schedule types
        (the shipped config: 'Linear', v1-finetune-ada.yaml:59,74-84) and, through ``_configure_adam``, AdamW / NAdam.  ``optimized_parameters``: what ``EmbeddingManager.optimized_parameters()``
        returns -- a list of {'params', 'lr_ratio', 'excluded_from_prodigy'} (embedding_manager.py:2078-2095).  As in the
        reference, Prodigy gets ONE flat list (lr = 1) of the requires-grad parameters of every group that is not
        ``excluded_from_prodigy`` -- the groups' learning-rate ratios (and ``model_lr``) only matter to the Adam variants
        (``optimizer_type: AdamW | NAdam`` -> ``_configure_adam``); ``unfreeze_model`` appends the UNet's (and
        ``extra_model_parameters``', e.g. the text encoder's) parameters (ddpm.py:5176-5181).  -> Lightning's [{'optimizer', 'frequency', 'lr_scheduler': {...}}]."""
        from ...prodigy import Prodigy
        from ...util import prodigy_schedule
        # Lightning's no-argument call: everything comes from the model and its trainer, as in the reference
        if optimized_parameters is None:
            optimized_parameters = self.embedding_manager.optimized_parameters()
        if max_steps is None:
            max_steps = self.trainer.max_steps
        if prodigy_config is None:
            prodigy_config = getattr(self, "prodigy_config", None)
        if weight_decay is None:
            weight_decay = getattr(self, "weight_decay", 0.0)           # set on the model by main.py:1172
        if unfreeze_model is None:
            unfreeze_model = bool(getattr(self, "unfreeze_model", False))
            if unfreeze_model and not extra_model_parameters and self.cond_stage_model is not None:
                extra_model_parameters = list(self.cond_stage_model.parameters())      # ddpm.py:5179
        otype = getattr(self, "optimizer_type", "Prodigy")
        if otype in ("AdamW", "NAdam"):
            return LatentDiffusion._configure_adam(self, otype, optimized_parameters, max_steps, weight_decay, unfreeze_model,
                                        extra_model_parameters)
        if otype != "Prodigy":
            raise NotImplementedError(f"optimizer_type {otype!r}: 'Prodigy' (the shipped config), 'AdamW' and 'NAdam' are "
                                      "built; 'ProdigyAdamW' (a second AdamW over the same parameters in the last cycle, "
                                      "ddpm.py:5274-5302) is not -- the reference's own branch cannot run either: it hands "
                                      "OneCycleLR a float total_steps (ddpm.py:5298-5300), which torch refuses")
        cfg = {"zs_betas": (0.9, 0.999), "betas": (0.985, 0.993), "d_coef": 2.0, "warm_up_steps": 500, "scheduler_cycles": 1,
               "scheduler_type": "Linear"}
        cfg.update(dict(prodigy_config or {}))
        groups = [{"params": [q for q in g["params"] if q.requires_grad],
                   "excluded_from_prodigy": g.get("excluded_from_prodigy", False)} for g in optimized_parameters]
        if unfreeze_model:
            groups.append({"params": [q for q in list(extra_model_parameters) + list(self.model.parameters())],
                           "excluded_from_prodigy": False})
        params = [q for g in groups if not g["excluded_from_prodigy"] for q in g["params"]]
        opt = Prodigy(params, lr=1.0, weight_decay=weight_decay,
                      betas=tuple(cfg["zs_betas"] if self.do_zero_shot else cfg["betas"]), d_coef=cfg["d_coef"],
                      safeguard_warmup=cfg["scheduler_cycles"] > 1, use_bias_correction=True)
        sched = prodigy_schedule(opt, max_steps=max_steps, warm_up_steps=cfg["warm_up_steps"],
                                 scheduler_cycles=cfg["scheduler_cycles"], scheduler_type=cfg["scheduler_type"])
        return [{"optimizer": opt, "frequency": 1, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}]

    def _configure_adam(self, otype, optimized_parameters, max_steps, weight_decay, unfreeze_model, extra_model_parameters):
        """``optimizer_type: AdamW | NAdam`` (ddpm.py:5134-5142, 5157-5196): one parameter group per embedding-manager
        group at ``learning_rate * lr_ratio`` (requires-grad parameters only), with ``unfreeze_model`` one more at
        ``model_lr`` holding the text encoder's and the UNet's parameters; betas from ``adam_config``; the LR multiplier
        of ``adam_config.scheduler_config`` (yaml:65-72, ``max_decay_steps`` <- ``trainer.max_steps``) through a
        ``LambdaLR``.  The optimisers are the flat-buffer HIP ones of ``ldm.adam`` (same arguments and state keys as
        torch's)."""
        from ...adam import AdamW, NAdam
        from ...util import instantiate_from_config
        from torch.optim.lr_scheduler import LambdaLR
        lr = getattr(self, "learning_rate", None)
        if lr is None:
            raise AttributeError("configure_optimizers: `learning_rate` is not set on the model (main.py:1169-1172 sets "
                                 "model.learning_rate and model.weight_decay before trainer.fit)")
        acfg = self.adam_config
        if acfg is None or "scheduler_config" not in acfg or "target" not in acfg["scheduler_config"]:
            raise ValueError(f"optimizer_type {otype!r} needs adam_config.betas and adam_config.scheduler_config.target "
                             "(yaml:63-72)")
        groups = []
        for g in optimized_parameters:
            if len(g["params"]) > 0:
                groups.append({"params": [q for q in g["params"] if q.requires_grad], "lr": lr * g["lr_ratio"],
                               "excluded_from_prodigy": g.get("excluded_from_prodigy", False)})
        if unfreeze_model:
            groups.append({"params": list(extra_model_parameters) + list(self.model.parameters()), "lr": self.model_lr,
                           "excluded_from_prodigy": False})
        opt = (AdamW if otype == "AdamW" else NAdam)(groups, weight_decay=weight_decay, betas=tuple(acfg["betas"]))
        scfg = {"target": acfg["scheduler_config"]["target"], "params": dict(acfg["scheduler_config"].get("params", {}))}
        scfg["params"]["max_decay_steps"] = max_steps
        sched = LambdaLR(opt, lr_lambda=instantiate_from_config(scfg).schedule)
        return [{"optimizer": opt, "frequency": 1, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}]

    def draw_iteration_flags(self, global_step, composition_regs_iter_gap=0, arc2face_distill_iter_prob=0.0,
                             mix_prompt_distill_weight=0.0, np_random=np.random):
        """The iteration-type draw at the top of the reference's ``training_step`` (ddpm.py:516-572), consuming
        ``np.random`` in the same order: every ``composition_regs_iter_gap``-th global step is a compositional
        regularisation iteration (one ``np.random.choice`` over the candidate types); otherwise, with probability
        ``arc2face_distill_iter_prob`` (one ``np.random.rand``), a normal-recon iteration distils from the Arc2Face
        teacher, which also switches the static prompt-delta loss off.  -> the reference's ``iter_flags`` subset."""
        flags = {"do_normal_recon": True, "do_arc2face_distill": False, "is_compos_iter": False,
                 "do_mix_prompt_distillation": False, "do_ada_prompt_delta_reg": False, "calc_clip_loss": False,
                 "do_static_prompt_delta_reg": self.prompt_emb_delta_reg_weight > 0}
        cand = []
        if mix_prompt_distill_weight > 0:
            cand.append("do_mix_prompt_distillation")
        elif self.prompt_emb_delta_reg_weight > 0:
            cand.append("do_ada_prompt_delta_reg")
        if cand and composition_regs_iter_gap > 0 and global_step % composition_regs_iter_gap == 0:
            kind = cand[np_random.choice(len(cand), p=np.ones(len(cand)) / len(cand))]
            flags["do_mix_prompt_distillation"] = kind == "do_mix_prompt_distillation"
            flags.update(do_ada_prompt_delta_reg=True, is_compos_iter=True, calc_clip_loss=True, do_normal_recon=False)
        if flags["do_normal_recon"] and arc2face_distill_iter_prob > 0 and np_random.rand() < arc2face_distill_iter_prob:
            flags.update(do_arc2face_distill=True, do_static_prompt_delta_reg=False)
        return flags

    def training_step(self, batch, batch_idx=None, optimizer=None, reducer=None, scheduler=None, auto_iteration=None,
                      **step_kwargs):
        """manual optimisation (ddpm.py:515-638).  Lightning's call ``training_step(batch, batch_idx)`` works when a trainer
        is attached (``self.trainer`` with ``max_steps`` and the ``optimizer`` / ``scheduler`` / ``reducer`` that
        ``adaprompt_amd.trainer.Trainer.fit`` set up from ``configure_optimizers``): the iteration type is then drawn from
        the model's own yaml settings, exactly the reference's preamble.  The keyword form below is the same step with
        the trainer-side objects passed explicitly.  ``reducer`` (adaprompt_amd.parallel.GradReducer) all-reduces
        the trainable gradients after every micro-batch backward, as DDP does in the reference (no no_sync).

        ``auto_iteration``: a dict with the trainer-side settings of the reference's ``training_step`` preamble
        (ddpm.py:516-572, 1839-1859) -- ``max_steps``, ``composition_regs_iter_gap``, ``arc2face_distill_iter_prob``,
        ``max_num_denoising_steps`` -- then ``training_percent`` is updated and the iteration type is DRAWN as in the
        reference: an Arc2Face-distillation iteration (with its number of denoising steps), else a plain recon
        iteration with timestep annealing; with the yaml's own conditioning side attached (``embedding_manager``) the
        compositional iteration of stage 2 runs through ``shared_step`` -> ``compos_distill_step`` (ldm/stage2.py)."""
        tr = getattr(self, "trainer", None)
        if optimizer is None and tr is not None and batch_idx is not None:
            optimizer, scheduler, reducer = tr.optimizer, tr.scheduler, tr.reducer
            if auto_iteration is None:
                auto_iteration = {"max_steps": tr.max_steps, "composition_regs_iter_gap": self.composition_regs_iter_gap,
                                  "arc2face_distill_iter_prob": self.arc2face_distill_iter_prob,
                                  "mix_prompt_distill_weight": self.mix_prompt_distill_weight,
                                  "max_num_denoising_steps": self.max_num_denoising_steps}
            if getattr(tr, "lanes", None) is not None and self.manual_accumulate_grad_batches > 1 and not step_kwargs:
                return self._training_step_in_windows(batch, batch_idx, tr, auto_iteration)
        if auto_iteration is not None:
            self._iteration_preamble(auto_iteration, step_kwargs)
        loss, grad, model_output, aux = self.shared_step(batch, **step_kwargs)
        self._micro_batch_backward(model_output, grad, aux, reducer)
        self.batch_idx += 1
        if optimizer is not None and self.batch_idx % self.manual_accumulate_grad_batches == 0:
            self._optimizer_step(optimizer, reducer, scheduler)
        return loss, aux

    # ---- Lightning's per-batch call on micro-batch lanes ---------------------------------------------------------------------
    def _training_step_in_windows(self, batch, batch_idx, tr, auto_iteration):
        """``training_step(batch, batch_idx)`` as Lightning calls it (ddpm.py:515), once per micro-batch, with a trainer that
        runs accumulation windows on lanes (``adaprompt_amd.trainer.Trainer(micro_batch_lanes=True)``): the micro-batches of a
        window are BUFFERED and the window runs -- ``training_window`` on ``tr.lanes``, F0 F1 B0 B1, step -- when its last one
        arrives.  The call that completes a window returns that micro-batch's ``(loss, aux)`` with ``aux['window']`` =
        [(loss, aux), ...] of all of them; the others return ``(None, {'deferred': True})``: their loss is deferred, nothing
        else changes (iteration flags and RNG are drawn per micro-batch, in order, when the window runs).
        ``tr.prefetch_windows`` = L > 0 additionally keeps L windows of batches buffered AHEAD of the one that runs, and their
        VAE encodes go to the prefetch stream behind the running window's backwards (``LatentPrefetcher``): the encodes then
        fill the window's tail and the optimiser step, where one lane alone leaves the chip idle.  The latent's posterior
        noise is then drawn at submission time, i.e. earlier in the device RNG sequence than the sequential loop draws it.
        ``flush_window()`` runs what is still buffered (an epoch's end)."""
        n = int(self.manual_accumulate_grad_batches)
        look = max(0, int(getattr(tr, "prefetch_windows", 0) or 0))
        st = self.__dict__.setdefault("_win_entry", {"buf": [], "submitted": 0, "pf": None})
        if not st["buf"] and self.batch_idx % n != 0:
            # (the open window was started outside this entry: finish it on one stream first)
            return self.training_step(batch, optimizer=tr.optimizer, reducer=tr.reducer, scheduler=tr.scheduler,
                                      auto_iteration=auto_iteration)
        st["buf"].append(batch)
        if len(st["buf"]) < n * (look + 1):
            return None, {"deferred": True}
        out = self._run_buffered_window(tr, auto_iteration, look)
        loss, aux = out[-1]
        aux = dict(aux or {})
        aux["window"] = out
        return loss, aux

    def _run_buffered_window(self, tr, auto_iteration, look):
        n = int(self.manual_accumulate_grad_batches)
        st = self._win_entry
        buf = st["buf"]
        window = buf[:n]
        kwargs, after_backward = None, None
        if look > 0:
            if st["pf"] is None:
                st["pf"] = self.make_prefetcher()
            pf = st["pf"]
            while st["submitted"] < min(len(buf), n * look):           # (the first windows: nothing is in flight yet)
                pf.submit(buf[st["submitted"]])
                st["submitted"] += 1

            def kwargs(k):
                return {"x_start": pf.get()}

            def after_backward(k):
                j = n * look + k
                if j < len(buf) and st["submitted"] <= j:
                    pf.submit(buf[j])
                    st["submitted"] = j + 1
        out = self.training_window(window, tr.optimizer, tr.reducer, tr.scheduler, tr.lanes, auto_iteration=auto_iteration,
                                   step_kwargs=kwargs, after_backward=after_backward)
        del buf[:n]
        st["submitted"] = max(0, st["submitted"] - n)
        return out

    def flush_window(self, run=True):
        """what ``training_step(batch, batch_idx)`` still holds buffered: whole windows run as windows, a partial one through
        the one-stream step (``run=False``: drop it, e.g. ``max_steps`` reached).  -> [(loss, aux), ...]"""
        st = self.__dict__.get("_win_entry")
        tr = getattr(self, "trainer", None)
        out = []
        if not st or not st["buf"]:
            return out
        n = int(self.manual_accumulate_grad_batches)
        auto = {"max_steps": tr.max_steps, "composition_regs_iter_gap": self.composition_regs_iter_gap,
                "arc2face_distill_iter_prob": self.arc2face_distill_iter_prob,
                "mix_prompt_distill_weight": self.mix_prompt_distill_weight,
                "max_num_denoising_steps": self.max_num_denoising_steps}
        look = max(0, int(getattr(tr, "prefetch_windows", 0) or 0))
        while run and len(st["buf"]) >= n and self.batch_idx % n == 0:
            out += self._run_buffered_window(tr, auto, look)
        pf = st["pf"]
        for b in list(st["buf"]):
            kw = {}
            if st["submitted"] > 0:
                kw["x_start"] = pf.get()                    # (drained either way: the queue must not leak into a later epoch)
                st["submitted"] -= 1
            if run:
                out.append(self.training_step(b, optimizer=tr.optimizer, reducer=tr.reducer, scheduler=tr.scheduler,
                                              auto_iteration=auto, **kw))
        st["buf"].clear()
        return out

    def _iteration_preamble(self, auto_iteration, step_kwargs):
        """the reference's ``training_step`` preamble (ddpm.py:516-572, 1839-1859): training_percent, the iteration type."""
        cfg = auto_iteration
        gstep = self.batch_idx // self.manual_accumulate_grad_batches            # Lightning's global_step
        self.training_percent = min(1.0, gstep * self.manual_accumulate_grad_batches / max(1, cfg.get("max_steps", 1)))
        flags = self.draw_iteration_flags(gstep, cfg.get("composition_regs_iter_gap", 0),
                                          cfg.get("arc2face_distill_iter_prob", 0.0),
                                          cfg.get("mix_prompt_distill_weight", 0.0))
        if flags["is_compos_iter"] and not (flags["do_mix_prompt_distillation"] and self.embedding_manager is not None
                                            and getattr(self, "clip_score_fn", None) is not None):
            raise NotImplementedError("a compositional iteration needs mix_prompt_distill_weight > 0, the reference's "
                                      "conditioning side (cond_stage_config + personalization_config) and a clip_score_fn; "
                                      "the ada-delta-only ablation form is not built")
        if flags["do_arc2face_distill"] and (self.cond_fn is not None or "cond" in step_kwargs):
            # (with the reference's own conditioning side, shared_step's front decides use_arc2face_as_target / ND)
            step_kwargs.update(use_arc2face_as_target=True, num_denoising_steps=self.draw_num_denoising_steps(
                cfg.get("max_num_denoising_steps", 7)))
        step_kwargs.setdefault("anneal_t", True)       # every normal-recon iteration, distillation included (:2851-2861)
        self.init_iteration_flags()
        self.iter_flags.update(flags)

    def _micro_batch_backward(self, model_output, grad, aux, reducer, lanes=None):
        if reducer is not None and lanes is None:
            # the previous micro-batch's all-reduce overlapped the forward above; it must have landed before this backward
            # adds into the same flat gradient buffer (AccumulateGrad and the weight-gradient kernels write it from the
            # very start of the backward)
            reducer.wait()
            reducer.begin_backward()         # chunks of the gradient buffer go out as the backward completes them
        # (with lanes the wait sits in MicroBatchLanes' gate, in front of the accumulation into the shared buffer only, and
        # the exchange is single-shot: a lane's backward starts while the previous lane's collective may still be in flight)
        self.manual_backward(model_output, grad, aux)                      # == manual_backward(loss), ddpm.py:595
        if reducer is not None:
            reducer.reduce()

    def _optimizer_step(self, optimizer, reducer, scheduler):
        """clip 0.5 -> step -> zero_grad -> scheduler.step, every ``manual_accumulate_grad_batches``-th micro-batch
        (ddpm.py:606-633)."""
        if reducer is not None:
            reducer.wait()
        from ...flatopt import FlatParams
        # sync-free: raises if a single-launch GroupNorm's exchange timed out since the previous optimiser step (its
        # outputs were NaN then, so the loss already is; this names the cause)
        ops.gn_poison_poll()
        if isinstance(optimizer, FlatParams):              # Prodigy / AdamW / NAdam: clip fused into the flat-buffer step
            optimizer.step(clip_norm=self.grad_clip if self.grad_clip else None)
        else:
            params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            if self.grad_clip and params:
                torch.nn.utils.clip_grad_norm_(params, self.grad_clip)
            optimizer.step()
        if reducer is not None:
            reducer.zero()
        else:
            optimizer.zero_grad(set_to_none=False)
        if scheduler is not None:
            scheduler.step()                               # ddpm.py:629-633

    def training_window(self, batches, optimizer, reducer=None, scheduler=None, lanes=None, auto_iteration=None,
                        step_kwargs=None, after_forward=None, after_backward=None, fuse=None):
        """The micro-batches of ONE accumulation window (``manual_accumulate_grad_batches`` of them: ddpm.py:591-633
        accumulates their gradients and steps once) issued forward-first -- F0 F1 .. B0 B1 .. step -- and, with ``lanes``
        (``MicroBatchLanes``), each on its own HIP stream.  All of a window's micro-batches read the SAME weights, so they
        are independent until their gradients meet in the shared buffer; on an MI355X one micro-batch leaves most CUs idle
        most of the time (its launches are small and dependent), and a second one beside it fills them.  The host order of
        the forward halves is the reference's (iteration flags and host RNG are drawn per micro-batch, in order; a backward
        draws nothing), the accumulation order into ``.grad`` is micro-batch 0, 1, .. (the lanes' gate), so the numbers are
        those of ``training_step`` called on the batches one after the other -- up to the summation order of the GroupNorm
        statistics on the second lane (two-pass kernels there: the single-launch exchange belongs to one stream per device).
        ``step_kwargs``: one dict per micro-batch, one for all, or a callable ``k -> dict`` evaluated on the micro-batch's
        lane right before its forward (RNG draws, a prefetched latent); ``after_forward(k)`` / ``after_backward(k)``: called on
        the lane once its forward / backward is issued (e.g. to submit the next VAE encode to a ``LatentPrefetcher``).
        ``fuse``: a window whose micro-batches are all plain recon iterations goes through the UNet as ONE batched pass
        (``_training_window_fused``: the same per-micro-batch fronts, losses and gradients; one forward, one backward at n times
        the batch).  Default (None): when the window cannot run on lanes -- no ``lanes`` given, or the conditioning side runs
        inside ``shared_step`` -- since a batch of 8 on one stream beats two batches of 4 one after the other (166 vs 130 images/s);
        with lanes available the lanes win (the hook's forward / backward and the losses of the two micro-batches then overlap the
        other lane instead of queueing behind one stream: 24.9 vs 26.3 ms per micro-batch), ``ADAP_WINDOW_FUSE=1`` or
        ``fuse=True`` forces it.  Windows that start with a distillation / compositional micro-batch always take the lanes path.
        -> [(loss, aux), ...]"""
        n = len(batches)
        assert n == self.manual_accumulate_grad_batches and self.batch_idx % n == 0, \
            "training_window: one whole accumulation window, starting on a window border"
        if lanes is not None and any(p.requires_grad for p in self.model.parameters()):
            lanes = None               # the UNet's own gradients are written through raw pointers during the whole backward
        if fuse is None:
            env = os.environ.get("ADAP_WINDOW_FUSE")
            fuse = (lanes is None and torch.cuda.is_available()) if env is None else env != "0"
        pre_kws = None
        if fuse and n >= 2:       # (also with ``unfreeze_model``: one backward at n times the batch = one weight-gradient pass per window)
            done = self._training_window_fused(batches, optimizer, reducer, scheduler, auto_iteration, step_kwargs, after_forward,
                                               after_backward)
            if not isinstance(done, dict):
                return done
            pre_kws = done                 # {0: kwargs of micro-batch 0}: it is not a plain recon iteration -- the lanes take the window
        import contextlib
        from .... import functional as HF
        side_lane_was, ksplit_was, gn_owner_was = HF.SIDE_LANE, None, None
        if lanes is not None:
            lanes.window_start()       # the weights (and whatever else lane 0 has queued so far) as the side lanes' starting point
            # inside a window on lanes, cached device data (weight packs ...) is complete before the other lane can read it
            # (ops.note_cache_fill).  Scoped to the window: outside it lane 0 orders everything (window_start), and the
            # one-stream paths of the same process (unfreeze_model rebuilds 686 packs per step) must not pay a drain per pack
            ops.multi_stream(+1)
            # a block's side lane (functional.side_lane: work off its dependency chain on a second stream) fills CUs the chain
            # leaves idle -- which the other micro-batch's lane does here; with two lanes, their two side lanes and the prefetch
            # stream the five streams outnumber the hardware queues, and the blocks' fork / join events then cost more than the
            # lanes buy (measured: 26.9 vs 26.4 ms per micro-batch; with a hardware queue per stream 34 ms)
            HF.SIDE_LANE = HF.SIDE_LANE and os.environ.get("ADAP_SIDE_LANE_WITH_LANES", "0") == "1"
            # split K less while two lanes keep the chip busy: the other lane fills the CUs a short grid leaves idle, and the slab
            # traffic + reduce launches go away (35 % of the lone-stream target: -0.2 ms per micro-batch; alone it costs +0.8 ms)
            if "ADAP_KSPLIT_SCALE" not in os.environ:
                ksplit_was = ops._lib.call_long("adap_conv_ksplit_scale", int(os.environ.get("ADAP_LANES_KSPLIT_SCALE", "35")))
            gn_lane = int(os.environ.get("ADAP_GN_OWNER_LANE", "0"))
            if gn_lane and gn_lane < len(lanes.streams):
                gn_owner_was = (lanes.main.device, ops.gn_single_launch_stream(lanes.main.device))
                ops.set_gn_single_launch_stream(lanes.main.device, lanes.streams[gn_lane].cuda_stream)
        try:
            return self._training_window(batches, optimizer, reducer, scheduler, lanes, auto_iteration, step_kwargs, after_forward,
                                         contextlib, after_backward, pre_kws)
        finally:
            if lanes is not None:
                ops.multi_stream(-1)
            HF.SIDE_LANE = side_lane_was
            if ksplit_was is not None:
                ops._lib.call_long("adap_conv_ksplit_scale", ksplit_was)
            if gn_owner_was is not None:
                ops.set_gn_single_launch_stream(*gn_owner_was)

    def _window_kwargs(self, k, step_kwargs, auto_iteration, cond=None):
        """micro-batch k's ``shared_step`` arguments: the caller's (a callable is evaluated NOW: RNG draws, a prefetched latent),
        then the iteration-type draw -- once per micro-batch, in order, as in the sequential loop."""
        kw = dict(step_kwargs(k) if callable(step_kwargs) else
                  step_kwargs[k] if isinstance(step_kwargs, (list, tuple)) else (step_kwargs or {}))
        if cond is not None and kw.get("cond") is None:
            kw["cond"] = cond
        if auto_iteration is not None:
            self._iteration_preamble(auto_iteration, kw)
        return kw

    @staticmethod
    def _is_plain_recon(kw):
        return (not kw.get("use_arc2face_as_target") and int(kw.get("num_denoising_steps", 1) or 1) == 1
                and kw.get("teacher_out") is None)

    _FUSE_FLAGS = ("use_layerwise_context", "use_conv_attn_kernel_size", "iter_type", "is_training", "capture_distill_attn",
                   "capture_token_maps_only", "debug_attn")
    _FUSE_MERGED = ("img_mask", "subj_indices", "bg_indices", "placeholder2indices")     # concatenated / shifted onto the merged batch
    # read by the micro-batch's OWN losses after the pass (never by the UNet), or written by the pass and sliced back
    _FUSE_PER_REQUEST = ("c_static_emb_4b", "prompt_emb_mask", "ca_layers_activations", "ca_tokmap_weights", "reg_tokmap_grads")

    def _denoise_fusable(self, reqs):
        """can these ``guided_denoise`` requests (x_start, noise, t, (c_emb, c_in, extra_info)) go through the UNet as one batch?"""
        x0, c0, e0 = reqs[0][0], reqs[0][3][0], reqs[0][3][2]
        for x, _n, _t, (c, _ci, ei) in reqs:
            if not (torch.is_tensor(c) and torch.is_tensor(c0)) or c.shape[1:] != c0.shape[1:] or c.dtype != c0.dtype:
                return False
            if x.shape[1:] != x0.shape[1:] or ei.get("placeholder2indices") is not None:
                return False
            if any(ei.get(f) != e0.get(f) for f in self._FUSE_FLAGS):
                return False
            if any((ei.get(f) is None) != (e0.get(f) is None) for f in ("img_mask", "subj_indices", "bg_indices")):
                return False
            # whatever else the conditioning side put there must be the SAME for all requests (it is carried into the merged
            # pass as is): a key that differs -- or that only some requests have -- would be dropped or applied to the wrong rows
            for f in set(ei) | set(e0):
                if f in self._FUSE_FLAGS or f in self._FUSE_MERGED or f in self._FUSE_PER_REQUEST:
                    continue
                a, b = ei.get(f), e0.get(f)
                if a is b:
                    continue
                if torch.is_tensor(a) or torch.is_tensor(b) or isinstance(a, (dict, list, tuple)) or isinstance(b, (dict, list, tuple)):
                    return False                   # (not the same object: a per-request payload this pass does not know how to merge)
                if a != b:
                    return False
        return True

    def _denoise_fused(self, reqs):
        """the requests' UNet passes as ONE: inputs concatenated along the batch dim (a layerwise context keeps an instance's 16
        layers together, so the contexts concatenate too), the (instance, token) index sets shifted onto the merged batch, and what
        the UNet captured sliced back into each request's own ``extra_info`` (views: their gradients meet in the one tape).
        -> [(model_output, x_noisy), ...] per request."""
        sizes = [r[0].shape[0] for r in reqs]
        offs = [sum(sizes[:k]) for k in range(len(sizes))]
        eis = [r[3][2] for r in reqs]
        ei = {f: v for f, v in eis[0].items() if f not in self._FUSE_MERGED and f not in self._FUSE_PER_REQUEST}
        ei["placeholder2indices"] = None           # (flags and the common remainder, equal for all requests: _denoise_fusable)
        if eis[0].get("img_mask") is not None:
            ei["img_mask"] = torch.cat([e["img_mask"] for e in eis])
        for f in ("subj_indices", "bg_indices"):
            if eis[0].get(f) is not None:
                ei[f] = self._merged_indices([e[f] for e in eis], offs)
        c_ins = [r[3][1] for r in reqs]
        c_in = sum((list(c) for c in c_ins), []) if all(isinstance(c, (list, tuple)) for c in c_ins) else c_ins[0]
        out, x_noisy = self.guided_denoise(torch.cat([r[0] for r in reqs]), torch.cat([r[1] for r in reqs]),
                                           torch.cat([r[2] for r in reqs]), (torch.cat([r[3][0] for r in reqs]), c_in, ei))
        acts, tok_w = ei.get("ca_layers_activations"), ei.get("ca_tokmap_weights")
        res = []
        for e, o, b in zip(eis, offs, sizes):
            if acts is not None:
                e["ca_layers_activations"] = {key: {li: v[o:o + b] for li, v in d.items()} for key, d in acts.items()}
            if tok_w is not None:
                e["ca_tokmap_weights"] = tok_w[o:o + b]
            res.append((out[o:o + b], x_noisy[o:o + b]))
        return res

    def _merged_indices(self, index_sets, offs):
        """[(instance idx, token idx), ...] of the requests -> one pair on the merged batch.  Cached by the identity (and version)
        of the parts: the conditioning side hands out the same tensors every iteration, and the token-weight matrix built from
        the merged pair is cached by ITS identity (ldm/util.py token_weight_matrix)."""
        cache = self.__dict__.setdefault("_merged_index_cache", {})
        key = tuple((id(bi), id(ti), bi._version, ti._version) for bi, ti in index_sets) + tuple(offs)
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 16:
                cache.clear()
            merged = (torch.cat([bi + o for (bi, _), o in zip(index_sets, offs)]), torch.cat([ti for _, ti in index_sets]))
            hit = cache[key] = (merged, [t for pair in index_sets for t in pair])           # (the parts are kept alive: ids stay theirs)
        return hit[0]

    def _training_window_fused(self, batches, optimizer, reducer, scheduler, auto_iteration, step_kwargs, after_forward, after_backward):
        """A window of plain recon micro-batches through ONE UNet pass.  Every micro-batch keeps its own front (iteration flags,
        RNG draws, conditioning, latents -- in order, as in the sequential loop), its own loss, regularisers and gradients
        (``_shared_step_gen`` up to and after its denoising request); the requests are served by one batched forward and the
        gradients enter ONE backward, so ``.grad`` receives g0 + g1 as from two accumulated backwards (the all-reduce of the sum
        is the sum of the all-reduces).  -> [(loss, aux), ...], or {0: kwargs} when micro-batch 0 is not a plain recon iteration
        (nothing has been issued for it then: the caller runs the window on lanes)."""
        n = len(batches)
        gens, reqs, results = [None] * n, [None] * n, [None] * n
        for k, batch in enumerate(batches):
            kw = self._window_kwargs(k, step_kwargs, auto_iteration)
            if k == 0 and not self._is_plain_recon(kw):
                return {0: kw}
            if self._is_plain_recon(kw):
                gens[k] = self._shared_step_gen(batch, **kw)
                # (a front that decides on an iteration type that runs its own passes finishes here: no request)
                reqs[k], results[k] = self._resume(gens[k], None)
            else:
                # a distillation micro-batch behind recon ones: what is pending is served first, then this one runs by itself
                self._serve_requests(gens, reqs, results)
                results[k] = self.shared_step(batch, **kw)
            self.batch_idx += 1
        self._serve_requests(gens, reqs, results)
        if after_forward is not None:
            for k in range(n):
                after_forward(k)
        if reducer is not None:
            reducer.wait()
            reducer.begin_backward()
        roots, grads = [], []
        for loss, grad, model_output, aux in results:
            r, g = self._backward_roots(model_output, grad, aux)
            roots += r
            grads += g
        if roots:
            from .... import functional as HF
            HF.prepare_tokmap_backward(roots, grads)
            torch.autograd.backward(roots, grads)
        if reducer is not None:
            reducer.reduce()
        if after_backward is not None:
            for k in range(n):
                after_backward(k)
        self._optimizer_step(optimizer, reducer, scheduler)
        return [(r[0], r[3]) for r in results]

    def _serve_requests(self, gens, reqs, results):
        pending = [k for k, r in enumerate(reqs) if r is not None and results[k] is None]
        if not pending:
            return
        batch = [reqs[k] for k in pending]
        outs = self._denoise_fused(batch) if len(batch) > 1 and self._denoise_fusable(batch) else \
            [self.guided_denoise(*r) for r in batch]
        for k, o in zip(pending, outs):
            again, results[k] = self._resume(gens[k], o)
            if again is not None:
                raise RuntimeError("training_window: a second denoising request")

    def _training_window(self, batches, optimizer, reducer, scheduler, lanes, auto_iteration, step_kwargs, after_forward, contextlib,
                         after_backward=None, pre_kws=None):
        # The host order is the sequential loop's, micro-batch by micro-batch: preamble (iteration flags, host RNG), the front of
        # ``shared_step`` (latent, trimming, t / noise draws), the conditioning, the timestep annealing -- and only the UNet
        # passes of the plain recon micro-batches (the generator's denoising requests) are held back until every micro-batch's
        # conditioning has been issued.
        # THIRD-PARTY KERNELS STAY ON LANE 0.  The conditioning side (the hook: SubjBasisGenerator / CLIP behind ``cond_fn``, or
        # the yaml-instantiated text encoder + embedding manager) runs vendor GEMMs; hipBLASLt's stream-K kernels wait inside
        # the launch for partial tiles of their other workgroups, and two of them in flight on two streams (the two lanes' hook
        # forwards start within a millisecond of each other) were seen to wait for ever (rocgdb: every wave at
        # ``label_SK_Fixup``; profiles/r04_streams.md).  So every micro-batch's conditioning REQUEST (``_CondRequest``) is
        # served on lane 0, one after the other, in front of lane 0's own UNet pass; its backward then also runs on lane 0
        # (autograd issues a node's backward on its forward's stream), in micro-batch order.  Beside lane 0's vendor kernels
        # only this package's own non-waiting kernels run.  A compositional micro-batch (text encoder, CLIP scoring and
        # decoder inside its own passes) runs on lane 0 as a whole.
        n = len(batches)
        gens, reqs, fronts, snaps, lane_of = [None] * n, [None] * n, [None] * n, [None] * n, list(range(n))

        def on_lane(k, closing=False):
            if lanes is None:
                return contextlib.nullcontext()
            return lanes.micro_batch(k, closing=closing, lane=lane_of[k])

        def cond_server(k):
            if lanes is None or lane_of[k] % len(lanes.streams) == 0:
                return None                                   # in place: this micro-batch runs on lane 0 anyway
            lane = lanes.streams[lane_of[k] % len(lanes.streams)]

            def serve(req):
                ev = torch.cuda.Event()
                ev.record(lane)                               # (what the front made on lane k and the request may read)
                lanes.main.wait_event(ev)
                with torch.cuda.stream(lanes.main):
                    cond = req.make()
                    ev = torch.cuda.Event()
                    ev.record(lanes.main)
                lanes.hold(cond, lane_of[k])
                lane.wait_event(ev)
                return cond
            return serve

        for k, batch in enumerate(batches):
            with on_lane(k):
                if pre_kws is not None and k in pre_kws:          # (already evaluated by the fused attempt: not twice)
                    kw = pre_kws[k]
                else:
                    kw = self._window_kwargs(k, step_kwargs, auto_iteration)
            if lanes is not None and k > 0 and auto_iteration is not None and self.iter_flags.get("is_compos_iter"):
                lanes.main.wait_stream(lanes.streams[k % len(lanes.streams)])     # (what evaluating the kwargs queued there)
                lane_of[k] = 0
            with on_lane(k):
                snaps[k] = (self.iter_flags, getattr(self, "training_percent", 0.0))
                gens[k] = self._shared_step_gen(batch, **kw)
                reqs[k], fronts[k] = self._resume(gens[k], None, cond_server(k))
                if reqs[k] is None and after_forward is not None:          # (an iteration that ran its own passes: issued)
                    after_forward(k)
            self.batch_idx += 1
        for k in range(n):
            if reqs[k] is None:
                continue
            with on_lane(k):
                self.iter_flags, self.training_percent = snaps[k]
                again, fronts[k] = self._resume(gens[k], self.guided_denoise(*reqs[k]), cond_server(k))
                if again is not None:
                    raise RuntimeError("training_window: a second denoising request")
                if after_forward is not None:
                    after_forward(k)
        self.iter_flags, self.training_percent = snaps[-1]
        for k, (loss, grad, model_output, aux) in enumerate(fronts):
            with on_lane(k, closing=True):
                self._micro_batch_backward(model_output, grad, aux, reducer, lanes)
                if after_backward is not None:
                    after_backward(k)
        if lanes is not None:
            lanes.join()
        self._optimizer_step(optimizer, reducer, scheduler)
        if lanes is not None:
            lanes.window_start()
        return [(f[0], f[3]) for f in fronts]


_LANE_DELAY = None
if os.environ.get("ADAP_DIAG_LANE_DELAY"):                 # "lane,microseconds"
    _LANE_DELAY = tuple(int(v) for v in os.environ["ADAP_DIAG_LANE_DELAY"].split(","))


class MicroBatchLanes:
    """One HIP stream per micro-batch of an accumulation window (``LatentDiffusion.training_window``).

    The window's micro-batches read the same weights (the optimiser steps once per window, ddpm.py:606-633) and meet only
    in the trainable parameters' ``.grad``.  Lane 0 is the stream current at construction (the optimiser runs there), the
    others are side streams that wait for the window's weights (``window_start``).  A tensor hook on every trainable
    parameter -- it runs in front of autograd's accumulation into ``.grad``, on whatever stream that accumulation is issued --
    makes that stream wait until the PREVIOUS micro-batch of the window has finished its backward (and, under data
    parallelism, until its all-reduce has landed: ``GradReducer.wait``), so ``.grad`` receives micro-batch 0, 1, .. in order
    exactly as in the sequential loop; everything before that point of a backward runs freely beside the other lane.
    Not for ``unfreeze_model`` (the UNet's weight gradients are written through raw pointers from the start of a backward:
    ``training_window`` falls back to one stream)."""

    def __init__(self, params, n=2, reducer=None):
        self.main = torch.cuda.current_stream()
        prio = int(os.environ.get("ADAP_LANE_PRIO", "0"))              # (tuning: HIP priority of the lanes' own streams)
        if os.environ.get("ADAP_LANES_OWN_STREAMS", "0") == "1":
            # (tuning) every lane on a stream of its own, the current stream only carries the optimiser: the lanes can then
            # all have a priority above the prefetch stream's (the default stream's priority is fixed)
            self.streams = [torch.cuda.Stream(priority=prio) for _ in range(n)]
            ops.set_gn_single_launch_stream(self.main.device, self.streams[0].cuda_stream)
        else:
            self.streams = [self.main] + [lane_stream(k) for k in range(1, n)]
        self.reducer = reducer
        self._prev_done = None
        self._gated, self._gate_ev = set(), None
        # Every stream this process will use takes its hardware queue NOW, in a fixed order -- the lanes, then each lane's block
        # side lane (functional.side_lane; suspended while the lanes run, used by every other leg of a process).  Which streams
        # end up sharing a queue depends on the order of their first use, and the order "lanes first, side lanes whenever a later
        # leg first needs them" cost config 2's distillation mix 17 % (65 vs 78 images/s: its teacher stream then shares a queue
        # with the student's).
        if os.environ.get("ADAP_DIAG_NO_STREAM_WARMUP") != "1" and params:
            from .... import functional as HF
            probe = next(p for p in params)
            for s_ in self.streams:
                with torch.cuda.stream(s_):
                    torch.zeros(1, device=probe.device)
                    lane = HF.side_lane(probe)
                    if lane is not None:
                        with lane:
                            torch.zeros(1, device=probe.device)
        self._window = torch.cuda.Event()
        self._window.record(self.main)
        self._hooks = [] if os.environ.get("ADAP_DIAG_LANES_NO_GATE") == "1" else \
            [p.register_hook(self._gate) for p in params if p.requires_grad]
        # autograd accumulates a leaf's gradient on the stream its AccumulateGrad node was created on (lane 0's, typically),
        # synchronised with the producing lane: intended here, and ordered by the gate
        self._warn_set = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if self._warn_set is not None:
            self._warn_set(False)

    def _gate(self, g):
        """in front of every accumulation into a trainable ``.grad``: the accumulating stream -- WHICHEVER it is; a gradient
        produced from conditioning made on another stream arrives on that stream -- waits for the previous micro-batch's
        backward and, under data parallelism, for its all-reduce.  The first gate of a backward consumes the exchange
        (``GradReducer.wait`` is one-shot: it also applies gloo's 1/world scale) and records ONE event behind it; every
        other stream that reaches a gate in the same backward waits for that event."""
        prev = self._prev_done
        if prev is None:
            return g
        s = torch.cuda.current_stream()
        key = s.cuda_stream
        if key in self._gated:
            return g
        if self._gate_ev is None:
            s.wait_event(prev)
            if self.reducer is not None:
                self.reducer.wait()
            self._gate_ev = torch.cuda.Event()
            self._gate_ev.record(s)
        else:
            s.wait_event(self._gate_ev)
        self._gated.add(key)
        return g

    def micro_batch(self, k, closing=False, lane=None):
        """context: the stream of the window's k-th micro-batch (``lane``: run it on that lane instead of lane k).
        ``closing``: this context issues the micro-batch's backward; on exit its end is the event the next micro-batch's
        accumulation waits for."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            s = self.streams[(k if lane is None else lane) % len(self.streams)]
            if s is not self.main and not closing:
                s.wait_event(self._window)
            if closing:
                self._gated, self._gate_ev = set(), None
                if k == 0:
                    self._prev_done = None             # (the optimiser step on lane 0 ordered everything before this window)
            with torch.cuda.stream(s):
                if _LANE_DELAY is not None and (k if lane is None else lane) % len(self.streams) == _LANE_DELAY[0]:
                    # DIAGNOSTIC (tools/lane_skew_soak.sh): one idle workgroup holds this lane back, so that the lanes drift apart
                    # by far more than they ever do alone -- a missing dependency between them then shows as a changed loss
                    ops._lib.call("adap_debug_occupy", 1, 64, _LANE_DELAY[1], 0, ops._stream())
                yield s
                if closing:
                    ev = torch.cuda.Event()
                    ev.record(s)
                    self._prev_done = ev
        return cm()

    def hold(self, obj, k):
        """tensors made on lane 0 and read by lane ``k``'s kernels (a micro-batch's conditioning): tell the caching allocator,
        or a block freed after the window could be handed out on lane 0 while lane ``k`` still reads it."""
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(self.streams[k % len(self.streams)])
        elif isinstance(obj, dict):
            for v in obj.values():
                self.hold(v, k)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                self.hold(v, k)

    def join(self):
        """lane 0 waits for the others (call before the optimiser step)."""
        for s in self.streams[1:]:
            self.main.wait_stream(s)
        self._prev_done = None

    def window_start(self):
        """the weights of the next window are final at this point of lane 0 (call after the optimiser step / zero_grad)."""
        self._window = torch.cuda.Event()
        self._window.record(self.main)

    def remove(self):
        """take the gates off the parameters and give autograd its stream-mismatch warning back (process-wide setting)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._warn_set is not None:
            self._warn_set(True)
            self._warn_set = None


_PREFETCH_STREAMS = {}


def prefetch_stream(device=None):
    """ONE prefetch stream per device, shared by every prefetcher of the process.  A HIP stream is bound to one of a few
    hardware queues when it is created (four here, dealt round-robin, the default stream on the first), and two streams on one
    queue run one behind the other: with the lanes' stream, the blocks' side lanes and a stream per prefetcher the distillation
    prefetcher's landed on the DEFAULT stream's queue, and config 2's mix (teacher rollout beside the student pass) fell from
    77 to 63 images/s.  Streams are therefore few and shared: prefetchers of different legs never run at the same time."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    s = _PREFETCH_STREAMS.get(idx)
    if s is None:
        s = _PREFETCH_STREAMS[idx] = torch.cuda.Stream(device=idx, priority=int(os.environ.get("ADAP_PF_PRIO", "0")))
    return s


_LANE_STREAMS = {}


def lane_stream(k, device=None):
    """the process-wide stream of micro-batch lane ``k`` >= 1 (``MicroBatchLanes``); also the distillation prefetcher's: a
    second stream of whole UNet passes beside the current one, whichever leg runs (few, shared streams: ``prefetch_stream``)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    s = _LANE_STREAMS.get((idx, k))
    if s is None:
        s = _LANE_STREAMS[(idx, k)] = torch.cuda.Stream(device=idx, priority=int(os.environ.get("ADAP_LANE_PRIO", "0")))
    return s


class LatentPrefetcher:
    """Software pipelining of the no-grad first stage: the VAE encode of micro-batch i+1 is issued on a second HIP
    stream while micro-batch i's UNet forward/backward runs on the main stream.  The UNet's 32x32 .. 8x8 levels launch
    only 16-128 workgroups, so most of the 256 CUs idle during them; the encoder's large grids fill those CUs.  The
    work per step is unchanged (one encode + one UNet pass); only the issue order across streams changes.

        pf = model.make_prefetcher(); pf.submit(batch0, noise0)
        for i: x_start = pf.get(); pf.submit(batch[i+1], noise[i+1]); model.shared_step(batch[i], x_start=x_start, ...)
    """

    def __init__(self, model):
        self.model = model
        self.stream = prefetch_stream()
        self._queue = []                  # FIFO: with MicroBatchLanes a whole window's encodes are in flight

    @property
    def _pending(self):
        return self._queue[0] if self._queue else None

    @_pending.setter
    def _pending(self, v):
        if v is None:
            if self._queue:
                self._queue.pop(0)
        else:
            self._queue.append(v)

    def _hold(self, *tensors):
        """inputs allocated on the main stream are read LATER by side-stream kernels: tell the caching allocator, or a
        temporary the caller drops right after submit() could be handed out again while those kernels still read it
        (for the timestep tensor that means garbage gather indices -> an out-of-bounds access)."""
        for x in tensors:
            if torch.is_tensor(x) and x.is_cuda:
                x.record_stream(self.stream)

    def submit(self, batch, post_noise=None, inline=False):
        """``inline``: encode on the CURRENT stream instead of the prefetcher's own (a micro-batch lane encoding its next
        latent behind its own backward: fewer streams than hardware queues)."""
        main = torch.cuda.current_stream()
        if inline:
            x_start, _ = self.model.get_input(batch, post_noise)
            x_start = x_start.contiguous()
            ev = torch.cuda.Event()
            ev.record(main)
            self._pending = (x_start, ev)
            return
        self.stream.wait_stream(main)                 # inputs written on the main stream are visible
        self._hold(post_noise, *batch.values())
        with torch.cuda.stream(self.stream):
            x_start, _ = self.model.get_input(batch, post_noise)
            x_start = x_start.contiguous()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._pending = (x_start, ev)

    def get(self):
        x_start, ev = self._pending
        self._pending = None
        torch.cuda.current_stream().wait_event(ev)
        x_start.record_stream(torch.cuda.current_stream())
        return x_start


class DistillPrefetcher(LatentPrefetcher):
    """The whole no-grad front of a distillation micro-batch on the side stream: VAE encode, trim to HALF_BS, shift t,
    and the teacher's ND-step rollout -- none of it depends on the trainable weights, so it may run while the previous
    micro-batch's student forward/backward (and optimiser step) occupy the main stream.  The teacher's passes run on
    1-2 instances and leave most CUs idle; the student's batched pass fills them.

        pf.submit(batch, post_noise, t, noise, nd);  x_start, t, noise, teacher_out = pf.get()
        model.shared_step(batch, x_start=x_start, t=t, noise=noise, num_denoising_steps=nd,
                          use_arc2face_as_target=True, trim_to_half_batch=False-equivalent handled by the caller ...)
    """

    def __init__(self, model):
        super().__init__(model)
        if os.environ.get("ADAP_DPF_STREAM", "lane") == "lane":
            self.stream = lane_stream(1)        # whole UNet passes (the teacher's rollout): the second compute stream

    def submit(self, batch, post_noise, t, noise, nd, anneal_t=True):
        """``anneal_t``: the recon iteration's timestep annealing, applied BEFORE the multi-step shift as in the reference
        (ddpm.py:2851-2861: distillation iterations are normal-recon iterations and go through both)."""
        m = self.model
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)
        self._hold(post_noise, t, noise, *batch.values())
        with torch.cuda.stream(self.stream):
            x_start, _ = m.get_input(batch, post_noise)
            x_start = x_start.contiguous()
            hb = m.half_batch_size(x_start.shape[0], nd) if nd > 1 else x_start.shape[0]
            x_start, t, noise = x_start[:hb].contiguous(), t[:hb].contiguous(), noise[:hb].contiguous()
            if anneal_t:
                from ...util import probably_anneal_t
                t = probably_anneal_t(t, getattr(m, "training_percent", 0.0), m.num_timesteps, ratio_range=(1, 1.3),
                                      keep_prob_range=(0.4, 0.2))
            t = m.shift_t_for_multistep(t, nd)
            teacher = m.arc2face(m, x_start, noise, t, batch["arc2face_prompt_emb"][:hb], num_denoising_steps=nd)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._pending = (x_start, t, noise, teacher, hb, ev)

    def get(self):
        x_start, t, noise, teacher, hb, ev = self._pending
        self._pending = None
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for ten in [x_start, t, noise] + [x for lst in teacher for x in lst]:
            ten.record_stream(cur)
        return x_start, t, noise, teacher, hb


class Arc2FaceWrapper(nn.Module):
    """The distillation teacher (reference ddpm.py:5402-5478).  There it is diffusers' ``UNet2DConditionModel`` under
    fp16 autocast; its topology is exactly SD-1.5's, so here the same weights (``load_diffusers_state_dict`` maps the
    diffusers key names onto the ldm names) run through this package's ``UNetModel`` -- bf16 MFMA operands, fp32
    accumulation, fp32 residual stream -- with the [B,21,768] Arc2Face prompt embedding repeated over the 16
    conditioned layers (a non-layerwise context IS the same context at every layer).  The Arc2Face text encoder /
    tokenizer (``gen_arc2face_prompt_embs``) stay third-party: the context is an input."""

    def __init__(self, unet_config=None, unet=None):
        super().__init__()
        self.unet = unet if unet is not None else instantiate_from_config(unet_config)
        for p in self.unet.parameters():
            p.requires_grad = False

    def load_diffusers_state_dict(self, sd, strict=True):
        from ...modules.diffusionmodules.openaimodel import diffusers_to_ldm_unet_state_dict
        return self.unet.load_state_dict(diffusers_to_ldm_unet_state_dict(sd), strict=strict)

    @staticmethod
    def layerwise(context):
        """[B, M, D] -> [16*B, M, D], the 16 layers of an instance contiguous (embedding_manager.py:1345-1349)."""
        B, M, D = context.shape
        return context[:, None].expand(B, 16, M, D).reshape(B * 16, M, D).contiguous()

    @torch.no_grad()
    def forward(self, ddpm_model, x_start, noise, t, context, num_denoising_steps=1, relative_ts=None, noises=None):
        """-> (noise_preds, pred_x0s, noises, ts).  ``relative_ts[i]`` ([B] in U(0,1)) / ``noises[i+1]`` replace the
        ``rand_like`` / ``randn_like`` draws of step i when given (parity tests)."""
        assert num_denoising_steps <= 10
        nd = int(num_denoising_steps)
        x_starts, noises_, ts, noise_preds = [x_start], [noise], [t], []
        ctx = self.layerwise(context.float())
        extra = {"use_layerwise_context": True, "use_conv_attn_kernel_size": -1, "iter_type": "arc2face_teacher",
                 "is_training": False, "capture_distill_attn": False, "img_mask": None}
        for i in range(nd):
            x0_i, t_i, n_i = x_starts[i], ts[i], noises_[i]
            x_noisy = ddpm_model.q_sample(x0_i, t_i, n_i)
            noise_pred = self.unet(x_noisy, t_i, context=ctx, extra_info=dict(extra))
            noise_preds.append(noise_pred)
            x_starts.append(ddpm_model.predict_start_from_noise(x_noisy, t_i, noise_pred))
            if i < nd - 1:
                rel = relative_ts[i] if relative_ts is not None else torch.rand_like(t_i.float())
                # long * python float -> float32, as ``t * np.power(...)`` in the reference
                t_lb = t_i * float(np.power(0.5, np.power(nd - 1, -0.3)))
                t_ub = t_i * float(np.power(0.7, np.power(nd - 1, -0.3)))
                ts.append(((t_ub - t_lb) * rel + t_lb).long())
                # (the reference draws randn_like(pred_x0) on a contiguous NCHW tensor; drawing by shape keeps the values
                # independent of the memory layout x0_i happens to have here, e.g. an NCHW view of a pixel-major latent)
                noises_.append(noises[i + 1] if noises is not None
                               else torch.randn(x0_i.shape, device=x0_i.device, dtype=x0_i.dtype))
        return noise_preds, x_starts[1:], noises_, ts

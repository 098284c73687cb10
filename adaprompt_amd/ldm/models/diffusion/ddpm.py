"""The diffusion trainer's hot path on the MI355X kernels (reference
ldm/models/diffusion/ddpm.py): schedule buffers (``register_schedule`` :240-292), ``q_sample``
(:416-419), ``predict_start_from_noise`` (:358-362), first-stage encode + posterior sample
(:955-962, :1381-1419, :1178-1256), ``apply_model`` / ``DiffusionWrapper`` (:2192-2297,
:5505-5544), ``guided_denoise`` (:2483-2532), ``calc_recon_loss`` (:3571-3595) and the
manual-optimisation ``training_step`` (:515-638: backward every micro-batch, clip 0.5 + step +
zero_grad every ``manual_accumulate_grad_batches``-th batch, loss not divided).

What is deliberately NOT here (SURVEY.md section 2 "out of scope" / section 8f "next"): Lightning, the
data pipeline, the CLIP text encoder + EmbeddingManager + SubjBasisGenerator internals (they stay
the reference's own classes behind ``cond_fn``), the Arc2Face teacher, compositional
distillation and its auxiliary losses, DDIM sampling.  ``cond_fn(batch) -> (c_static_emb
[16*B, 77, 768], prompts, extra_info)`` is the embedding hook: whatever produced the context (the
reference's ``get_learned_conditioning``) is called as-is and only its output enters the path.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .... import ops
from ...modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from ...modules.distributions.distributions import DiagonalGaussianDistribution
from ...util import default, instantiate_from_config


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

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, v_posterior=0.0, parameterization="eps", conditioning_key="crossattn",
                 manual_accumulate_grad_batches=2, grad_clip=0.5, **unused):
        super().__init__()
        assert parameterization == "eps"
        self.parameterization = parameterization
        self.v_posterior = v_posterior
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.manual_accumulate_grad_batches = manual_accumulate_grad_batches
        self.grad_clip = grad_clip
        self.register_schedule(given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s)

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


class LatentDiffusion(DDPM):
    """The recon-distillation iteration core of the reference ``LatentDiffusion`` (ddpm.py:710-3457)."""

    def __init__(self, first_stage_config, unet_config, cond_fn=None, scale_factor=0.18215, bg_pixel_weight=0.1,
                 ckpt_path=None, **kwargs):
        kwargs.setdefault("linear_start", 0.00085)
        kwargs.setdefault("linear_end", 0.012)
        super().__init__(unet_config=unet_config, **kwargs)
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        for p in self.first_stage_model.parameters():
            p.requires_grad = False
        self.scale_factor = scale_factor
        self.bg_pixel_weight = bg_pixel_weight
        self.cond_fn = cond_fn
        self.batch_idx = 0
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

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

    def freeze_unet(self):
        """``unfreeze_model: False`` (yaml:26, ddpm.py:775-786)."""
        for p in self.model.parameters():
            p.requires_grad = False

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
        moments = self.encode_first_stage_moments(img, mask)              # pixel-major [B,h,w,8]
        if post_noise is not None:
            post_noise = post_noise.permute(0, 2, 3, 1).contiguous()
        z = self.sample_latent_nhwc(moments, post_noise)                  # pixel-major [B,h,w,4]
        return z.permute(0, 3, 1, 2), mask

    # ---- denoising -------------------------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond):
        """cond = (c_static_emb, c_in, extra_info)  (ddpm.py:2192-2297, the un-tiled branch :2292)."""
        return self.model(x_noisy, t, c_crossattn=[cond])

    def guided_denoise(self, x_start, noise, t, cond, unet_has_grad=True):
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        with torch.set_grad_enabled(unet_has_grad):
            model_output = self.apply_model(x_noisy, t, cond)
        return model_output, x_noisy

    def calc_recon_loss(self, model_output, target, img_mask, fg_mask, fg_pixel_weight=1, bg_pixel_weight=1):
        """returns (loss, d loss / d model_output); both NCHW-shaped."""
        mo = model_output.permute(0, 2, 3, 1)
        tg = target.permute(0, 2, 3, 1)
        im = None if img_mask is None else img_mask.reshape(mo.shape[:-1])
        fg = None if fg_mask is None else fg_mask.reshape(mo.shape[:-1])
        loss, grad = ops.masked_mse(mo.detach(), tg, im, fg, fg_pixel_weight, bg_pixel_weight)
        return loss, grad.permute(0, 3, 1, 2)

    # ---- one micro-batch of pure recon distillation ------------------------------------------------------------
    def shared_step(self, batch, t=None, noise=None, post_noise=None, cond=None, x_start=None):
        """``x_start``: a latent already encoded for this batch (e.g. by ``LatentPrefetcher`` on a side stream while
        the previous micro-batch's UNet pass was running); otherwise the batch is encoded here."""
        if x_start is None:
            x_start, _mask = self.get_input(batch, post_noise)
        B = x_start.shape[0]
        if t is None:
            t = torch.randint(0, self.num_timesteps, (B,), device=x_start.device).long()
        if noise is None:
            noise = torch.randn_like(x_start)
        if cond is None:
            cond = self.cond_fn(batch)
        hw = x_start.shape[-2:]
        fg = batch.get("fg_mask")
        aug = batch.get("aug_mask")
        img_mask = None if aug is None else torch.nn.functional.interpolate(aug[:, None].float(), size=hw, mode="nearest")
        fg_mask = None if fg is None else torch.nn.functional.interpolate(fg[:, None].float(), size=hw, mode="nearest")
        c_emb, c_in, extra_info = cond
        extra_info = dict(extra_info)
        extra_info["img_mask"] = img_mask                                  # ddpm.py:2876
        model_output, x_noisy = self.guided_denoise(x_start, noise, t, (c_emb, c_in, extra_info))
        loss, grad = self.calc_recon_loss(model_output, noise, img_mask, fg_mask, 1.0, self.bg_pixel_weight)
        return loss, grad, model_output, {"x_start": x_start, "x_noisy": x_noisy, "t": t, "extra_info": extra_info}

    def make_prefetcher(self):
        return LatentPrefetcher(self)

    def training_step(self, batch, optimizer=None, reducer=None, scheduler=None, **step_kwargs):
        """manual optimisation (ddpm.py:583-633).  ``reducer`` (adaprompt_amd.parallel.GradReducer) all-reduces
        the trainable gradients after every micro-batch backward, as DDP does in the reference (no no_sync)."""
        loss, grad, model_output, aux = self.shared_step(batch, **step_kwargs)
        if model_output.requires_grad:
            model_output.backward(grad)                                    # == manual_backward(loss)
        if reducer is not None:
            reducer.reduce()
        self.batch_idx += 1
        if optimizer is not None and self.batch_idx % self.manual_accumulate_grad_batches == 0:
            if reducer is not None:
                reducer.wait()
            from ...prodigy import Prodigy
            if isinstance(optimizer, Prodigy):                 # clip fused into the flat-buffer step
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
        return loss, aux


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
        self.stream = torch.cuda.Stream()
        self._pending = None

    def submit(self, batch, post_noise=None):
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)                 # inputs written on the main stream are visible
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

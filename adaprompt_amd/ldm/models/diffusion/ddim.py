"""DDIM sampling on the MI355X UNet (reference ldm/models/diffusion/ddim.py -- "SAMPLING ONLY").

``DDIMSampler(model).sample(S, batch_size, shape, conditioning, guidance_scale=(max, min),
unconditional_conditioning, eta, x_T)`` keeps the reference signature and semantics: uniform / quad timestep
selection (util.py:46-60), per-step classifier-free guidance on a doubled batch in the order (conditional,
unconditional) (ddim.py:229-258), guidance annealing from max to min over the steps (ddim.py:170-183, 213-216), the
DDIM update x_{t-1} = sqrt(a_prev) * pred_x0 + sqrt(1 - a_prev - sigma^2) * eps + sigma * noise (ddim.py:267-291).
The per-step arithmetic around the UNet call is a handful of elementwise torch ops on [B,4,64,64] tensors; the UNet
forward (batch 2B under guidance) is where the time goes and runs on the HIP kernels.

Not built: ``mask`` / ``x0`` inpainting blend, ``score_corrector``, ``quantize_denoised``, ``stochastic_encode`` /
``decode`` (img2img) -- they raise.  As in the reference, ``guidance_scale`` must be a (max, min) pair when guidance
is used (ddim.py:173-177 reads ``max_guide_scale`` before assigning it otherwise)."""
import numpy as np
import torch

from ...modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps, noise_like


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and not attr.is_cuda:
            attr = attr.to(self.model.betas.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        ac = alphas_cumprod.detach().float().cpu()
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(alphacums=ac, ddim_timesteps=self.ddim_timesteps,
                                                                    eta=ddim_eta, verbose=verbose)
        # host-side tables: the loop reads one scalar of each per step
        self.ddim_sigmas = np.asarray(sigmas, dtype=np.float64)
        self.ddim_alphas = np.asarray(alphas, dtype=np.float64)
        self.ddim_alphas_prev = np.asarray(alphas_prev, dtype=np.float64)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - self.ddim_alphas)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, guidance_scale=guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      guidance_scale=1., unconditional_conditioning=None, **kwargs):
        if mask is not None or x0 is not None or score_corrector is not None or quantize_denoised or \
                ddim_use_original_steps or timesteps is not None:
            raise NotImplementedError("DDIMSampler (MI355X): inpainting blend, score corrector, quantisation, original-"
                                      "step sampling and timestep subsets are not built")
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        timesteps = self.ddim_timesteps
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        if isinstance(guidance_scale, (list, tuple)):
            max_guide_scale, min_guide_scale = guidance_scale
        else:
            if unconditional_conditioning is not None and guidance_scale != 1.:
                raise ValueError("guidance_scale must be a (max, min) pair, as in the reference (ddim.py:173-177)")
            max_guide_scale = min_guide_scale = float(guidance_scale)
        max_guide_anneal_steps = total_steps - 1
        guide_scale_step_delta = (max_guide_scale - min_guide_scale) / max(1, max_guide_anneal_steps)
        guide_scale = max_guide_scale
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              noise_dropout=noise_dropout, guidance_scale=guide_scale,
                                              unconditional_conditioning=unconditional_conditioning, **kwargs)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
            guide_scale = guide_scale - guide_scale_step_delta if i <= max_guide_anneal_steps else 1
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      guidance_scale=1., unconditional_conditioning=None, noise=None):
        if use_original_steps or quantize_denoised or score_corrector is not None:
            raise NotImplementedError("p_sample_ddim (MI355X): only the DDIM-subset, eps-parameterised path is built")
        b, device = x.shape[0], x.device
        if unconditional_conditioning is None or guidance_scale == 1.:
            e_t = self.model.apply_model(x, t, c)
        else:
            x_in = torch.cat([x] * 2)
            t_in = torch.cat([t] * 2)
            if isinstance(c, tuple):
                c_c, c_in_c, extra_info = c
                c_u, c_in_u, _ = unconditional_conditioning
                twin_in = None if (c_in_c is None or c_in_u is None) else sum([c_in_c, c_in_u], [])
                c2 = (torch.cat([c_c, c_u]), twin_in, extra_info)          # (conditional, unconditional)
            else:
                c2 = torch.cat([c, unconditional_conditioning])
            e_t, e_t_uncond = self.model.apply_model(x_in, t_in, c2).chunk(2)
            e_t = e_t_uncond + guidance_scale * (e_t - e_t_uncond)
        a_t = float(self.ddim_alphas[index])
        a_prev = float(self.ddim_alphas_prev[index])
        sigma_t = float(self.ddim_sigmas[index])
        sqrt_one_minus_at = float(self.ddim_sqrt_one_minus_alphas[index])
        pred_x0 = (x - sqrt_one_minus_at * e_t) / (a_t ** 0.5)
        dir_xt = ((1. - a_prev - sigma_t ** 2) ** 0.5) * e_t
        x_prev = (a_prev ** 0.5) * pred_x0 + dir_xt
        if sigma_t != 0.:
            unscaled = noise_like(x.shape, device, repeat_noise) if noise is None else noise
            nz = sigma_t * unscaled * temperature
            if noise_dropout > 0.:
                nz = torch.nn.functional.dropout(nz, p=noise_dropout)
            x_prev = x_prev + nz
        return x_prev, pred_x0

    def stochastic_encode(self, *args, **kwargs):
        raise NotImplementedError("img2img encode/decode is not built (SURVEY 8f)")

    decode = stochastic_encode

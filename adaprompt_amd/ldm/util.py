"""Config plumbing of the drop-in boundary: ``instantiate_from_config`` with the reference's
``{target: dotted.path, params: {...}}`` convention (reference ldm/util.py:104-111, 142-147).
Dotted targets of the reference yaml (``ldm.modules.diffusionmodules.openaimodel.UNetModel``,
``ldm.models.autoencoder.AutoencoderKL`` ...) resolve to this package's MI355X implementations,
either through the top-level ``ldm`` alias package of this repo or by the rewrite below."""
import importlib
import importlib.util
from bisect import bisect_right

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim.lr_scheduler import ConstantLR, CosineAnnealingWarmRestarts, CyclicLR, PolynomialLR, SequentialLR

_PREFIX = "adaprompt_amd."


def get_obj_from_str(string, reload=False):
    """``ldm.<...>`` targets resolve to this package's mirror when it has the module, otherwise to whatever ``ldm.<...>``
    imports as -- with a reference checkout behind this repo on ``sys.path`` that is the reference's own module (embedding
    manager, text encoder, LR scheduler, datasets: the top-level ``ldm`` alias package)."""
    module, cls = string.rsplit(".", 1)
    if module.startswith("ldm."):
        try:
            found = importlib.util.find_spec(_PREFIX + module) is not None
        except ModuleNotFoundError:
            found = False
        if found:
            module = _PREFIX + module
    mod = importlib.import_module(module)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config, **kwargs):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()), **kwargs)


def load_config(path):
    """a yaml config as plain dicts / lists, with OmegaConf's number resolution (``2e-4`` without a dot is a float there;
    PyYAML's YAML-1.1 resolver would leave it a string).  The reference loads its configs with ``OmegaConf.load``
    (main.py:822); omegaconf is not a dependency of this package."""
    import re
    import yaml

    class Loader(yaml.SafeLoader):
        pass
    Loader.add_implicit_resolver(
        "tag:yaml.org,2002:float",
        re.compile(r"^[-+]?(?:[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?|\.[0-9_]+(?:[eE][-+]?[0-9]+)?"
                   r"|[0-9][0-9_]*[eE][-+]?[0-9]+|\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"), list("-+0123456789."))
    with open(path) as fh:
        return yaml.load(fh, Loader=Loader)


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


class SequentialLR2(SequentialLR):
    """``SequentialLR`` whose hand-over restarts the next scheduler at its epoch 0 unless that scheduler carries
    ``start_from_epoch_0 = False`` (reference ldm/util.py:26-41)."""

    def step(self):
        self.last_epoch += 1
        idx = bisect_right(self._milestones, self.last_epoch)
        scheduler = self._schedulers[idx]
        if idx > 0 and self._milestones[idx - 1] == self.last_epoch and \
                scheduler.__dict__.get("start_from_epoch_0", True):
            scheduler.step(0)
        else:
            scheduler.step()
        self._last_lr = scheduler.get_last_lr()


def prodigy_schedule(opt, max_steps, warm_up_steps, scheduler_cycles=1, scheduler_type="Linear"):
    """The Prodigy LR schedules of ``configure_optimizers`` (reference ddpm.py:5215-5272), chained by SequentialLR2 after a
    ConstantLR(factor 1) warm-up of ``warm_up_steps``:
      'Linear'  ``scheduler_cycles`` linear decays, each to 0.1/1.1 of the base LR (PolynomialLR power 1 over 1.1 x the
                cycle length), every cycle restarting at the base LR;
      'CosineAnnealingWarmRestarts'  T_0 = int(cycle length), eta_min 0.1;
      'CyclicLR'  triangular 0.1 <-> 1 with half-cycles of cycle length / 2, entered at its peak (``last_epoch`` = half a
                cycle, and the hand-over does not reset it: ``start_from_epoch_0 = False``); as the first half-cycle runs
                downwards, ``scheduler_cycles`` counts half a cycle less."""
    total_cycle_steps = max_steps - warm_up_steps
    ncyc = scheduler_cycles - 0.5 if scheduler_type == "CyclicLR" else scheduler_cycles
    single = total_cycle_steps / ncyc
    last = total_cycle_steps - single * (ncyc - 1)
    milestones = [warm_up_steps]
    schedulers = [ConstantLR(opt, factor=1.0, total_iters=warm_up_steps)]
    if scheduler_type == "Linear":
        ncyc = int(ncyc)
        for c in range(ncyc):
            steps = last if c == ncyc - 1 else single
            if c != ncyc - 1:
                milestones.append(milestones[-1] + steps)
            schedulers.append(PolynomialLR(opt, power=1, total_iters=steps * 1.1))
    elif scheduler_type == "CosineAnnealingWarmRestarts":
        schedulers.append(CosineAnnealingWarmRestarts(opt, T_0=int(single), T_mult=1, eta_min=0.1, last_epoch=-1))
    elif scheduler_type == "CyclicLR":
        schedulers.append(CyclicLR(opt, base_lr=0.1, max_lr=1, step_size_up=single / 2, last_epoch=single / 2 - 1,
                                   cycle_momentum=False))
        schedulers[-1].start_from_epoch_0 = False
    else:
        raise NotImplementedError(f"Prodigy scheduler_type {scheduler_type!r} (ddpm.py:5268-5269 raises as well)")
    return SequentialLR2(opt, schedulers=schedulers, milestones=milestones)


def prodigy_linear_schedule(opt, max_steps, warm_up_steps, scheduler_cycles=1):
    """the shipped config's schedule (yaml:80-84): ``prodigy_schedule(..., scheduler_type='Linear')``."""
    return prodigy_schedule(opt, max_steps, warm_up_steps, scheduler_cycles, "Linear")


# ----------------------------------------------------------------------------------------------------------------
# Regulariser helpers of the recon iteration (reference ldm/util.py: ortho_subtract :280, demean :425,
# calc_ref_cosine_loss :437, ScaleGrad / gen_gradient_scaler :1084-1131, normalize_dict_values :1423,
# normalized_sum :2110, calc_prompt_emb_delta_loss :2037).  Host-side torch on small tensors (token embeddings,
# [instances, pixels] score maps): same names, arguments and return values as the reference.
# ----------------------------------------------------------------------------------------------------------------


def ortho_subtract(a, b, on_last_n_dims=1, return_align_coeffs=False):
    """the component of ``a`` orthogonal to ``b`` over the last ``on_last_n_dims`` dims (w = <a,b> / (<b,b> + 1e-6))."""
    assert a.ndim == b.ndim, "Tensors a and b must have the same number of dimensions"
    if (a.is_cuda and on_last_n_dims == 1 and not return_align_coeffs and a.shape == b.shape and a.dtype == torch.float32
            and b.dtype == torch.float32):
        from .. import functional as HF            # the fused HIP kernel (forward + analytic backward)
        return HF.OrthoRowsFn.apply(a, b)
    shape = None
    if on_last_n_dims > 1:
        full = torch.broadcast_shapes(a.shape, b.shape)
        a, b = a.expand(full), b.expand(full)
        shape = a.shape
        a = a.reshape(*shape[:-on_last_n_dims], -1)
        b = b.reshape(*shape[:-on_last_n_dims], -1)
    # (a row-wise dot product as einsum becomes a batched GEMM with 1x1 outputs: 0.4 ms per call on [4,16,77,768])
    coeff = (a * b).sum(dim=-1) / ((b * b).sum(dim=-1) + 1e-6)
    out = a - coeff[..., None] * b
    if shape is not None:
        out = out.reshape(shape)
        coeff = coeff.reshape(*shape[:-on_last_n_dims], *([1] * on_last_n_dims))
    return (out, coeff) if return_align_coeffs else out


def demean(x, demean_dims=(-1,)):
    if demean_dims is None:
        return x
    return x - x.mean(dim=tuple(demean_dims), keepdim=True)


class ScaleGrad(torch.autograd.Function):
    """identity whose backward multiplies the gradient by ``alpha``."""

    @staticmethod
    def forward(ctx, input_, alpha_, debug=False):
        ctx.alpha = float(alpha_)
        return input_.view_as(input_)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output * ctx.alpha, None, None


class GradientScaler(nn.Module):
    def __init__(self, alpha=1., debug=False):
        super().__init__()
        self._alpha = float(alpha)

    def forward(self, input_):
        return ScaleGrad.apply(input_, self._alpha)


def gen_gradient_scaler(alpha, debug=False):
    """alpha == 1: identity; 0 < alpha: GradientScaler; alpha == 0: torch.detach."""
    if alpha == 1:
        return nn.Identity()
    if alpha > 0:
        return GradientScaler(alpha, debug=debug)
    assert alpha == 0
    return torch.detach


def cosine_loss_rows(rows, ref_rows, exponent=2, do_demean_first=False, ref_grad_scale=0, aim_to_align=True):
    """[..., D] x [..., D] -> [...]: 1 - cos(row, ref^exponent) (or max(0, cos) when not aligning), ref^exponent sign
    preserving, both optionally demeaned over D, the reference's gradient scaled by ``ref_grad_scale``."""
    if rows.is_cuda and exponent in (1, 2, 3) and rows.dtype == torch.float32 and ref_rows.dtype == torch.float32:
        from .. import functional as HF            # the fused HIP kernel (forward + analytic backward)
        return HF.CosineRowsFn.apply(rows, ref_rows.expand_as(rows), do_demean_first, aim_to_align, ref_grad_scale, exponent)
    if do_demean_first:
        rows, ref_rows = demean(rows), demean(ref_rows)
    ref_rows = gen_gradient_scaler(ref_grad_scale)(ref_rows)
    target = ref_rows * ref_rows.abs().pow(exponent - 1)
    flat, flat_t = rows.reshape(-1, rows.shape[-1]), target.reshape(-1, rows.shape[-1])
    label = torch.full_like(flat[:, 0], 1.0 if aim_to_align else -1.0)
    return F.cosine_embedding_loss(flat, flat_t, label, reduction="none").reshape(rows.shape[:-1])


def calc_ref_cosine_loss(delta, ref_delta, batch_mask=None, emb_mask=None, exponent=2, do_demean_first=False,
                         first_n_dims_to_flatten=3, ref_grad_scale=0, aim_to_align=True, margin=0, debug=False):
    """Mean over the counted samples of a (weighted) mean over rows of ``cosine_loss_rows``; a sample's rows are its
    leading ``first_n_dims_to_flatten`` dims flattened, weighted by ``emb_mask`` (a zero weight removes a row).

    The reference walks the batch in a Python loop and drops the rows whose mask is <= 0 by boolean indexing (a device
    -> host sync per sample for the row count).  Here all samples go through ONE set of batched ops: a zero weight
    removes a row from numerator and denominator alike, so weighting every row gives the same number -- a few dozen
    kernel launches per call instead of a few dozen per sample (the recon step calls this 23 times)."""
    B = delta.shape[0]
    if batch_mask is not None:
        assert batch_mask.shape == (B,)
        if batch_mask.sum() == 0:
            return 0
    lead = delta.shape[1:first_n_dims_to_flatten]
    rows = delta.reshape(B, lead.numel(), -1)
    ref_rows = ref_delta.reshape(rows.shape)
    per_row = cosine_loss_rows(rows, ref_rows, exponent, do_demean_first, ref_grad_scale, aim_to_align)      # [B, R]
    if emb_mask is None:
        per_sample = per_row.mean(dim=1)
    else:
        weights = emb_mask.squeeze(-1).expand(B, *lead).reshape(B, -1).clamp(min=0)
        per_sample = (per_row * weights).sum(dim=1) / (weights.sum(dim=1) + 1e-8)
    if batch_mask is None:
        if margin > 0:
            per_sample = torch.clamp(per_sample - margin, min=0)
        return per_sample.mean()
    per_sample = per_sample * batch_mask
    if margin > 0:
        per_sample = torch.clamp(per_sample - margin, min=0)
    return per_sample.sum() / batch_mask.sum()


def normalize_dict_values(d):
    total = np.sum(list(d.values()))
    if total == 0:
        return d
    return {k: v / total for k, v in d.items()}


def to_float(x):
    return x.item() if isinstance(x, torch.Tensor) else x


def normalized_sum(losses_list, norm_pow=0):
    if norm_pow == 0 and len(losses_list) > 2 and all(torch.is_tensor(l) and l.dim() == 0 for l in losses_list):
        return torch.stack(list(losses_list)).sum()          # two launches (and one in the backward) instead of n - 1 adds
    plain = sum(losses_list)
    if norm_pow == 0 or len(losses_list) == 0:
        return plain
    rescaled = sum(l / np.power(np.abs(to_float(l)) + 1e-8, norm_pow) for l in losses_list)
    return rescaled * to_float(plain) / (to_float(rescaled) + 1e-8)


def calc_prompt_emb_delta_loss(static_embeddings, prompt_emb_mask, cls_delta_grad_scale=0.05):
    """static_embeddings [4*BS, 16, 77, 768]: subject-single, subject-comp, class-single, class-comp blocks;
    prompt_emb_mask [4*BS, 77, 1].  The (ortho-subtracted) subject delta comp - single is pulled towards the class
    delta; tokens weigh (m_single + m_comp)^2 / 4 with the start token excluded.  Zeroes ``prompt_emb_mask[:, 0]`` in
    place, as the reference does."""
    subj_single, subj_comp, cls_single, cls_comp = static_embeddings.chunk(4)
    token_weights = None
    if prompt_emb_mask is not None:
        prompt_emb_mask[:, 0] = 0
        m_single, m_comp = prompt_emb_mask.chunk(4)[:2]          # the class prompts have the same masks
        token_weights = ((m_single + m_comp).pow(2) / 4).unsqueeze(1)
    return calc_ref_cosine_loss(ortho_subtract(subj_comp, subj_single), ortho_subtract(cls_comp, cls_single),
                                emb_mask=token_weights, do_demean_first=True, first_n_dims_to_flatten=3,
                                ref_grad_scale=cls_delta_grad_scale, aim_to_align=True)


_TOKEN_WEIGHTS = {}


def token_weight_matrix(index_groups, batch, ntok):
    """[batch, ntok, G] f32, one column per group of (instance idx, token idx) pairs (subject tokens, background
    tokens): w[b, m, g] = how often (b, m) is listed in group g.  A sum over a group's tokens of a [.., ntok] tensor is
    then a contraction with column g.  Cached per index tensors -- the conditioning side reuses them every iteration,
    and building the matrix costs an index_put."""
    # keyed by the tensor OBJECTS (and their versions); the entry keeps them alive, so neither an id nor a device
    # address can come back with other contents while the entry exists
    key = tuple((id(bi), id(ti), bi._version, ti._version) for bi, ti in index_groups) + (batch, ntok)
    hit = _TOKEN_WEIGHTS.get(key)
    if hit is not None:
        return hit[0]
    if len(_TOKEN_WEIGHTS) >= 16:
        _TOKEN_WEIGHTS.clear()
    dev = index_groups[0][0].device
    w = torch.zeros(batch, ntok, len(index_groups), device=dev, dtype=torch.float32)
    for g, (bi, ti) in enumerate(index_groups):
        w[:, :, g].index_put_((bi, ti), torch.ones(bi.numel(), device=dev), accumulate=True)
    _TOKEN_WEIGHTS[key] = (w, [t for pair in index_groups for t in pair])
    from .. import ops
    ops.note_cache_fill()            # (cached: the other micro-batch lane reads it too, ops.note_cache_fill)
    return w


# ----------------------------------------------------------------------------------------------------------------
# helpers of the fg / bg complementary loss (reference ldm/util.py: masked_mean :1450, resize_mask_for_feat_or_attn
# :1570, sel_emb_attns_by_indices :1945)
# ----------------------------------------------------------------------------------------------------------------
def masked_mean(ts, mask, instance_weights=None, dim=None, keepdim=False):
    """sum(ts * instance_weights * mask) / max(sum(mask), 1e-6) over ``dim`` (all dims when None); mask is broadcast to
    ts first so that the count is right; mask None: plain mean of ts * instance_weights."""
    if instance_weights is None:
        instance_weights = 1
    elif isinstance(instance_weights, torch.Tensor):
        instance_weights = instance_weights.view(*instance_weights.shape, *([1] * (ts.ndim - instance_weights.ndim)))
    if mask is None:
        return (ts * instance_weights).mean()
    mask = mask.expand(ts.shape)
    count = mask.sum(dim=dim, keepdim=keepdim).clamp(min=1e-6)
    return (ts * instance_weights * mask).sum(dim=dim, keepdim=keepdim) / count


def resize_mask_for_feat_or_attn(feat_or_attn, mask, mask_name="mask", num_spatial_dims=1, mode="nearest|bilinear",
                                 warn_on_all_zero=False):
    """mask [B,1,H,W] at the (square) resolution of ``feat_or_attn``'s trailing spatial dims: the element-wise maximum of
    the nearest and the bilinear resize.  (The reference's all-zero warning costs a device -> host sync per call; it is
    off by default here.)"""
    side = int(np.sqrt(feat_or_attn.shape[-num_spatial_dims:].numel()))
    out = F.interpolate(mask.float(), size=(side, side), mode="nearest")
    if mode == "nearest|bilinear":
        out = torch.maximum(out, F.interpolate(mask.float(), size=(side, side), mode="bilinear", align_corners=False))
    if warn_on_all_zero and (out.sum(dim=(1, 2, 3)) == 0).any():
        print(f"WARNING: {mask_name} has all-zero masks.")
    return out


def sel_emb_attns_by_indices(attn_mat, indices, all_token_weights=None, do_sum=True, do_mean=False, do_sqrt_norm=False):
    """attn_mat [B, tokens, heads, N]; indices (instance idx, token idx), the instances in ascending blocks -> for every
    listed instance the (weighted) sum / mean over its listed tokens: [n_instances, heads, N].  One contraction with a
    [n_instances, tokens] weight matrix instead of a Python loop over instances."""
    inst, tok = indices
    uniq, inverse, counts = torch.unique(inst, return_inverse=True, return_counts=True)
    w = torch.zeros(len(uniq), attn_mat.shape[1], device=attn_mat.device, dtype=attn_mat.dtype)
    vals = torch.ones(len(tok), device=attn_mat.device, dtype=attn_mat.dtype) if all_token_weights is None \
        else all_token_weights[inst, tok].to(attn_mat.dtype)
    w.index_put_((inverse, tok), vals, accumulate=True)
    if not (do_sum or do_mean):
        raise NotImplementedError("sel_emb_attns_by_indices without a reduction over the tokens is not on the path")
    out = torch.einsum("bthn,bt->bhn", attn_mat[uniq], w)
    if do_mean and not do_sum:
        out = out / counts.view(-1, 1, 1).to(out.dtype)
    if do_sqrt_norm:
        out = out / counts.view(-1, 1, 1).to(out.dtype).sqrt()
    return out


# ----------------------------------------------------------------------------------------------------------------
# timestep annealing of the recon iteration (reference ldm/util.py:1468-1530, applied at ddpm.py:2851-2866)
# ----------------------------------------------------------------------------------------------------------------
def anneal_value(training_percent, final_percent, value_range):
    assert 0 - 1e-6 <= training_percent <= 1 + 1e-6
    v_init, v_final = value_range
    return v_init + (v_final - v_init) * training_percent if training_percent < final_percent else v_final


def anneal_array(training_percent, final_percent, begin_array, end_array):
    assert len(begin_array) == len(end_array)
    return anneal_value(training_percent, final_percent, (np.array(begin_array), np.array(end_array)))


def draw_annealed_bool(training_percent, final_percent, true_prob_range):
    import random
    return random.random() < anneal_value(training_percent, final_percent, true_prob_range)


def probably_anneal_t(t, training_percent, num_timesteps, ratio_range, keep_prob_range=(0, 0.5)):
    """with an annealed probability keep ``t``; otherwise redraw every t_i uniformly from
    [clip(int(t_i * lb), 0, T-1), min(int(t_i * ub) + 1, T)).  The keep decision consumes one ``random.random()``.
    Host tensors: the redraw consumes one ``np.random.randint`` per element, exactly like the reference.  Device
    tensors: the same distribution from torch's device generator in three element-wise ops -- the reference's
    ``int(t_i * lb)`` on a device tensor is a device -> host sync per element."""
    if draw_annealed_bool(training_percent, 1.0, keep_prob_range):
        return t.clone()
    lb, ub = ratio_range
    assert lb < ub
    if t.is_cuda:
        lo = (t.double() * lb).floor().clamp(0, num_timesteps - 1)
        hi = ((t.double() * ub).floor() + 1).clamp(max=num_timesteps)
        u = torch.rand(t.shape, device=t.device, dtype=torch.float64)
        return (lo + (u * (hi - lo)).floor()).clamp(max=num_timesteps - 1).to(t.dtype)
    out = t.clone()
    flat, src = out.view(-1), t.reshape(-1).tolist()
    for i, ti in enumerate(src):
        lo = min(max(int(ti * lb), 0), num_timesteps - 1)
        hi = min(int(ti * ub) + 1, num_timesteps)
        flat[i] = int(np.random.randint(lo, hi))
    return out


# ----------------------------------------------------------------------------------------------------------------
# host-side helpers of the conditioning assembly (reference ldm/util.py: repeat_selected_instances :1410,
# add_noise_to_tensor :2123, anneal_add_noise_to_embedding :2144, distribute_embedding_to_M_tokens(_by_dict) :882-932;
# used by LatentDiffusion.forward, ddpm.py:1710-2042)
# ----------------------------------------------------------------------------------------------------------------
def repeat_selected_instances(sel_indices, REPEAT, *args):
    """every non-None argument: pick ``sel_indices`` along dim 0 and tile that REPEAT times along dim 0 (tensors), or
    slice and repeat the list / tuple (subject names, is_face flags) -- the active definition in the reference is the
    second one, ldm/util.py:1856-1870."""
    out = []
    for a in args:
        if a is None:
            out.append(None)
        elif torch.is_tensor(a):
            out.append(a[sel_indices].repeat([REPEAT] + [1] * (a.ndim - 1)))
        elif isinstance(a, (list, tuple)):
            out.append(a[sel_indices] * REPEAT)
        else:
            raise TypeError(f"repeat_selected_instances: {type(a)}")
    return out


def add_noise_to_tensor(ts, noise_std, noise_std_is_relative=True, keep_norm=False, std_dim=-1, norm_dim=-1):
    """ts + N(0, noise_std) -- relative: noise_std times the mean std of ts over ``std_dim``; keep_norm: rescale to
    the original norms over ``norm_dim``."""
    if noise_std_is_relative:
        noise_std = noise_std * ts.std(dim=std_dim).mean().detach()
    noisy = ts + torch.randn_like(ts) * noise_std
    if keep_norm:
        noisy = noisy * ts.norm(dim=norm_dim, keepdim=True) / (noisy.norm(dim=norm_dim, keepdim=True).detach() + 1e-8)
    return noisy


def anneal_add_noise_to_embedding(embeddings, training_percent, begin_noise_std_range, end_noise_std_range, add_noise_prob,
                                  noise_std_is_relative=True, keep_norm=False, std_dim=-1, norm_dim=-1):
    """with probability ``add_noise_prob`` (one ``random.random()``) add noise whose std is drawn (one
    ``np.random.uniform``) from a range annealed from ``begin_`` to ``end_noise_std_range`` over training."""
    import random
    if random.random() > add_noise_prob:
        return embeddings
    if end_noise_std_range is not None:
        lo = anneal_value(training_percent, 1, (begin_noise_std_range[0], end_noise_std_range[0]))
        hi = anneal_value(training_percent, 1, (begin_noise_std_range[1], end_noise_std_range[1]))
    else:
        lo, hi = begin_noise_std_range
    return add_noise_to_tensor(embeddings, np.random.uniform(lo, hi), noise_std_is_relative, keep_norm, std_dim, norm_dim)


def distribute_embedding_to_M_tokens(text_embedding, placeholder_indices_N, divide_scheme="sqrt_M"):
    """text_embedding [B, N, D]: copy the embedding at the FIRST of the M listed token positions to all M of them,
    divided by sqrt(M) ('sqrt_M'), M ('M') or 1 ('none')."""
    if placeholder_indices_N is None:
        return text_embedding
    pos = torch.unique(placeholder_indices_N)
    M = len(pos)
    if M == 1:
        return text_embedding
    div = {"sqrt_M": np.sqrt(M), "M": M, "none": 1, None: 1}[divide_scheme]
    mask = torch.zeros_like(text_embedding)
    mask[:, pos] = 1
    repl = torch.zeros_like(text_embedding)
    repl[:, pos] = text_embedding[:, pos[:1]].repeat(1, M, 1) / div
    return text_embedding * (1 - mask) + repl * mask


def distribute_embedding_to_M_tokens_by_dict(text_embedding, placeholder_indices_dict, divide_scheme="sqrt_M"):
    if placeholder_indices_dict is None:
        return text_embedding
    for indices in placeholder_indices_dict.values():
        if indices is not None and len(indices[1]) > 1:
            text_embedding = distribute_embedding_to_M_tokens(text_embedding, indices[1])
    return text_embedding


# ---- index bookkeeping of the conditioning assembly (reference ldm/util.py:999-1036, 1313-1343) ------------------------
def join_list_of_indices(*indices_list):
    """[(idx_B, idx_N), ...] -> one (idx_B, idx_N) pair, concatenated in order."""
    return (torch.cat([b for b, _ in indices_list], dim=0), torch.cat([n for _, n in indices_list], dim=0))


def join_dict_of_indices_with_key_filter(indices_dict, key_filter_list):
    """the index pairs of the placeholders named in ``key_filter_list``, joined; None when there are none."""
    if indices_dict is None:
        return None
    sel = [v for k, v in indices_dict.items() if k in key_filter_list and v is not None]
    return join_list_of_indices(*sel) if sel else None


def halve_token_indices(token_indices):
    """first half of an (idx_B, idx_N) pair (``chunk(2)[0]`` of each), or of every pair of a dict."""
    if isinstance(token_indices, dict):
        return {k: halve_token_indices(v) for k, v in token_indices.items()}
    if token_indices is None:
        return None
    return (token_indices[0].chunk(2)[0], token_indices[1].chunk(2)[0])


def merge_cls_token_embeddings(prompt_embedding, cls_delta_string_indices, subj_name_to_cls_delta_token_weights):
    """A class-delta string of M tokens ("young woman") stands where the subject prompts have ONE subject token: replace its
    M embeddings by their weighted sum and shift the rest of the prompt left by M-1 (the EOS at the end stays), so the class
    prompts stay token-aligned with the subject prompts.  ``cls_delta_string_indices``: [(batch_i, start_N, M, subj_name)];
    several strings in one prompt accumulate their shifts in start order."""
    if cls_delta_string_indices is None or len(cls_delta_string_indices) == 0:
        return prompt_embedding
    out = prompt_embedding.clone()
    shifted = {}
    for bi, start, M, name in sorted(cls_delta_string_indices, key=lambda x: (x[0], x[1])):
        off = shifted.get(bi, 0)
        w = subj_name_to_cls_delta_token_weights[name].unsqueeze(1).to(prompt_embedding.device)
        out[bi, start - off] = (prompt_embedding[bi, start:start + M] * w).sum(dim=0)
        out[bi, start + 1 - off:-(M + off)] = prompt_embedding[bi, start + M:-1]
        shifted[bi] = off + M - 1
    return out

def __getattr__(name):
    """names of the reference's ``ldm/util.py`` that this mirror does not define (tokenizer / embedding-manager helpers,
    logging utilities: boundary-side code) resolve to the reference's own module when a reference
    checkout follows this repo on ``sys.path`` -- see the top-level ``ldm`` alias package."""
    if name.startswith("__"):
        raise AttributeError(name)
    import sys
    _alias = sys.modules.get("_adaprompt_ldm_alias")
    if _alias is None:
        import ldm as _alias                     # noqa: F811  (installs the finder and registers itself)
    ref = _alias.reference_module("ldm.util") if hasattr(_alias, "reference_module") else None
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"module 'ldm.util' has no attribute {name!r} (not mirrored in adaprompt_amd.ldm.util"
                         + ("" if ref is not None else "; no reference checkout on sys.path to fall through to") + ")")

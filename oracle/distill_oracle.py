"""CPU restatement of the Arc2Face teacher rollout and the multi-step distillation loss of the Stage-1 iteration.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline); the product path never imports it.

Follows, by source text (ldm/models/diffusion/ddpm.py imports pytorch_lightning / insightface / cv2 and cannot be
imported in the build container):
  * ``Arc2FaceWrapper.forward``               ddpm.py:5432-5478   -> arc2face_rollout
  * the timestep draws before the recon branch ddpm.py:2851-2861   -> shift_t_for_multistep (+ probably_anneal_t)
  * the num_denoising_steps draw / HALF_BS     ddpm.py:1839-1859   -> num_denoising_steps_probs, half_batch_size
  * the arc2face-distillation loss             ddpm.py:2950-3039   -> arc2face_distill_loss
``probably_anneal_t`` / ``anneal_value`` / ``anneal_array`` (ldm/util.py:1468-1530) ARE importable and are pinned by
tests/golden/anneal_t.npz; the ddpm.py parts are pinned by known-answer tests only (tests/test_distill_oracle.py):
PARITY UNPINNED against reference outputs for those, by construction of the reference.
The teacher network itself (diffusers ``UNet2DConditionModel`` + fp16 autocast) is third-party and absent; here the
teacher is any eps-predictor ``fn(x_noisy, t, context) -> eps`` (the tests use the fp32 UNet restatement).
"""
import random as _pyrandom

import numpy as np
import torch

from . import ldm_oracle as O


# ---------------------------------------------------------------------------- ldm/util.py:1468-1530
def anneal_value(training_percent, final_percent, value_range):
    assert 0 - 1e-6 <= training_percent <= 1 + 1e-6
    v_init, v_final = value_range
    if training_percent < final_percent:
        return v_init + (v_final - v_init) * training_percent
    return v_final


def probably_anneal_t(t, training_percent, num_timesteps, ratio_range, keep_prob_range=(0, 0.5),
                      py_random=_pyrandom, np_random=np.random):
    """with annealed probability keep t, else redraw every t_i uniformly from [int(t_i*lb), int(t_i*ub)+1) clipped to
    the schedule (util.py:1508-1530).  Consumes one ``random.random()`` and, if not kept, len(t) ``np.random.randint``."""
    t_annealed = t.clone()
    if py_random.random() < anneal_value(training_percent, 1.0, keep_prob_range):
        return t_annealed
    lb, ub = ratio_range
    assert lb < ub
    for i, ti in enumerate(t):
        lo = min(max(int(ti * lb), 0), num_timesteps - 1)
        hi = min(int(ti * ub) + 1, num_timesteps)
        t_annealed[i] = np_random.randint(lo, hi)
    return t_annealed


# ---------------------------------------------------------------------------- ddpm.py:2851-2861
def shift_t_for_multistep(t, num_denoising_steps, num_timesteps=1000):
    """weighted average of t and num_timesteps so that the later, smaller timesteps of a rollout stay sensible."""
    if num_denoising_steps > 1:
        return (4 * t + (num_denoising_steps - 1) * num_timesteps) // (3 + num_denoising_steps)
    return t


# ---------------------------------------------------------------------------- ddpm.py:1839-1859
def num_denoising_steps_probs(max_num_denoising_steps):
    cand = [s for s in (1, 3, 5, 7) if s <= max_num_denoising_steps]
    p = np.array([0.4, 0.3, 0.2, 0.1])[:len(cand)]       # begin_array == end_array: annealing is the identity
    return cand, p / np.sum(p)


def half_batch_size(batch_size, num_denoising_steps):
    """``torch.arange(BS).chunk(ND)[0].shape[0]`` (ddpm.py:1857): 4 -> 4, 2, 1, 1 for ND = 1, 3, 5, 7."""
    if num_denoising_steps <= 1:
        return batch_size
    return torch.arange(batch_size).chunk(int(num_denoising_steps))[0].shape[0]


# ---------------------------------------------------------------------------- ddpm.py:5432-5478
def arc2face_rollout(teacher_eps_fn, sched, x_start, noise, t, context, num_denoising_steps=1, relative_ts=None,
                     noises=None):
    """-> (noise_preds, pred_x0s, noises, ts), lists of length ND (ts, noises: the values USED at step i).
    ``relative_ts[i]`` ~ U(0,1) [B] and ``noises[i+1]`` ~ N(0,1) are the draws of step i (supplied for parity)."""
    assert num_denoising_steps <= 10
    x_starts, noises_, ts, noise_preds = [x_start], [noise], [t], []
    nd = num_denoising_steps
    with torch.no_grad():
        for i in range(nd):
            x0_i, t_i, n_i = x_starts[i], ts[i], noises_[i]
            x_noisy = O.q_sample(sched, x0_i, t_i, n_i)
            eps = teacher_eps_fn(x_noisy, t_i, context)
            noise_preds.append(eps)
            x_starts.append(O.predict_start_from_noise(sched, x_noisy, t_i, eps))
            if i < nd - 1:
                rel = relative_ts[i] if relative_ts is not None else torch.rand_like(t_i.float())
                # long tensor * numpy float64 scalar -> float32 tensor, exactly as in the reference
                t_lb = t_i * np.power(0.5, np.power(nd - 1, -0.3))
                t_ub = t_i * np.power(0.7, np.power(nd - 1, -0.3))
                ts.append(((t_ub - t_lb) * rel + t_lb).long())
                noises_.append(noises[i + 1] if noises is not None else torch.randn_like(x0_i))
    return noise_preds, x_starts[1:], noises_, ts


# ---------------------------------------------------------------------------- ddpm.py:2950-3039
MAX_ACCUMU_BATCH_SIZE = 7


def arc2face_distill_loss(student_eps_fn, sched, teacher_out, img_mask, fg_mask, num_denoising_steps):
    """student passes + loss of the ``use_arc2face_as_target`` branch.  ``student_eps_fn(x_noisy, t) -> eps`` (with
    grad).  Reproduces the reference's indexing literally: the student's step s starts from the teacher's
    ``pred_x0s[s-1]`` -- for s = 0 that is Python's ``[-1]``, the LAST teacher prediction (ddpm.py:2978) -- re-noised
    with the teacher's ``noises[s]`` at ``ts[s]``; targets are the teacher's eps; bg_pixel_weight = 0; the sum of
    the per-step masked MSEs is divided by sqrt(ND) (ddpm.py:3035).
    -> (loss, per-step losses, model_outputs, loss_start_step)"""
    noise_preds, pred_x0s, noises, ts = teacher_out
    nd = num_denoising_steps
    B = pred_x0s[0].shape[0]
    max_num_loss_steps = MAX_ACCUMU_BATCH_SIZE // B
    loss_start_step = max(0, nd - max_num_loss_steps)
    targets = noise_preds[loss_start_step:]
    model_outputs = []
    for s in range(loss_start_step, nd):
        pred_x0, noise2, t2 = pred_x0s[s - 1], noises[s], ts[s]
        x_noisy = O.q_sample(sched, pred_x0, t2, noise2)
        model_outputs.append(student_eps_fn(x_noisy, t2))
    losses = []
    for s in range(nd - loss_start_step):
        l, _ = O.calc_recon_loss(model_outputs[s], targets[s], img_mask, fg_mask, 1.0, 0.0)
        losses.append(l)
    return sum(losses) / np.sqrt(nd), losses, model_outputs, loss_start_step

"""CPU restatement of the reference's Prodigy optimiser step, gradient-norm clip and LR schedule.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker; the product path (adaprompt_amd/ldm/prodigy.py -> adaprompt_hip optim kernels) never imports it.

Follows /root/reference/ldm/prodigy.py:97-252 (``Prodigy.step``), torch's ``clip_grad_norm_`` as called by
Lightning's ``clip_gradients(optimizer, 0.5, "norm")`` at ddpm.py:606-633, and the scheduler built at
ddpm.py:5219-5247 out of ``SequentialLR2`` (ldm/util.py:26-41).  Pinned by tests/golden/prodigy_*.npz, which
tests/golden/make_golden.py captured from the reference's own class.
"""
import math
from bisect import bisect_right

import torch


class ProdigyOracle:
    """Functional restatement over a list of fp32 tensors (one "param group"; the reference requires all
    groups with lr > 0 to share the same lr, prodigy.py:150-151)."""

    def __init__(self, params, lr=1.0, betas=(0.9, 0.999), beta3=None, eps=1e-8, weight_decay=0.0, decouple=True,
                 use_bias_correction=False, safeguard_warmup=False, d0=1e-6, d_coef=1.0, growth_rate=float("inf")):
        self.params = params
        self.lr = lr
        self.beta1, self.beta2 = betas
        self.beta3 = beta3 if beta3 is not None else math.sqrt(self.beta2)     # prodigy.py:113-115
        self.eps, self.weight_decay, self.decouple = eps, weight_decay, decouple
        self.use_bias_correction, self.safeguard_warmup = use_bias_correction, safeguard_warmup
        self.d0, self.d, self.d_max, self.d_coef, self.growth_rate = d0, d0, d0, d_coef, growth_rate
        self.d_numerator, self.d_denom, self.d_hat, self.k = 0.0, 0.0, d0, 0
        self.state = None

    def step(self, grads):
        b1, b2, b3 = self.beta1, self.beta2, self.beta3
        d, k, lr = self.d, self.k, self.lr
        if self.use_bias_correction:                                             # prodigy.py:124-127
            bias_correction = ((1 - b2 ** (k + 1)) ** 0.5) / (1 - b1 ** (k + 1))
        else:
            bias_correction = 1
        dlr = d * lr * bias_correction                                           # prodigy.py:129
        d_numerator = self.d_numerator * b3                                      # prodigy.py:135-136
        d_denom = 0.0
        if self.state is None:                                                   # prodigy.py:166-173
            self.state = [dict(s=torch.zeros_like(p), p0=p.clone(), exp_avg=torch.zeros_like(p),
                               exp_avg_sq=torch.zeros_like(p)) for p in self.params]
        for p, g, st in zip(self.params, grads, self.state):
            if self.weight_decay != 0 and not self.decouple:                     # prodigy.py:160-161
                g = g + self.weight_decay * p
            if lr > 0.0:
                d_numerator += (d / self.d0) * dlr * torch.dot(g.flatten(), (st["p0"] - p).flatten()).item()
                st["exp_avg"].mul_(b1).add_(g, alpha=d * (1 - b1))               # prodigy.py:185-186
                st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=d * d * (1 - b2))
                if self.safeguard_warmup:                                        # prodigy.py:188-191
                    st["s"].mul_(b3).add_(g, alpha=(d / self.d0) * d)
                else:
                    st["s"].mul_(b3).add_(g, alpha=(d / self.d0) * dlr)
                d_denom += st["s"].abs().sum().item()
        if d_denom == 0:                                                         # prodigy.py:200-201
            return False
        d_hat = d
        if lr > 0.0:                                                             # prodigy.py:203-219
            d_hat = self.d_coef * d_numerator / d_denom
            if d == self.d0:
                d = max(d, d_hat)
            self.d_max = max(self.d_max, d_hat)
            d = min(self.d_max, d * self.growth_rate)
        self.d_numerator, self.d_denom, self.d, self.d_hat = d_numerator, d_denom, d, d_hat
        for p, st in zip(self.params, self.state):                               # prodigy.py:231-248
            denom = st["exp_avg_sq"].sqrt().add_(d * self.eps)                   # the NEW d, the OLD dlr
            if self.weight_decay != 0 and self.decouple:
                p.add_(p, alpha=-self.weight_decay * dlr)
            p.addcdiv_(st["exp_avg"], denom, value=-dlr)
        self.k = k + 1
        return True


def clip_grad_norm(grads, max_norm, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_(norm_type=2): coef = max_norm / (total_norm + 1e-6) clamped to <= 1,
    grads scaled in place.  -> total_norm"""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + eps), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def linear_schedule_lrs(base_lr, max_steps, warm_up_steps, scheduler_cycles=1, n=None):
    """LR seen by optimiser step i = 0..n-1 under ConstantLR(factor 1, warm_up_steps) followed by
    ``scheduler_cycles`` x PolynomialLR(power 1, total_iters = cycle_steps * 1.1) chained by SequentialLR2
    (ddpm.py:5219-5247; util.py:26-41: at a milestone the next scheduler restarts from its epoch 0)."""
    total_cycle_steps = max_steps - warm_up_steps
    single = total_cycle_steps / scheduler_cycles
    last = total_cycle_steps - single * (scheduler_cycles - 1)
    milestones, totals = [warm_up_steps], []
    for c in range(int(scheduler_cycles)):
        steps = last if c == int(scheduler_cycles) - 1 else single
        if c != int(scheduler_cycles) - 1:
            milestones.append(milestones[-1] + steps)
        totals.append(steps * 1.1)
    n = max_steps if n is None else n
    out = []
    for e in range(n):
        idx = bisect_right(milestones, e)
        if idx == 0:
            out.append(base_lr)
        else:
            local = e - milestones[idx - 1]
            out.append(base_lr * max(0.0, 1.0 - min(local, totals[idx - 1]) / totals[idx - 1]))
    return out

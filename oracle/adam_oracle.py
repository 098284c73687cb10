"""CPU restatement of the Adam-type optimiser steps the reference's other ``optimizer_type`` values run.

TEST INFRASTRUCTURE ONLY.  Imported by tests/ as the checker; the product path (adaprompt_amd/ldm/adam.py ->
``adap_adam_update``) never imports it.

The algorithm lives in a third-party dependency, not in /root/reference: ``optimizer_type: AdamW`` / ``NAdam`` instantiate
``torch.optim.AdamW`` / ``torch.optim.NAdam`` (ddpm.py:5134-5142) with per-group learning rates, ``weight_decay`` and
``adam_config.betas`` (ddpm.py:5188-5190), everything else at torch's defaults (eps 1e-8; NAdam: momentum_decay 4e-3, L2
rather than decoupled weight decay).  Restated here from the update rules torch documents for the two classes (torch
2.10.0, the version in this image; single-tensor form, no amsgrad / maximize):

  AdamW   p <- p (1 - lr wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2
          p <- p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
  NAdam   g <- g + wd p  (or p <- p (1 - lr wd) when decoupled);  mu_t = b1 (1 - 0.5 * 0.96^(t psi)),  mu_{t+1} likewise
          m, v as above;  den = sqrt(v / (1 - b2^t)) + eps
          p <- p - lr (1 - mu_t) / (1 - prod_{i<=t} mu_i) * g / den - lr mu_{t+1} / (1 - mu_{t+1} prod_{i<=t} mu_i) * m / den

Pinned by tests/test_adam_host.py against torch's own classes on seeded trajectories (two groups, LambdaLR, clip)."""
import torch


class AdamOracle:
    """functional, fp32: ``groups`` = [{'params': [tensors], 'lr': float}, ...]; ``step(grads)`` takes one gradient per
    parameter in group order.  The learning rates may be changed between steps (``groups[i]['lr']``)."""

    def __init__(self, groups, variant="AdamW", betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, momentum_decay=4e-3,
                 decoupled_weight_decay=False):
        assert variant in ("AdamW", "NAdam")
        self.groups, self.variant = groups, variant
        self.b1, self.b2, self.eps, self.wd = betas[0], betas[1], eps, weight_decay
        self.psi, self.decoupled = momentum_decay, decoupled_weight_decay or variant == "AdamW"
        self.t = 0
        self.mu_product = 1.0
        self.m = [[torch.zeros_like(p) for p in g["params"]] for g in groups]
        self.v = [[torch.zeros_like(p) for p in g["params"]] for g in groups]

    def step(self, grads):
        self.t += 1
        t, b1, b2 = self.t, self.b1, self.b2
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        mu = b1 * (1.0 - 0.5 * 0.96 ** (t * self.psi))
        mu_next = b1 * (1.0 - 0.5 * 0.96 ** ((t + 1) * self.psi))
        self.mu_product *= mu
        it = iter(grads)
        for gi, grp in enumerate(self.groups):
            lr = grp["lr"]
            for pi, p in enumerate(grp["params"]):
                g = next(it).to(torch.float32)
                m, v = self.m[gi][pi], self.v[gi][pi]
                if self.decoupled:
                    p.mul_(1.0 - lr * self.wd)
                elif self.wd != 0:
                    g = g + self.wd * p
                m.mul_(b1).add_(g, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                if self.variant == "AdamW":
                    den = v.sqrt() / (bc2 ** 0.5) + self.eps
                    p.addcdiv_(m, den, value=-lr / bc1)
                else:
                    den = (v / bc2).sqrt() + self.eps
                    p.addcdiv_(g, den, value=-lr * (1.0 - mu) / (1.0 - self.mu_product))
                    p.addcdiv_(m, den, value=-lr * mu_next / (1.0 - self.mu_product * mu_next))


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (2-norm over all gradients), in place; -> the norm before clipping."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)

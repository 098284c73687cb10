"""MI355X drop-ins for ``torch.optim.AdamW`` and ``torch.optim.NAdam`` -- the reference's ``optimizer_type: AdamW`` /
``NAdam`` (ddpm.py:5134-5142; built at ddpm.py:5188-5196 with per-group learning rates ``lr * lr_ratio`` and a
``LambdaLR``).  Same constructor arguments, same param-group keys and the same ``state[p]`` keys (``step, exp_avg,
exp_avg_sq`` and NAdam's ``mu_product``) as torch's classes; the arithmetic is ONE launch of ``adap_adam_update``
(csrc/optim.hip) per parameter group over the flat buffers of ``flatopt.FlatParams``.  Everything that depends on the
step count -- bias corrections, NAdam's momentum schedule and its running product -- is a host-side fp64 scalar, so a
step never reads the device.

Differences a caller can observe, all deliberate (the same as ``ldm.prodigy.Prodigy``'s):
  * ``step(clip_norm=0.5)`` fuses ``clip_grad_norm_`` into the step; the clipped gradient is not written back.
  * gradients are never ``None``: a TRAINABLE parameter that received no gradient in a step sees g = 0 (torch would skip
    it, leaving its moments and step count untouched).  ``requires_grad=False`` parameters of a group (as of the first
    step) are laid out behind the group's trainable ones and never touched: no decay, no update, as in torch.
  * ``amsgrad``, ``maximize``, ``capturable``, ``differentiable`` and tensor learning rates raise.
There is no CPU fallback: parameters must be CUDA tensors and the HIP library must load."""
import torch

from .. import _lib
from .flatopt import FlatParams, _stream


class _FlatAdam(FlatParams, torch.optim.Optimizer):
    def __init__(self, params, defaults, **unsupported):
        for k, v in unsupported.items():
            if v:
                raise NotImplementedError(f"{self._name} (MI355X): {k}={v!r} is not built")
        lr, (b1, b2), eps, wd = defaults["lr"], defaults["betas"], defaults["eps"], defaults["weight_decay"]
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError(f"{self._name} (MI355X): tensor learning rates are not built")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= b1 < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {b1}")
        if not 0.0 <= b2 < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {b2}")
        if not 0.0 <= wd:
            raise ValueError(f"Invalid weight_decay value: {wd}")
        super().__init__(params, defaults)
        self._flat = None
        self._m = self._v = None
        self._steps = [0] * len(self.param_groups)

    def _build_flat(self):
        super()._build_flat()
        self._steps += [0] * (len(self.param_groups) - len(self._steps))        # groups added after construction

    def _init_moments(self):
        self._m = torch.zeros_like(self._flat)
        self._v = torch.zeros_like(self._flat)
        gi_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        # torch's state keeps the count as a CPU float tensor per parameter; here a group's parameters share ONE (they
        # always step together), so a step costs one host-side increment per group, not one per tensor
        self._step_t = [torch.tensor(float(t)) for t in self._steps]
        for p, o, k in self._views:
            st = self.state[p]
            st["step"] = self._step_t[gi_of[id(p)]]
            st["exp_avg"] = self._m[o:o + k].view(p.shape)
            st["exp_avg_sq"] = self._v[o:o + k].view(p.shape)
            self._init_param_state(st, gi_of[id(p)])

    def _init_param_state(self, st, gi):
        pass

    def _scalars(self, group, t, gi):
        """-> (decay, weight_decay_coupled, inv_bias_correction2, coef_grad, coef_moment) of step ``t`` (1-based)."""
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None, clip_norm=None):
        """One optimisation step.  ``clip_norm``: fuse ``clip_grad_norm_(params, clip_norm)`` (ddpm.py:606-607)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._flat is None:
            self._build_flat()
        self._gather_stray_grads()
        if self._m is None:
            self._init_moments()
        self._clip(clip_norm)
        self._check_layout()
        st, s = self._state.data_ptr(), _stream()
        for gi, g in enumerate(self.param_groups):
            o, k = self._train_ranges[gi]               # the group's trainable span (flatopt: frozen tensors lie behind it)
            if k == 0:
                continue
            self._steps[gi] += 1
            b1, b2 = g["betas"]
            decay, wdc, inv_bc2, cg, cm = self._scalars(g, self._steps[gi], gi)
            _lib.call("adap_adam_update", self._flat[o:].data_ptr(), self._grad[o:].data_ptr(), self._m[o:].data_ptr(),
                      self._v[o:].data_ptr(), k, st, float(b1), float(b2), float(g["eps"]), float(decay), float(wdc),
                      float(inv_bc2), float(cg), float(cm), s)
            self._step_t[gi] += 1
        self._touched()
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is None:
            self._build_flat()
        loaded = {p: dict(self.state[p]) for p, _, _ in self._views if p in self.state and "exp_avg" in self.state[p]}
        if not loaded:
            return
        gi_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        for p, st in loaded.items():
            self._steps[gi_of[id(p)]] = int(float(st.get("step", 0)))
            self._load_param_state(st, gi_of[id(p)])
        self._init_moments()
        for p, o, k in self._views:
            st = loaded.get(p)
            if st:
                self._m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))

    def _load_param_state(self, loaded, gi):
        pass


class AdamW(_FlatAdam):
    """torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2): decoupled decay
    ``p *= 1 - lr * wd``, then ``p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)``."""
    _name = "AdamW"

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *,
                 maximize=False, foreach=None, capturable=False, differentiable=False, fused=None):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                        foreach=foreach, capturable=capturable, differentiable=differentiable, fused=fused)
        super().__init__(params, defaults, amsgrad=amsgrad, maximize=maximize, capturable=capturable,
                         differentiable=differentiable)

    def _scalars(self, g, t, gi):
        b1, b2 = g["betas"]
        return g["lr"] * g["weight_decay"], 0.0, 1.0 / (1.0 - b2 ** t), 0.0, g["lr"] / (1.0 - b1 ** t)


class NAdam(_FlatAdam):
    """torch.optim.NAdam(params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, momentum_decay=4e-3):
    ``mu_t = b1 (1 - 0.5 * 0.96^(t * momentum_decay))``, ``p -= lr (1 - mu_t) / (1 - prod mu) * g / den +
    lr mu_{t+1} / (1 - mu_{t+1} prod mu) * m / den``, ``den = sqrt(v / (1 - b2^t)) + eps``; weight decay is L2 on the
    gradient unless ``decoupled_weight_decay`` (the reference passes neither: ddpm.py:5139-5142)."""
    _name = "NAdam"

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, momentum_decay=4e-3,
                 decoupled_weight_decay=False, *, foreach=None, maximize=False, capturable=False, differentiable=False):
        if not 0.0 <= momentum_decay:
            raise ValueError(f"Invalid momentum_decay value: {momentum_decay}")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, momentum_decay=momentum_decay,
                        decoupled_weight_decay=decoupled_weight_decay, maximize=maximize, foreach=foreach,
                        capturable=capturable, differentiable=differentiable)
        super().__init__(params, defaults, maximize=maximize, capturable=capturable, differentiable=differentiable)
        self._mu_product = [1.0] * len(self.param_groups)
        self._mu_t = {}

    def _build_flat(self):
        super()._build_flat()
        self._mu_product += [1.0] * (len(self.param_groups) - len(self._mu_product))

    def _init_param_state(self, st, gi):
        if gi not in self._mu_t:
            self._mu_t[gi] = torch.tensor(self._mu_product[gi])       # shared by the group's parameters, like "step"
        st["mu_product"] = self._mu_t[gi]

    def _load_param_state(self, loaded, gi):
        self._mu_product[gi] = float(loaded.get("mu_product", 1.0))
        self._mu_t.pop(gi, None)

    def _scalars(self, g, t, gi):
        b1, b2 = g["betas"]
        md, lr, wd = g["momentum_decay"], g["lr"], g["weight_decay"]
        mu = b1 * (1.0 - 0.5 * (0.96 ** (t * md)))
        mu_next = b1 * (1.0 - 0.5 * (0.96 ** ((t + 1) * md)))
        self._mu_product[gi] *= mu
        mp = self._mu_product[gi]
        if gi in self._mu_t:
            self._mu_t[gi].fill_(mp)
        dec = g["decoupled_weight_decay"]
        return (lr * wd if dec else 0.0), (0.0 if dec else wd), 1.0 / (1.0 - b2 ** t), lr * (1.0 - mu) / (1.0 - mp), \
            lr * mu_next / (1.0 - mp * mu_next)

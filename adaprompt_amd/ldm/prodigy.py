"""MI355X drop-in for the reference's ``ldm.prodigy.Prodigy`` (ldm/prodigy.py:17-252): same constructor, same
param-group keys, same ``state[p]`` keys (``step, s, p0, exp_avg, exp_avg_sq``) -- but the arithmetic runs in the
flat-buffer HIP kernels of csrc/optim.hip.

Layout (``flatopt.FlatParams``, shared with ``ldm.adam``): every parameter of every group is re-pointed (``p.data``)
into ONE flat fp32 buffer, group ranges padded to 16 bytes; ``p.grad`` are views of a second flat buffer (share it with
``adaprompt_amd.parallel.GradReducer`` through ``grad_buffer``), and ``p0 / exp_avg / exp_avg_sq / s`` are four more.
d, d_max, d_numerator, k ... live in a 16-double device array, so ``step()`` issues five kernel launches and never
synchronises with the host (the reference calls ``.item()`` twice per parameter).  ``sync_group_state()`` copies them into ``param_groups`` on demand
(checkpointing, logging).

Differences a caller can observe, all deliberate:
  * ``step(clip_norm=0.5)`` fuses ``clip_grad_norm_`` into the step; the clipped gradient is not written back.
  * gradients are never ``None`` (``zero_grad`` memsets the flat buffer whatever ``set_to_none`` says), so the
    reference's "skip parameters without a gradient" (prodigy.py:154) never triggers: a parameter that received no
    gradient sees g = 0, which changes nothing but the decay of its moments.
  * ``d0`` must be the same in all groups (the reference mixes group-local and leaked loop variables there).
  * FSDP (``fsdp_in_use``) is not supported: raises.
There is no CPU fallback: parameters must be CUDA tensors and the HIP library must load."""
import math

import torch

from .. import _lib
from .flatopt import FlatParams

_ST_KEYS = ("d", "d_max", "d_numerator", "d_denom", "d_hat", "k")


_stream = _lib.current_stream


class Prodigy(FlatParams, torch.optim.Optimizer):
    _name = "Prodigy"

    def __init__(self, params, lr=1.0, betas=(0.9, 0.999), beta3=None, eps=1e-8, weight_decay=0, decouple=True,
                 use_bias_correction=False, safeguard_warmup=False, d0=1e-6, d_coef=1.0, growth_rate=float("inf"),
                 fsdp_in_use=False):
        if not 0.0 < d0:
            raise ValueError("Invalid d0 value: {}".format(d0))
        if not 0.0 < lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 < eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if fsdp_in_use:
            raise NotImplementedError("Prodigy (MI355X): sharded parameters are not supported; trainable weights are "
                                      "replicated per GPU (SURVEY 8e)")
        if decouple and weight_decay > 0:
            print("Using decoupled weight decay")
        defaults = dict(lr=lr, betas=betas, beta3=beta3, eps=eps, weight_decay=weight_decay, d=d0, d0=d0, d_max=d0,
                        d_numerator=0.0, d_coef=d_coef, k=0, growth_rate=growth_rate,
                        use_bias_correction=use_bias_correction, decouple=decouple,
                        safeguard_warmup=safeguard_warmup, fsdp_in_use=fsdp_in_use)
        self.d0 = d0
        super().__init__(params, defaults)
        self._flat = None

    @property
    def supports_memory_efficient_fp16(self):
        return False

    @property
    def supports_flat_params(self):
        return True

    # ------------------------------------------------------------------ flat storage (flatopt.FlatParams)
    def _check_groups(self, groups):
        if any(g["d0"] != self.d0 for g in groups):
            raise RuntimeError("Prodigy (MI355X): d0 must be the same in every parameter group")

    def _state_d0(self):
        return self.d0

    def _build_flat(self):
        super()._build_flat()
        self._p0 = self._m = self._v = self._s = None

    def _init_moments(self):
        # prodigy.py:166-173: state is created at the first step, p0 = the parameters at that moment
        self._p0 = self._flat.clone()
        self._m = torch.zeros_like(self._flat)
        self._v = torch.zeros_like(self._flat)
        self._s = torch.zeros_like(self._flat)
        for p, o, k in self._views:
            st = self.state[p]
            st["step"] = 0
            st["s"] = self._s[o:o + k].view(p.shape)
            st["p0"] = self._p0[o:o + k].view(p.shape)
            st["exp_avg"] = self._m[o:o + k].view(p.shape)
            st["exp_avg_sq"] = self._v[o:o + k].view(p.shape)

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self, closure=None, clip_norm=None):
        """One optimisation step (prodigy.py:97-252).  ``clip_norm``: fuse ``clip_grad_norm_(params, clip_norm)``."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._flat is None:
            self._build_flat()
        self._gather_stray_grads()
        if self._p0 is None:
            self._init_moments()
        g0 = self.param_groups[0]
        beta1, beta2 = g0["betas"]
        beta3 = g0["beta3"] if g0["beta3"] is not None else math.sqrt(beta2)
        lr = max(g["lr"] for g in self.param_groups)
        for g in self.param_groups:
            if g["lr"] not in [lr, 0.0]:
                raise RuntimeError("Setting different lr values in different parameter groups is only supported for "
                                   "values of 0")
        st, ws, s = self._state.data_ptr(), self._ws.data_ptr(), _stream()
        self._clip(clip_norm)
        # frozen (requires_grad=False) tensors of a group lie behind its stepped span (flatopt ``_train_ranges``): the reference
        # skips parameters without a gradient (prodigy.py:160 ``if p.grad is None: continue``) -- no decay, no moments
        self._check_layout()
        active = [i for i, g in enumerate(self.param_groups) if g["lr"] > 0.0 and self._train_ranges[i][1] > 0]
        if not active:                               # every d_denom term is absent: prodigy.py:200-201
            return loss
        ubc = int(bool(g0["use_bias_correction"]))
        for slot, gi in enumerate(active):
            g = self.param_groups[gi]
            o, k = self._train_ranges[gi]
            coupled = float(g["weight_decay"]) if (g["weight_decay"] != 0 and not g0["decouple"]) else 0.0
            _lib.call("adap_prodigy_moments", self._flat[o:].data_ptr(), self._p0[o:].data_ptr(),
                      self._grad[o:].data_ptr(), self._m[o:].data_ptr(), self._v[o:].data_ptr(),
                      self._s[o:].data_ptr(), k, st, ws, slot, float(lr), float(beta1), float(beta2), float(beta3),
                      float(self.d0), coupled, ubc, int(bool(g["safeguard_warmup"])), s)
        _lib.call("adap_prodigy_finish", st, ws, len(active), float(lr), float(beta1), float(beta2), float(beta3),
                  float(self.d0), float(g0["d_coef"]), float(g0["growth_rate"]), ubc, s)
        for gi, g in enumerate(self.param_groups):   # second loop of the reference runs over ALL groups
            o, k = self._train_ranges[gi]
            if k == 0:
                continue
            dec = float(g["weight_decay"]) if (g["weight_decay"] != 0 and g0["decouple"]) else 0.0
            _lib.call("adap_prodigy_update", self._flat[o:].data_ptr(), self._m[o:].data_ptr(),
                      self._v[o:].data_ptr(), k, st, float(g["eps"]), dec, s)
        self._touched()
        return loss

    # ------------------------------------------------------------------ host view of the device state
    def device_state(self):
        """{'d','d_max','d_numerator','d_denom','d_hat','k','clip_coef','grad_norm','skipped'} -- one D2H copy."""
        if self._flat is None:
            self._build_flat()
        v = self._state.cpu().tolist()
        out = dict(zip(_ST_KEYS, v[:6]))
        out["k"] = int(out["k"])
        out.update(clip_coef=v[6], grad_norm=v[7], skipped=bool(v[8]))
        return out

    def sync_group_state(self):
        ds = self.device_state()
        for g in self.param_groups:
            for key in _ST_KEYS:
                g[key] = ds[key]
        for p, _, _ in self._views:
            if p in self.state and "step" in self.state[p]:
                self.state[p]["step"] = ds["k"]
        return ds

    def state_dict(self):
        if self._flat is not None:
            self.sync_group_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is None:
            self._build_flat()
        g0 = self.param_groups[0]
        vals = [float(g0.get(k, 0.0)) for k in _ST_KEYS]
        self._state[:6] = torch.tensor(vals, dtype=torch.float64)
        if any("exp_avg" in self.state.get(p, {}) for p, _, _ in self._views):
            loaded = {p: dict(self.state[p]) for p, _, _ in self._views if p in self.state}
            self._init_moments()
            for p, o, k in self._views:
                st = loaded.get(p)
                if not st or "exp_avg" not in st:
                    continue
                for key, buf in (("s", self._s), ("p0", self._p0), ("exp_avg", self._m), ("exp_avg_sq", self._v)):
                    buf[o:o + k].copy_(st[key].reshape(-1))
                self.state[p]["step"] = int(st.get("step", 0))

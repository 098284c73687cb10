"""The flat-buffer parameter layout the HIP optimisers share (``ldm.prodigy.Prodigy``, ``ldm.adam.AdamW`` / ``NAdam``).

Every parameter of every group is re-pointed (``p.data``) into ONE flat fp32 buffer, group ranges padded to 16 bytes;
``p.grad`` are views of a second flat buffer -- hand it to ``adaprompt_amd.parallel.GradReducer(flat=...)`` and the
data-parallel exchange and the optimiser work on the same memory.  A 16-double device array carries the step's scalars
(index 6: the gradient-clip coefficient ``adap_grad_clip_coef`` writes), so a step never synchronises with the host.
There is no CPU fallback: parameters must be CUDA tensors and the HIP library must load."""
import torch

from .. import _lib

_stream = _lib.current_stream


class FlatParams:
    _flat = None
    _name = "optimiser"

    def _build_flat(self):
        groups = self.param_groups
        plist = [p for g in groups for p in g["params"]]
        if not plist:
            raise ValueError(f"{self._name}: no parameters")
        dev = plist[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"{self._name} (MI355X): parameters must live on the GPU -- there is no CPU path")
        self._check_groups(groups)
        self._ranges, self._train_ranges, self._views, off = [], [], [], 0
        for g in groups:
            start = off
            # a group's trainable parameters first, its requires_grad=False ones behind them: ``_train_ranges`` is the span an
            # optimiser that must leave frozen tensors alone steps over (torch.optim skips parameters without a gradient;
            # the reference's unfreeze group is list(model.parameters()), frozen buffers-as-parameters included)
            ordered = [p for p in g["params"] if p.requires_grad] + [p for p in g["params"] if not p.requires_grad]
            train_end = off
            for p in ordered:
                if p.dtype != torch.float32 or p.device != dev:
                    raise TypeError(f"{self._name} (MI355X): parameters must be fp32 on one device")
                self._views.append((p, off, p.numel()))
                off += p.numel()
                if p.requires_grad:
                    train_end = off
            off = (off + 3) // 4 * 4                      # next group starts 16-byte aligned
            self._ranges.append((start, off - start))
            self._train_ranges.append((start, train_end - start))
        n = off
        f32 = dict(device=dev, dtype=torch.float32)
        self._flat = torch.zeros(n, **f32)
        old_grads = []
        for p, o, k in self._views:
            self._flat[o:o + k].copy_(p.detach().reshape(-1))
            p.data = self._flat[o:o + k].view(p.shape)
            old_grads.append(p.grad)
        self._grad = torch.zeros(n, **f32)
        for (p, o, k), g in zip(self._views, old_grads):
            if g is not None:
                self._grad[o:o + k].copy_(g.reshape(-1))
            p.grad = self._grad[o:o + k].view(p.shape)
        self._state = torch.zeros(16, device=dev, dtype=torch.float64)
        _lib.call("adap_prodigy_state_init", self._state.data_ptr(), float(self._state_d0()), _stream())
        self._ws = torch.zeros(_lib.call_long("adap_optim_workspace_doubles", len(groups)), device=dev,
                               dtype=torch.float64)
        self._n = n
        self._views_trainable = [p.requires_grad for p, _, _ in self._views]

    def _check_layout(self):
        """the layout fixed ``requires_grad`` when the flat buffers were built (trainable tensors first in every group): a
        parameter frozen or unfrozen since would be stepped / skipped wrongly -- rebuild (``zero_grad()`` drops the layout) or
        construct the optimiser after the change."""
        for (p, o, k), was in zip(self._views, self._views_trainable):
            if p.requires_grad != was:
                raise RuntimeError(f"{self._name} (MI355X): a parameter's requires_grad changed after the flat buffers were built "
                                   "(trainable tensors lie in front of the frozen ones of their group); build a new optimiser")

    def add_param_group(self, param_group):
        """torch calls this for every group at construction; once the flat buffers exist the layout is fixed."""
        if self._flat is not None:
            raise RuntimeError(f"{self._name} (MI355X): parameter groups cannot be added once the flat buffers exist "
                               "(every parameter's storage and gradient already live in them)")
        super().add_param_group(param_group)

    def _check_groups(self, groups):
        pass

    def _state_d0(self):
        return 1.0

    @property
    def grad_buffer(self):
        """the flat fp32 gradient buffer all ``p.grad`` are views of (hand it to GradReducer(flat=...))."""
        if self._flat is None:
            self._build_flat()
        return self._grad

    @property
    def param_buffer(self):
        if self._flat is None:
            self._build_flat()
        return self._flat

    def _gather_stray_grads(self):
        """a caller (or autograd after set_to_none) may have replaced p.grad: fold it back into the flat buffer."""
        for p, o, k in self._views:
            g = p.grad
            want = self._grad[o:o + k]
            if g is None:
                want.zero_()
                p.grad = want.view(p.shape)
            elif g.data_ptr() != want.data_ptr():
                want.copy_(g.reshape(-1))
                p.grad = want.view(p.shape)

    def _clip(self, clip_norm):
        """state[6] <- min(1, clip_norm / (||g|| + 1e-6)) over the whole flat gradient, or 1."""
        if clip_norm is not None and clip_norm > 0:
            _lib.call("adap_grad_clip_coef", self._grad.data_ptr(), self._n, float(clip_norm), self._state.data_ptr(),
                      self._ws.data_ptr(), _stream())
        else:
            self._state[6] = 1.0

    def _touched(self):
        # the kernels wrote the parameters through raw pointers: tell torch, so that anything keyed on Tensor._version
        # (the bf16 weight packs of a training UNet, functional.WeightCache) is rebuilt
        torch.autograd.graph.increment_version([p for p, _, _ in self._views])

    def zero_grad(self, set_to_none=False):
        if self._flat is None:
            return super().zero_grad(set_to_none=set_to_none)
        self._gather_stray_grads()
        self._grad.zero_()

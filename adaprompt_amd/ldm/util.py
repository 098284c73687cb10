"""Config plumbing of the drop-in boundary: ``instantiate_from_config`` with the reference's
``{target: dotted.path, params: {...}}`` convention (reference ldm/util.py:104-111, 142-147).
Dotted targets of the reference yaml (``ldm.modules.diffusionmodules.openaimodel.UNetModel``,
``ldm.models.autoencoder.AutoencoderKL`` ...) resolve to this package's MI355X implementations,
either through the top-level ``ldm`` alias package of this repo or by the rewrite below."""
import importlib
from bisect import bisect_right

from torch.optim.lr_scheduler import ConstantLR, PolynomialLR, SequentialLR

_PREFIX = "adaprompt_amd."


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if module.startswith("ldm."):
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


def prodigy_linear_schedule(opt, max_steps, warm_up_steps, scheduler_cycles=1):
    """The 'Linear' Prodigy LR schedule of ``configure_optimizers`` (reference ddpm.py:5219-5247, 5290-5296):
    ConstantLR(factor 1) for ``warm_up_steps``, then ``scheduler_cycles`` linear decays, each to 0.1/1.1 of the base
    LR (PolynomialLR power 1 over 1.1 x the cycle length), chained by SequentialLR2."""
    total_cycle_steps = max_steps - warm_up_steps
    ncyc = int(scheduler_cycles)
    single = total_cycle_steps / scheduler_cycles
    last = total_cycle_steps - single * (scheduler_cycles - 1)
    milestones = [warm_up_steps]
    schedulers = [ConstantLR(opt, factor=1.0, total_iters=warm_up_steps)]
    for c in range(ncyc):
        steps = last if c == ncyc - 1 else single
        if c != ncyc - 1:
            milestones.append(milestones[-1] + steps)
        schedulers.append(PolynomialLR(opt, power=1, total_iters=steps * 1.1))
    return SequentialLR2(opt, schedulers=schedulers, milestones=milestones)

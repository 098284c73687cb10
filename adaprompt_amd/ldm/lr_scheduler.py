"""LR-multiplier schedules the yaml's ``adam_config.scheduler_config`` names (v1-finetune-ada.yaml:65-72: ``target:
ldm.lr_scheduler.LambdaWarmUpCosineScheduler``), handed to ``torch.optim.lr_scheduler.LambdaLR`` by
``configure_optimizers`` for the Adam-type optimisers (ddpm.py:5193-5196).  Host-side arithmetic, one float per
optimiser step.  Mirrors the three classes of the reference's ldm/lr_scheduler.py (:4-34, :36-78, :81-98) with the same
constructor keywords, ``schedule(n)`` / ``__call__(n)`` and ``last_lr`` / ``last_f`` attributes; all three are one
piecewise curve -- a linear ramp ``start -> max`` over the warm-up, then a decay ``max -> min`` over the rest of the
cycle, cosine- or line-shaped."""
import bisect
import math


def _ramp_then_decay(n, warm, f_start, f_max, f_min, span_end, shape):
    """multiplier at step ``n`` of a cycle that warms up for ``warm`` steps and ends at ``span_end``."""
    if n < warm:
        return f_start + (f_max - f_start) * n / warm
    if shape == "cosine":
        frac = min((n - warm) / (span_end - warm), 1.0)
        return f_min + (f_max - f_min) * 0.5 * (1.0 + math.cos(math.pi * frac))
    return f_min + (f_max - f_min) * (span_end - n) / span_end          # "linear": measured from the cycle's start


class LambdaWarmUpCosineScheduler:
    """one cycle; use with a base lr of 1.0 (ldm/lr_scheduler.py:4-34)."""

    def __init__(self, warm_up_steps, lr_min, lr_max, lr_start, max_decay_steps, verbosity_interval=0):
        self.lr_warm_up_steps, self.lr_start, self.lr_min, self.lr_max = warm_up_steps, lr_start, lr_min, lr_max
        self.lr_max_decay_steps = max_decay_steps
        self.verbosity_interval = verbosity_interval
        self.last_lr = 0.

    def schedule(self, n, **kwargs):
        if self.verbosity_interval > 0 and n % self.verbosity_interval == 0:
            print(f"current step: {n}, recent lr-multiplier: {self.last_lr}")
        self.last_lr = _ramp_then_decay(n, self.lr_warm_up_steps, self.lr_start, self.lr_max, self.lr_min,
                                        self.lr_max_decay_steps, "cosine")
        return self.last_lr

    __call__ = schedule


class LambdaWarmUpCosineScheduler2:
    """repeated cycles, every argument a list with one entry per cycle (ldm/lr_scheduler.py:36-78)."""
    _shape = "cosine"

    def __init__(self, warm_up_steps, f_min, f_max, f_start, cycle_lengths, verbosity_interval=0):
        if not len(warm_up_steps) == len(f_min) == len(f_max) == len(f_start) == len(cycle_lengths):
            raise AssertionError("one entry per cycle in every list")
        self.lr_warm_up_steps, self.f_start, self.f_min, self.f_max = warm_up_steps, f_start, f_min, f_max
        self.cycle_lengths = cycle_lengths
        self.cum_cycles = [0]
        for c in cycle_lengths:
            self.cum_cycles.append(self.cum_cycles[-1] + c)
        self.verbosity_interval = verbosity_interval
        self.last_f = 0.

    def find_in_interval(self, n):
        """index of the cycle step ``n`` falls in: a cycle's last step is its end point (n <= cumulative length)."""
        i = bisect.bisect_left(self.cum_cycles, n, lo=1)
        return i - 1 if i < len(self.cum_cycles) else None

    def schedule(self, n, **kwargs):
        c = self.find_in_interval(n)
        n = n - self.cum_cycles[c]
        if self.verbosity_interval > 0 and n % self.verbosity_interval == 0:
            print(f"current step: {n}, recent lr-multiplier: {self.last_f}, current cycle {c}")
        self.last_f = _ramp_then_decay(n, self.lr_warm_up_steps[c], self.f_start[c], self.f_max[c], self.f_min[c],
                                       self.cycle_lengths[c], self._shape)
        return self.last_f

    __call__ = schedule


class LambdaLinearScheduler(LambdaWarmUpCosineScheduler2):
    """the same cycles with a straight-line decay (ldm/lr_scheduler.py:81-98)."""
    _shape = "linear"

"""Golden vectors from the reference's OWN ``LatentDiffusion.configure_optimizers`` (ddpm.py:5134-5345), called unbound on a
bare object carrying the attributes it reads (the import machinery is make_golden_ddpm.py's): for every ``optimizer_type`` /
``prodigy_config.scheduler_type`` / ``scheduler_cycles`` case, the optimiser's class and group structure and the learning
rates of every parameter group before each of ``max_steps`` optimiser steps (``optimizer.step(); scheduler.step()``).

    python tests/golden/make_golden_sched.py        # writes tests/golden/optim_schedules.npz (own process)
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_ddpm import import_reference_ddpm  # noqa: E402


class AttrDict(dict):
    """what the reference reads its OmegaConf nodes as: mapping + attribute access, assignable"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def to_attr(x):
    return AttrDict({k: to_attr(v) for k, v in x.items()}) if isinstance(x, dict) else x


MAX_STEPS = 40
GROUP_SIZES, LR_RATIOS, EXCLUDED = (4, 5, 6), (1.0, 0.1, 2.0), (False, True, False)
ADAM_CONFIG = {"betas": [0.9, 0.993],
               "scheduler_config": {"target": "ldm.lr_scheduler.LambdaWarmUpCosineScheduler",
                                    "params": {"verbosity_interval": 0, "warm_up_steps": 10, "lr_start": 0.01, "lr_max": 1.0,
                                               "lr_min": 0.1}}}
CASES = {
    "prodigy_linear_c1": dict(optimizer_type="Prodigy", scheduler_type="Linear", scheduler_cycles=1),
    "prodigy_linear_c2": dict(optimizer_type="Prodigy", scheduler_type="Linear", scheduler_cycles=2),
    "prodigy_cosine_c1": dict(optimizer_type="Prodigy", scheduler_type="CosineAnnealingWarmRestarts", scheduler_cycles=1),
    "prodigy_cosine_c2": dict(optimizer_type="Prodigy", scheduler_type="CosineAnnealingWarmRestarts", scheduler_cycles=2),
    "prodigy_cyclic_c2": dict(optimizer_type="Prodigy", scheduler_type="CyclicLR", scheduler_cycles=2),
    "adamw": dict(optimizer_type="AdamW"),
    "nadam_unfrozen": dict(optimizer_type="NAdam", unfreeze_model=True),
}


def main():
    D = import_reference_ddpm()
    out = {"max_steps": MAX_STEPS, "cases": json.dumps(CASES), "adam_config": json.dumps(ADAM_CONFIG),
           "group_sizes": np.array(GROUP_SIZES), "lr_ratios": np.array(LR_RATIOS), "excluded": np.array(EXCLUDED)}
    for name, case in CASES.items():
        params = [torch.nn.Parameter(torch.zeros(n)) for n in GROUP_SIZES]
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)          # filtered out (ddpm.py:5170)
        groups = [{"params": [p] + ([frozen] if i == 0 else []), "lr_ratio": r, "excluded_from_prodigy": e}
                  for i, (p, r, e) in enumerate(zip(params, LR_RATIOS, EXCLUDED))]
        fake = types.SimpleNamespace(
            optimizer_type=case["optimizer_type"], learning_rate=4e-4, model_lr=1e-6, weight_decay=0.0, do_zero_shot=True,
            unfreeze_model=case.get("unfreeze_model", False),
            embedding_manager=types.SimpleNamespace(optimized_parameters=lambda: groups),
            cond_stage_model=torch.nn.Linear(2, 1), model=torch.nn.Linear(3, 2),
            trainer=types.SimpleNamespace(max_steps=MAX_STEPS), adam_config=to_attr(ADAM_CONFIG),
            prodigy_config=AttrDict(zs_betas=[0.9, 0.999], betas=[0.985, 0.993], d_coef=2, warm_up_steps=10,
                                    scheduler_cycles=case.get("scheduler_cycles", 1),
                                    scheduler_type=case.get("scheduler_type", "Linear")))
        conf = D.LatentDiffusion.configure_optimizers(fake)
        assert isinstance(conf, list) and len(conf) == 1
        opt, sched = conf[0]["optimizer"], conf[0]["lr_scheduler"]["scheduler"]
        lrs = []
        for _ in range(MAX_STEPS):
            lrs.append([g["lr"] for g in opt.param_groups])
            opt._step_count = getattr(opt, "_step_count", 0) + 1        # (an optimiser step without gradients: LRs only)
            sched.step()
        out[name + "/lrs"] = np.array(lrs, dtype=np.float64)
        out[name + "/group_numel"] = np.array([sum(p.numel() for p in g["params"]) for g in opt.param_groups])
        out[name + "/opt_class"] = type(opt).__name__
        out[name + "/sched_class"] = type(sched).__name__
        out[name + "/betas"] = np.array(opt.param_groups[0]["betas"], dtype=np.float64)
        print(name, type(opt).__name__, type(sched).__name__, out[name + "/group_numel"], np.round(out[name + "/lrs"][::6, 0], 6))
    np.savez_compressed(os.path.join(HERE, "optim_schedules.npz"), **out)


if __name__ == "__main__":
    main()

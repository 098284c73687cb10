"""Golden vectors from the reference's OWN ldm/lr_scheduler.py (:4-98): the LR multipliers its three schedule classes
return, for the shipped ``adam_config.scheduler_config`` (v1-finetune-ada.yaml:65-72 with ``max_decay_steps`` <- 60000,
ddpm.py:5191) and for a two-cycle list configuration.  The fixture holds constructor arguments, step numbers and values.

    python tests/golden/make_golden_lr.py        # writes tests/golden/lr_schedules.npz
"""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("_ref_lr_scheduler", "/root/reference/ldm/lr_scheduler.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

one = dict(warm_up_steps=500, lr_start=0.01, lr_max=1.0, lr_min=0.1, max_decay_steps=60000, verbosity_interval=0)
two = dict(warm_up_steps=[100, 50], f_min=[0.1, 0.2], f_max=[1.0, 0.8], f_start=[0.01, 0.05], cycle_lengths=[1000, 700])
steps_one = np.unique(np.concatenate([np.arange(0, 520), np.arange(520, 60000, 997), [59999, 60000, 60001, 75000]]))
steps_two = np.arange(0, 1701)
out = {"one_kwargs": json.dumps(one), "two_kwargs": json.dumps(two), "steps_one": steps_one, "steps_two": steps_two}
s = R.LambdaWarmUpCosineScheduler(**one)
out["LambdaWarmUpCosineScheduler"] = np.array([s.schedule(int(n)) for n in steps_one], dtype=np.float64)
for name in ("LambdaWarmUpCosineScheduler2", "LambdaLinearScheduler"):
    s = getattr(R, name)(**two)
    out[name] = np.array([s(int(n)) for n in steps_two], dtype=np.float64)
    out[name + "_interval"] = np.array([s.find_in_interval(int(n)) for n in steps_two], dtype=np.int64)
np.savez_compressed(os.path.join(HERE, "lr_schedules.npz"), **out)
print("wrote lr_schedules.npz", {k: (v.shape if hasattr(v, "shape") else len(v)) for k, v in out.items()})

"""Host side of the reference's other optimiser types (``optimizer_type: AdamW | NAdam``, ddpm.py:5134-5142, 5157-5196):
the LR-multiplier schedules against vectors from the reference's own ldm/lr_scheduler.py, ``configure_optimizers``'
group / learning-rate / scheduler assembly, and the per-step scalars ``ldm.adam`` hands to ``adap_adam_update`` -- with the
kernel's formula (include/adaprompt_hip.h) evaluated in fp32 torch on the CPU -- against ``torch.optim.AdamW`` /
``torch.optim.NAdam``, which are what the reference instantiates.  The HIP kernel itself: tests/test_adam_gpu.py."""
import json
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err


def test_lr_schedules_match_reference_golden():
    from ldm import lr_scheduler as L                     # the yaml's dotted target resolves to the mirror
    import adaprompt_amd.ldm.lr_scheduler as M
    assert L.LambdaWarmUpCosineScheduler is M.LambdaWarmUpCosineScheduler
    g = load_golden("lr_schedules")
    one, two = json.loads(str(g["one_kwargs"])), json.loads(str(g["two_kwargs"]))
    s = L.LambdaWarmUpCosineScheduler(**one)
    got = np.array([s.schedule(int(n)) for n in g["steps_one"]])
    np.testing.assert_allclose(got, g["LambdaWarmUpCosineScheduler"].numpy(), rtol=0, atol=1e-15)
    assert s.last_lr == got[-1] and s(0) == one["lr_start"]
    for name in ("LambdaWarmUpCosineScheduler2", "LambdaLinearScheduler"):
        s = getattr(L, name)(**two)
        got = np.array([s(int(n)) for n in g["steps_two"]])
        np.testing.assert_allclose(got, g[name].numpy(), rtol=0, atol=1e-15)
        assert [s.find_in_interval(int(n)) for n in g["steps_two"]] == g[name + "_interval"].tolist()
        assert s.last_f == got[-1]
    with pytest.raises(AssertionError):
        L.LambdaLinearScheduler([1], [0.1, 0.2], [1.0], [0.0], [10])


def test_configure_optimizers_adam_branch():
    """ddpm.py:5157-5196: per-group lr = learning_rate * lr_ratio over the requires-grad parameters, the unfrozen model's
    group at model_lr, adam_config.betas, LambdaLR over the yaml's schedule with max_decay_steps <- max_steps."""
    from adaprompt_amd.ldm.adam import AdamW, NAdam
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion

    class Stub:
        optimizer_type, do_zero_shot = "AdamW", True
        model = torch.nn.Linear(3, 2)
        learning_rate, model_lr = 4e-4, 1e-6
        adam_config = {"betas": [0.9, 0.993],
                       "scheduler_config": {"target": "ldm.lr_scheduler.LambdaWarmUpCosineScheduler",
                                            "params": {"verbosity_interval": 0, "warm_up_steps": 500, "lr_start": 0.01,
                                                       "lr_max": 1.0, "lr_min": 0.1}}}

    a, b, c, d = (torch.nn.Parameter(torch.zeros(n)) for n in (4, 5, 6, 7))
    b.requires_grad_(False)
    groups = [{"params": [a, b], "lr_ratio": 1.0, "excluded_from_prodigy": False},
              {"params": [c], "lr_ratio": 0.1, "excluded_from_prodigy": True},
              {"params": [], "lr_ratio": 3.0, "excluded_from_prodigy": False},
              {"params": [d], "lr_ratio": 2.0, "excluded_from_prodigy": False}]
    text = [torch.nn.Parameter(torch.zeros(2))]
    out = LatentDiffusion.configure_optimizers(Stub(), groups, max_steps=2000, weight_decay=0.0, unfreeze_model=True,
                                               extra_model_parameters=text)
    assert len(out) == 1 and out[0]["frequency"] == 1 and out[0]["lr_scheduler"]["interval"] == "step"
    opt, sched = out[0]["optimizer"], out[0]["lr_scheduler"]["scheduler"]
    assert isinstance(opt, AdamW) and isinstance(sched, torch.optim.lr_scheduler.LambdaLR)
    pg = opt.param_groups
    assert [len(g["params"]) for g in pg] == [1, 1, 1, 3]            # b dropped (no grad), the empty group dropped
    assert pg[0]["params"][0] is a and pg[3]["params"][0] is text[0] and pg[3]["params"][1] is Stub.model.weight
    assert [g["initial_lr"] for g in pg] == [4e-4, 4e-4 * 0.1, 4e-4 * 2.0, 1e-6]
    assert all(tuple(g["betas"]) == (0.9, 0.993) and g["weight_decay"] == 0.0 for g in pg)
    assert pg[1]["excluded_from_prodigy"] and not pg[0]["excluded_from_prodigy"]
    # LambdaLR evaluates the multiplier at construction (step 0) and after every scheduler.step()
    from adaprompt_amd.ldm.lr_scheduler import LambdaWarmUpCosineScheduler
    want = LambdaWarmUpCosineScheduler(500, 0.1, 1.0, 0.01, 2000)
    assert math.isclose(pg[0]["lr"], 4e-4 * want(0), rel_tol=1e-12)
    for n in (1, 2, 3):
        opt._step_count = n                                           # (silences torch's order-of-calls warning)
        sched.step()
        assert math.isclose(pg[2]["lr"], 8e-4 * want(n), rel_tol=1e-12)
    assert "max_decay_steps" not in Stub.adam_config["scheduler_config"]["params"]      # the caller's config is not edited
    Stub.optimizer_type = "NAdam"
    out = LatentDiffusion.configure_optimizers(Stub(), groups, max_steps=2000, weight_decay=0.0, unfreeze_model=False)
    assert isinstance(out[0]["optimizer"], NAdam) and len(out[0]["optimizer"].param_groups) == 3
    del Stub.learning_rate
    with pytest.raises(AttributeError):
        LatentDiffusion.configure_optimizers(Stub(), groups, max_steps=2000, weight_decay=0.0, unfreeze_model=False)


def adam_update_formula(p, g, m, v, b1, b2, eps, decay, wdc, inv_bc2, cg, cm, clip=1.0):
    """include/adaprompt_hip.h, adap_adam_update -- in fp32, the scalars rounded to fp32 as the C entry does."""
    f = lambda x: torch.tensor(x, dtype=torch.float32)               # noqa: E731
    g = g * f(clip) + f(wdc) * p
    p = p - f(decay) * p
    m = f(b1) * m + f(1 - b1) * g
    v = f(b2) * v + f(1 - b2) * g * g
    den = torch.sqrt(v * f(inv_bc2)) + f(eps)
    return p - (f(cg) * g + f(cm) * m) / den, m, v


ADAM_CASES = {
    "AdamW": (torch.optim.AdamW, dict(betas=(0.9, 0.993), weight_decay=0.02)),
    "AdamW_nowd": (torch.optim.AdamW, dict(betas=(0.9, 0.993), weight_decay=0.0)),
    "NAdam": (torch.optim.NAdam, dict(betas=(0.9, 0.993), weight_decay=0.0)),
    "NAdam_l2": (torch.optim.NAdam, dict(betas=(0.8, 0.99), weight_decay=0.01, momentum_decay=0.01)),
    "NAdam_decoupled": (torch.optim.NAdam, dict(betas=(0.9, 0.999), weight_decay=0.05, decoupled_weight_decay=True)),
}
ADAM_SHAPES = [(37, 19), (129,), (5, 3, 3, 3), (1,)]
ADAM_LRS = (4e-4, 3e-3)                   # two groups: shapes [0, 1] and [2, 3]


def adam_data(case, nsteps):
    from adaprompt_amd import synth
    ps = [synth.synthetic_input(f"adam.{case}.p{i}", sh, 0, 0.3).clone() for i, sh in enumerate(ADAM_SHAPES)]
    gs = [[synth.synthetic_input(f"adam.{case}.g{i}.s{t}", sh, 0, 1.0) * (0.01 if t != 3 else 3.0)
           for i, sh in enumerate(ADAM_SHAPES)] for t in range(nsteps)]
    return ps, gs


def torch_optim_trajectory(case, nsteps, clip=0.0):
    """what the reference runs: torch's own optimiser (fp32, CPU) under a LambdaLR, clip_grad_norm_ before each step."""
    cls, kw = ADAM_CASES[case]
    p0, gs = adam_data(case, nsteps)
    ps = [torch.nn.Parameter(p.clone()) for p in p0]
    opt = cls([{"params": ps[:2], "lr": ADAM_LRS[0]}, {"params": ps[2:], "lr": ADAM_LRS[1]}], foreach=False, **kw)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda n: 0.5 + 0.1 * n)
    traj = []
    for t in range(nsteps):
        for p, g in zip(ps, gs[t]):
            p.grad = g.clone()
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(ps, clip)
        opt.step()
        sched.step()
        traj.append(torch.cat([p.detach().flatten() for p in ps]).clone())
    return torch.stack(traj), opt


@pytest.mark.parametrize("case", list(ADAM_CASES))
def test_adam_scalars_with_the_kernel_formula_vs_torch_optim(case):
    from adaprompt_amd.ldm import adam as A
    nsteps = 8
    want, _ = torch_optim_trajectory(case, nsteps)
    cls, kw = ADAM_CASES[case]
    mine = A.AdamW if cls is torch.optim.AdamW else A.NAdam
    p0, gs = adam_data(case, nsteps)
    ps = [torch.nn.Parameter(p.clone()) for p in p0]
    opt = mine([{"params": ps[:2], "lr": ADAM_LRS[0]}, {"params": ps[2:], "lr": ADAM_LRS[1]}], **kw)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda n: 0.5 + 0.1 * n)
    x = [p.detach().clone() for p in ps]
    m = [torch.zeros_like(p) for p in x]
    v = [torch.zeros_like(p) for p in x]
    for t in range(nsteps):
        for gi, g in enumerate(opt.param_groups):
            sc = opt._scalars(g, t + 1, gi)
            for i in (0, 1) if gi == 0 else (2, 3):
                x[i], m[i], v[i] = adam_update_formula(x[i], gs[t][i], m[i], v[i], *g["betas"], g["eps"], *sc)
        opt._step_count = t + 1
        sched.step()
        got = torch.cat([q.flatten() for q in x])
        assert rel_err(got, want[t]) < 2e-6, (t, rel_err(got, want[t]))


def test_flat_adam_refuses_what_it_does_not_build():
    from adaprompt_amd.ldm.adam import AdamW, NAdam
    p = [torch.nn.Parameter(torch.zeros(3))]
    for kw in (dict(amsgrad=True), dict(maximize=True), dict(capturable=True), dict(differentiable=True)):
        with pytest.raises(NotImplementedError):
            AdamW(p, **kw)
    with pytest.raises(NotImplementedError):
        NAdam(p, maximize=True)
    with pytest.raises(ValueError):
        AdamW(p, betas=(1.0, 0.9))
    with pytest.raises(ValueError):
        NAdam(p, momentum_decay=-1.0)
    opt = AdamW(p, lr=1e-3)
    p[0].grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):          # the product never computes on the CPU
        opt.step()


def _sched_cases():
    g = load_golden("optim_schedules")
    return g, json.loads(str(g["cases"]))


@pytest.mark.parametrize("name", ["prodigy_linear_c1", "prodigy_linear_c2", "prodigy_cosine_c1", "prodigy_cosine_c2",
                                  "prodigy_cyclic_c2", "adamw", "nadam_unfrozen"])
def test_configure_optimizers_lr_sequences_match_the_reference(name):
    """every optimizer_type / scheduler_type the mirror builds, against the reference's own ``configure_optimizers``
    (tests/golden/make_golden_sched.py): optimiser and scheduler class, the parameter groups' sizes, and the learning rate of
    every group before each of 40 optimiser steps."""
    from adaprompt_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    g, cases = _sched_cases()
    case = cases[name]
    sizes, ratios, excluded = g["group_sizes"].tolist(), g["lr_ratios"].tolist(), g["excluded"].tolist()
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
    groups = [{"params": [p] + ([frozen] if i == 0 else []), "lr_ratio": r, "excluded_from_prodigy": bool(e)}
              for i, (p, r, e) in enumerate(zip(params, ratios, excluded))]

    class Stub:
        optimizer_type, do_zero_shot = case["optimizer_type"], True
        learning_rate, model_lr = 4e-4, 1e-6
        model = torch.nn.Linear(3, 2)
        adam_config = json.loads(str(g["adam_config"]))

    text = list(torch.nn.Linear(2, 1).parameters())
    max_steps = int(g["max_steps"])
    conf = LatentDiffusion.configure_optimizers(
        Stub(), groups, max_steps=max_steps, weight_decay=0.0, unfreeze_model=case.get("unfreeze_model", False),
        extra_model_parameters=text,
        prodigy_config={"zs_betas": [0.9, 0.999], "betas": [0.985, 0.993], "d_coef": 2, "warm_up_steps": 10,
                        "scheduler_cycles": case.get("scheduler_cycles", 1),
                        "scheduler_type": case.get("scheduler_type", "Linear")})
    opt, sched = conf[0]["optimizer"], conf[0]["lr_scheduler"]["scheduler"]
    assert type(opt).__name__ == str(g[name + "/opt_class"]) and type(sched).__name__ == str(g[name + "/sched_class"])
    assert [sum(p.numel() for p in gr["params"]) for gr in opt.param_groups] == g[name + "/group_numel"].tolist()
    assert list(opt.param_groups[0]["betas"]) == g[name + "/betas"].tolist()
    want = g[name + "/lrs"].numpy()
    for t in range(max_steps):
        got = [gr["lr"] for gr in opt.param_groups]
        np.testing.assert_allclose(got, want[t], rtol=1e-12, atol=1e-15, err_msg=f"step {t}")
        opt._step_count = t + 1
        sched.step()


def test_flat_optimisers_fix_their_groups_once_the_buffers_exist():
    """``add_param_group`` works as in torch until the flat buffers are built (they hold every parameter's storage and
    gradient, so the layout cannot grow afterwards)."""
    from adaprompt_amd.ldm.adam import NAdam
    from adaprompt_amd.ldm.prodigy import Prodigy
    for cls in (NAdam, Prodigy):
        a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(5))
        opt = cls([a], lr=1e-3 if cls is NAdam else 1.0)
        opt.add_param_group({"params": [b]})
        assert len(opt.param_groups) == 2 and opt.param_groups[1]["params"][0] is b
        assert opt.param_groups[1]["betas"] == opt.param_groups[0]["betas"]          # defaults filled in, as torch does
        opt._flat = torch.zeros(8)                   # what _build_flat leaves behind (it needs the GPU)
        with pytest.raises(RuntimeError, match="cannot be added"):
            opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(2))]})


@pytest.mark.parametrize("clip", [0.0, 0.5])
@pytest.mark.parametrize("case", list(ADAM_CASES))
def test_adam_oracle_matches_torch_optim(case, clip):
    """oracle/adam_oracle.py (the documented update rules, restated) against the third-party classes the reference
    instantiates -- the same seeded two-group trajectories under a LambdaLR the HIP optimisers are held to on the GPU."""
    from oracle.adam_oracle import AdamOracle, clip_grad_norm
    nsteps = 8
    want, _ = torch_optim_trajectory(case, nsteps, clip)
    cls, kw = ADAM_CASES[case]
    p0, gs = adam_data(case, nsteps)
    ps = [p.clone() for p in p0]
    groups = [{"params": ps[:2], "lr": ADAM_LRS[0]}, {"params": ps[2:], "lr": ADAM_LRS[1]}]
    okw = {k: v for k, v in kw.items() if k in ("betas", "weight_decay", "momentum_decay", "decoupled_weight_decay")}
    orc = AdamOracle(groups, variant="AdamW" if cls is torch.optim.AdamW else "NAdam", **okw)
    for t in range(nsteps):
        for g, base in zip(groups, ADAM_LRS):
            g["lr"] = base * (0.5 + 0.1 * t)                          # the LambdaLR of torch_optim_trajectory
        grads = [g.clone() for g in gs[t]]
        if clip > 0:
            clip_grad_norm(grads, clip)
        orc.step(grads)
        got = torch.cat([p.flatten() for p in ps])
        assert rel_err(got, want[t]) < 1e-6, (t, rel_err(got, want[t]))

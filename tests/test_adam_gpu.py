"""``adap_adam_update`` behind ``ldm.adam.AdamW`` / ``NAdam`` on the MI355X against what the reference instantiates for
``optimizer_type: AdamW | NAdam`` -- torch's own optimisers (ddpm.py:5134-5142), run in fp32 on the CPU with the same seeded
parameters, gradients, per-group learning rates, LambdaLR and ``clip_grad_norm_``.  Tolerance: relative L2 of the fp32
parameter trajectory (the HIP pass fuses multiply-adds and takes sqrt(v / bc2) where torch's AdamW takes sqrt(v) / sqrt(bc2))."""
import pytest
import torch

from conftest import rel_err
from test_adam_host import ADAM_CASES, ADAM_LRS, adam_data, torch_optim_trajectory

pytestmark = pytest.mark.gpu

TRAJ_TOL = 5e-6
NSTEPS = 8


def _make(case):
    from ldm import adam as A
    cls, kw = ADAM_CASES[case]
    mine = A.AdamW if cls is torch.optim.AdamW else A.NAdam
    p0, gs = adam_data(case, NSTEPS)
    ps = [torch.nn.Parameter(p.cuda()) for p in p0]
    opt = mine([{"params": ps[:2], "lr": ADAM_LRS[0]}, {"params": ps[2:], "lr": ADAM_LRS[1]}], **kw)
    assert opt.grad_buffer.numel() >= sum(p.numel() for p in ps)      # builds the flat buffers; p.grad are views now
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda n: 0.5 + 0.1 * n)
    return ps, gs, opt, sched


@pytest.mark.parametrize("clip", [0.0, 0.5])
@pytest.mark.parametrize("case", list(ADAM_CASES))
def test_flat_adam_matches_torch_optim_trajectory(case, clip):
    want, ref = torch_optim_trajectory(case, NSTEPS, clip)
    ps, gs, opt, sched = _make(case)
    for t in range(NSTEPS):
        for p, g in zip(ps, gs[t]):
            p.grad.copy_(g.cuda())
        opt.step(clip_norm=clip if clip > 0 else None)
        sched.step()
        flat = torch.cat([p.detach().flatten() for p in ps]).cpu()
        assert rel_err(flat, want[t]) < TRAJ_TOL, (t, rel_err(flat, want[t]))
    rps = [p for g in ref.param_groups for p in g["params"]]
    for key in ("exp_avg", "exp_avg_sq"):
        got = torch.cat([opt.state[p][key].flatten() for p in ps]).cpu()
        ref_v = torch.cat([ref.state[p][key].flatten() for p in rps])
        assert rel_err(got, ref_v) < TRAJ_TOL, key
    assert set(opt.state[ps[0]]) == set(ref.state[rps[0]])              # step, exp_avg, exp_avg_sq (+ mu_product)
    assert float(opt.state[ps[0]]["step"]) == float(ref.state[rps[0]]["step"]) == NSTEPS
    if "mu_product" in ref.state[rps[0]]:
        assert abs(float(opt.state[ps[3]]["mu_product"]) - float(ref.state[rps[3]]["mu_product"])) < 1e-6


@pytest.mark.parametrize("case", ["AdamW", "NAdam_l2"])
def test_flat_adam_state_dict_roundtrip_continues_identically(case):
    import copy
    ps, gs, opt, sched = _make(case)
    for t in range(4):
        for p, g in zip(ps, gs[t]):
            p.grad.copy_(g.cuda())
        opt.step(clip_norm=0.5)
        sched.step()
    sd = copy.deepcopy(opt.state_dict())             # groups carry step 4's learning rates; no scheduler from here on
    ps2, _, opt2, _ = _make(case)
    with torch.no_grad():
        for p, q in zip(ps2, ps):
            p.copy_(q)
    opt2.load_state_dict(sd)
    assert float(opt2.state[ps2[0]]["step"]) == 4.0
    for t in range(4, NSTEPS):
        for o, pp in ((opt, ps), (opt2, ps2)):
            for p, g in zip(pp, gs[t]):
                p.grad.copy_(g.cuda())
            o.step(clip_norm=0.5)
    for p, q in zip(ps, ps2):
        assert torch.equal(p, q)


def test_flat_adam_shares_its_gradient_buffer_with_the_reducer_and_counts_versions():
    from adaprompt_amd.parallel import GradReducer
    ps, gs, opt, _ = _make("AdamW")
    red = GradReducer(ps, flat=opt.grad_buffer)
    assert red.bytes_per_reduce == opt.grad_buffer.numel() * 4
    v0 = [p._version for p in ps]
    for p, g in zip(ps, gs[0]):
        p.grad.copy_(g.cuda())
    opt.step()
    assert all(p._version > v for p, v in zip(ps, v0))               # weight packs keyed on the version get rebuilt
    ps[0].grad = None                                                   # a stray None gradient counts as zero
    opt.step()
    assert ps[0].grad is not None and float(ps[0].grad.abs().sum()) == 0.0

"""HIP Prodigy step (csrc/optim.hip behind adaprompt_amd.ldm.prodigy.Prodigy) against the vectors captured from
the reference's ldm/prodigy.py and, at a size the fixtures cannot hold, against the pinned CPU oracle.
Floating point: the tolerances below are relative L2 errors of fp32 trajectories (fp64 partial sums on the GPU
vs fp32 torch.dot per parameter in the reference)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, PRODIGY_CASES, PRODIGY_SHAPES, prodigy_params, prodigy_grads

pytestmark = pytest.mark.gpu

TRAJ_TOL = 5e-6       # parameters after each of the 8 steps
D_TOL = 5e-5          # d, d_max, d_numerator, d_denom, d_hat


def _make(case, **over):
    from ldm.prodigy import Prodigy
    ps = [torch.nn.Parameter(p.cuda()) for p in prodigy_params(case)]
    kw = dict(PRODIGY_CASES[case])
    kw.update(over)
    opt = Prodigy(ps, lr=1.0, **kw)
    assert opt.grad_buffer.numel() >= sum(p.numel() for p in ps)      # builds the flat buffers; p.grad are views now
    return ps, opt


@pytest.mark.parametrize("clip", [0.0, 0.5])
@pytest.mark.parametrize("case", list(PRODIGY_CASES))
def test_prodigy_matches_reference_trajectory(case, clip):
    from ldm.util import prodigy_linear_schedule
    g = load_golden(f"prodigy_{case}_clip{int(clip * 10)}")
    ps, opt = _make(case)
    sched = prodigy_linear_schedule(opt, max_steps=8, warm_up_steps=2, scheduler_cycles=1)
    for step in range(int(g["nsteps"])):
        assert abs(opt.param_groups[0]["lr"] - float(g["lrs"][step])) < 1e-12
        for p, gr in zip(ps, prodigy_grads(case, step)):
            p.grad.copy_(gr.cuda())
        opt.step(clip_norm=clip if clip > 0 else None)
        sched.step()
        flat = torch.cat([p.detach().flatten() for p in ps]).cpu()
        assert rel_err(flat, g["params"][step]) < TRAJ_TOL, (step, rel_err(flat, g["params"][step]))
        ds = opt.device_state()
        want = g["dstate"][step].tolist()
        np.testing.assert_allclose([ds["d"], ds["d_max"], ds["d_numerator"], ds["d_denom"], ds["d_hat"]], want[:5],
                                   rtol=D_TOL)
        assert ds["k"] == int(want[5])
        if clip > 0:
            assert abs(ds["grad_norm"] - float(g["grad_norms"][step])) <= 1e-5 * max(1.0, float(g["grad_norms"][step]))
    for key, mine in (("exp_avg", "exp_avg"), ("exp_avg_sq", "exp_avg_sq"), ("s", "s"), ("p0", "p0")):
        got = torch.cat([opt.state[p][mine].flatten() for p in ps]).cpu()
        assert rel_err(got, g[key]) < TRAJ_TOL, key
    sd = opt.state_dict()                      # reference-shaped: groups carry d/k, state carries the 5 keys
    assert sd["param_groups"][0]["k"] == int(g["dstate"][-1][5])
    assert set(sd["state"][0]) == {"step", "s", "p0", "exp_avg", "exp_avg_sq"}


def test_prodigy_zero_first_step_is_a_noop():
    from ldm.prodigy import Prodigy
    from adaprompt_amd import synth
    g = load_golden("prodigy_zero_first_step")
    p = torch.nn.Parameter(synth.synthetic_input("prodigy.zero.p0", (11,), 0, 0.3).clone().cuda())
    opt = Prodigy([p], lr=1.0)
    p.grad = torch.zeros(11, device="cuda")
    opt.step()
    ds = opt.device_state()
    assert torch.equal(p.detach().cpu(), g["params"]) and ds["k"] == 0 and ds["d"] == float(g["d"]) and ds["skipped"]


def test_prodigy_large_flat_vs_oracle_and_determinism():
    """1.3 M values in ragged tensors, 2 groups (one frozen with lr 0), 12 steps, clip on: HIP vs the pinned oracle;
    and two identical runs are bit-identical (fixed-order fp64 reductions).

    Gradients are those of 0.5 * scale * |p - target|^2 evaluated at each side's own parameters, i.e. correlated
    from step to step as in training, so that d grows by orders of magnitude.  (With i.i.d. random gradients d stays
    at d0 and d_numerator = sum <g, p0 - p> is a pure cancellation residue of differences only ~60 ulp wide: the
    reference's own value then moves by 5e-6 with the host's thread count and by 1.6e-4 if the parameter update is
    rounded once instead of twice -- nothing to pin there.)"""
    from ldm.prodigy import Prodigy
    from oracle.prodigy_oracle import ProdigyOracle, clip_grad_norm
    shapes = [(1021, 517), (768, 1000), (3,), (4099,)]
    kw = dict(betas=(0.9, 0.999), d_coef=2.0, use_bias_correction=True, weight_decay=0.01)
    gen = torch.Generator().manual_seed(7)
    init = [torch.randn(s, generator=gen) * 0.2 for s in shapes]
    target = [torch.randn(s, generator=gen) * 0.2 for s in shapes]
    frozen_init = torch.randn(257, generator=gen)
    nsteps, gscale = 12, 2e-3      # |g| = 0.72 > 0.5: the clip engages on every step

    def run():
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        tg = [t.cuda() for t in target]
        fz = torch.nn.Parameter(frozen_init.clone().cuda())
        opt = Prodigy([{"params": ps}, {"params": [fz], "lr": 0.0}], lr=1.0, **kw)
        assert opt.grad_buffer.data_ptr() == ps[0].grad.data_ptr()
        for _ in range(nsteps):
            for p, t in zip(ps, tg):
                p.grad.copy_((p.detach() - t) * gscale)
            fz.grad.fill_(0.01)
            opt.step(clip_norm=0.5)
            opt.zero_grad()
        return [p.detach().cpu() for p in ps], fz.detach().cpu(), opt.device_state()

    a, fz_a, ds_a = run()
    b, fz_b, ds_b = run()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert ds_a == ds_b
    # oracle: the frozen group's gradient takes part in the clip norm but not in the moments; the reference's second
    # loop still applies decoupled decay to it (prodigy.py:242-243)
    ops = [t.clone() for t in init]
    orc = ProdigyOracle(ops, lr=1.0, **kw)
    fz_ref = frozen_init.clone()
    for _ in range(nsteps):
        gs = [(p - t) * gscale for p, t in zip(ops, target)] + [torch.full((257,), 0.01)]
        clip_grad_norm(gs, 0.5)
        k_before, d_before = orc.k, orc.d
        bc = ((1 - 0.999 ** (k_before + 1)) ** 0.5) / (1 - 0.9 ** (k_before + 1))
        orc.step(gs[:-1])
        fz_ref.add_(fz_ref, alpha=-0.01 * d_before * 1.0 * bc)
    assert orc.d > 100 * orc.d0, "the scenario is meant to make d grow"
    for x, y in zip(a, ops):
        assert rel_err(x, y) < TRAJ_TOL, rel_err(x, y)
    assert rel_err(fz_a, fz_ref) < 1e-6
    np.testing.assert_allclose([ds_a["d"], ds_a["d_max"], ds_a["d_numerator"], ds_a["d_denom"], ds_a["d_hat"]],
                               [orc.d, orc.d_max, orc.d_numerator, orc.d_denom, orc.d_hat], rtol=D_TOL)
    assert ds_a["k"] == orc.k == nsteps


def test_prodigy_state_dict_roundtrip_continues_identically():
    ps, opt = _make("zs")
    for step in range(3):
        for p, gr in zip(ps, prodigy_grads("zs", step)):
            p.grad.copy_(gr.cuda())
        opt.step()
    sd = opt.state_dict()
    snap = [p.detach().clone() for p in ps]
    ps2 = [torch.nn.Parameter(t.clone()) for t in snap]
    from ldm.prodigy import Prodigy
    opt2 = Prodigy(ps2, lr=1.0, **PRODIGY_CASES["zs"])
    opt2.load_state_dict(sd)
    assert ps2[0].grad is not None
    for o, pp in ((opt, ps), (opt2, ps2)):
        for p, gr in zip(pp, prodigy_grads("zs", 3)):
            p.grad.copy_(gr.cuda())
        o.step()
    for x, y in zip(ps, ps2):
        assert torch.equal(x.detach(), y.detach())
    assert opt.device_state() == opt2.device_state()


def test_prodigy_rejects_cpu_and_mixed_lr():
    from ldm.prodigy import Prodigy
    p = torch.nn.Parameter(torch.zeros(4))
    opt = Prodigy([p])
    p.grad = torch.zeros(4)
    with pytest.raises(RuntimeError):
        opt.step()
    a, b = torch.nn.Parameter(torch.zeros(4, device="cuda")), torch.nn.Parameter(torch.zeros(4, device="cuda"))
    opt = Prodigy([{"params": [a]}, {"params": [b], "lr": 0.5}], lr=1.0)
    with pytest.raises(RuntimeError):
        opt.step()

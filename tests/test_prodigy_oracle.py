"""The Prodigy / clip / LR-schedule restatement (oracle/prodigy_oracle.py) against vectors captured from the
reference's own ldm/prodigy.py + torch schedulers under SequentialLR2 (tests/golden/make_golden.py: run_prodigy)."""
import numpy as np
import pytest
import torch

from oracle import prodigy_oracle as PO
from conftest import load_golden, rel_err, PRODIGY_CASES, prodigy_params, prodigy_grads


@pytest.mark.parametrize("clip", [0.0, 0.5])
@pytest.mark.parametrize("case", list(PRODIGY_CASES))
def test_prodigy_trajectory(case, clip):
    g = load_golden(f"prodigy_{case}_clip{int(clip * 10)}")
    ps = prodigy_params(case)
    opt = PO.ProdigyOracle(ps, lr=1.0, **PRODIGY_CASES[case])
    lrs = PO.linear_schedule_lrs(1.0, max_steps=8, warm_up_steps=2, scheduler_cycles=1, n=8)
    np.testing.assert_allclose(lrs, g["lrs"].numpy(), rtol=1e-12, atol=1e-15)
    norms = []
    for step in range(int(g["nsteps"])):
        grads = prodigy_grads(case, step)
        opt.lr = lrs[step]
        if clip > 0:
            norms.append(PO.clip_grad_norm(grads, clip))
        opt.step(grads)
        flat = torch.cat([p.flatten() for p in ps])
        assert rel_err(flat, g["params"][step]) < 2e-6, (step, rel_err(flat, g["params"][step]))
        d, d_max, d_num, d_den, d_hat, k = g["dstate"][step].tolist()
        np.testing.assert_allclose([opt.d, opt.d_max, opt.d_numerator, opt.d_denom, opt.d_hat],
                                   [d, d_max, d_num, d_den, d_hat], rtol=2e-5)
        assert opt.k == int(k)
    if clip > 0:
        np.testing.assert_allclose(norms, g["grad_norms"].numpy(), rtol=1e-6)
        assert max(norms) > clip > min(n for n in norms if n > 0)        # the clip engaged on some steps only
    for key in ("exp_avg", "exp_avg_sq", "s", "p0"):
        mine = torch.cat([st[key].flatten() for st in opt.state])
        assert rel_err(mine, g[key]) < 2e-6, key


def test_prodigy_zero_first_step_is_a_noop():
    g = load_golden("prodigy_zero_first_step")
    from adaprompt_amd import synth
    p = synth.synthetic_input("prodigy.zero.p0", (11,), 0, 0.3).clone()
    opt = PO.ProdigyOracle([p])
    assert opt.step([torch.zeros(11)]) is False
    assert torch.equal(p, g["params"]) and opt.k == int(g["k"]) == 0 and opt.d == float(g["d"])

"""oracle/distill_oracle.py: the importable helpers against vectors captured from the reference (ldm/util.py), the
ddpm.py restatements (not importable: pytorch_lightning / insightface / cv2) against known answers."""
import math
import random

import numpy as np
import torch

from oracle import distill_oracle as D
from oracle import ldm_oracle as O
from conftest import load_golden


def test_probably_anneal_t_matches_reference_draws():
    g = load_golden("anneal_t")
    for row, want in zip(g["cases"].tolist(), g["outs"].tolist()):
        seed, tp, lb, ub, k0, k1 = row[:6]
        t = torch.tensor([int(v) for v in row[6:]])
        random.seed(100 + int(seed))
        np.random.seed(200 + int(seed))
        lb = int(lb) if lb == int(lb) else lb          # the reference was called with ints where they are ints
        got = D.probably_anneal_t(t, tp, 1000, ratio_range=(lb, ub), keep_prob_range=(k0, k1))
        assert got.tolist() == want, (row, got.tolist(), want)
    av = [D.anneal_value(tp, fp, (0.2, 0.9)) for tp in (0.0, 0.3, 0.5, 1.0) for fp in (0.5, 1.0)]
    np.testing.assert_allclose(av, g["anneal_values"].numpy(), rtol=0, atol=0)
    aa = D.anneal_value(0.25, 0.5, (np.array([0.4, 0.3, 0.2, 0.1]), np.array([0.1, 0.2, 0.3, 0.4])))
    np.testing.assert_allclose(aa, g["anneal_array"].numpy(), rtol=0, atol=0)


def test_multistep_bookkeeping_known_answers():
    assert [D.half_batch_size(4, nd) for nd in (1, 3, 5, 7)] == [4, 2, 1, 1]           # ddpm.py:1855-1857 comments
    assert [D.half_batch_size(3, nd) for nd in (1, 3)] == [3, 1]
    cand, p = D.num_denoising_steps_probs(7)
    assert cand == [1, 3, 5, 7] and np.allclose(p, [0.4, 0.3, 0.2, 0.1])
    cand, p = D.num_denoising_steps_probs(5)
    assert cand == [1, 3, 5] and np.allclose(p, np.array([0.4, 0.3, 0.2]) / 0.9)
    t = torch.tensor([0, 500, 999])
    assert D.shift_t_for_multistep(t, 1).tolist() == [0, 500, 999]
    assert D.shift_t_for_multistep(t, 3).tolist() == [2000 // 6, 4000 // 6, (3996 + 2000) // 6]
    assert D.shift_t_for_multistep(t, 7).tolist() == [600, 800, (3996 + 6000) // 10]


def test_rollout_with_a_perfect_teacher_recovers_x0_and_walks_t_down():
    sched = O.make_schedule()
    g = torch.Generator().manual_seed(0)
    B, nd = 3, 5
    x0 = torch.randn(B, 4, 8, 8, generator=g)
    noises = [torch.randn(B, 4, 8, 8, generator=g) for _ in range(nd)]
    rel = [torch.rand(B, generator=g) for _ in range(nd - 1)]
    t = torch.tensor([900, 500, 120])
    calls = []

    def teacher(x_noisy, t_i, ctx):                    # returns the very noise that was mixed in
        i = len(calls)
        calls.append((x_noisy, t_i))
        return noises[i]

    preds, x0s, used_noises, ts = D.arc2face_rollout(teacher, sched, x0, noises[0], t, None, nd, rel, noises)
    assert len(preds) == len(x0s) == len(used_noises) == len(ts) == nd
    for p in x0s:                                       # eps exact -> predict_start_from_noise inverts q_sample
        assert torch.allclose(p, x0, atol=2e-4), float((p - x0).abs().max())
    lo, hi = 0.5 ** ((nd - 1) ** -0.3), 0.7 ** ((nd - 1) ** -0.3)
    for i in range(nd - 1):
        a, b = ts[i].float() * lo, ts[i].float() * hi
        assert torch.all(ts[i + 1] >= a.floor().long()) and torch.all(ts[i + 1] <= b.long())
        assert torch.equal(ts[i + 1], ((b - a) * rel[i] + a).long())
        assert torch.equal(used_noises[i + 1], noises[i + 1])
    assert torch.equal(calls[2][1], ts[2])


def test_distill_loss_indexing_normalisation_and_skipped_steps():
    sched = O.make_schedule()
    g = torch.Generator().manual_seed(1)
    for B, nd, want_start in ((2, 3, 0), (2, 5, 2), (1, 7, 0), (4, 1, 0)):
        noise_preds = [torch.randn(B, 4, 8, 8, generator=g) for _ in range(nd)]
        pred_x0s = [torch.randn(B, 4, 8, 8, generator=g) for _ in range(nd)]
        noises = [torch.randn(B, 4, 8, 8, generator=g) for _ in range(nd)]
        ts = [torch.randint(0, 1000, (B,), generator=g) for _ in range(nd)]
        seen = []

        def student(x_noisy, t2):
            seen.append((x_noisy, t2))
            return torch.full_like(x_noisy, 0.0)

        loss, losses, outs, start = D.arc2face_distill_loss(student, sched, (noise_preds, pred_x0s, noises, ts),
                                                            None, None, nd)
        assert start == want_start and len(outs) == len(losses) == nd - start
        for j, s in enumerate(range(start, nd)):
            want_in = O.q_sample(sched, pred_x0s[s - 1], ts[s], noises[s])     # s = 0 -> pred_x0s[-1], literally
            assert torch.equal(seen[j][0], want_in) and torch.equal(seen[j][1], ts[s])
            # masks None -> plain mean of squares of (0 - teacher eps)
            assert abs(float(losses[j]) - float((noise_preds[s] ** 2).mean())) < 1e-6
        assert abs(float(loss) - sum(float(l) for l in losses) / math.sqrt(nd)) < 1e-6

"""Rows a7 / a8 with the REFERENCE's own hook: ``adaface.subj_basis_generator.SubjBasisGenerator`` (background path, which
runs in the build container) is called through this package's boundary code -- ``hook_bridge`` and ``LatentDiffusion``'s
conditioning assembly -- and must give what a direct call of the reference class gave when the fixture was captured
(tests/golden/make_golden_hook.py): output, gradient into its 2.8 M parameters for a fixed upstream gradient, gradient into
the input features.  Needs /root/reference (skipped on the GPU box; the fixture itself is checked there for integrity)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FIX = os.path.join(ROOT, "tests", "golden", "hook_bg_sbg.npz")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "adaface")), reason="no reference checkout on this box")

_CHILD = r'''
import sys, types, json
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r)
import numpy as np, torch
sys.path.insert(0, %(root)r + "/tests/golden")
import make_golden_hook as G                     # build_reference_bg_sbg: the same seeded construction as the fixture
hook = G.build_reference_bg_sbg()
hook.train()
fix = np.load(%(fix)r)
gen = torch.Generator().manual_seed(int(fix["input_seed"]))
feats = (torch.randn(2, 257, 1024, generator=gen) * 0.5).requires_grad_(True)
U = torch.randn(2, 16, 4, 768, generator=gen)
assert np.allclose(G.sample(feats, 256).numpy(), fix["clip_features_sample"]) and np.allclose(G.sample(U, 256).numpy(), fix["upstream_sample"])

from adaprompt_amd import hook_bridge as HB
res = {}
# ---- (1) the bridge: hook -> context -> upstream gradient placed at the subject rows of the context
base = torch.randn(16, 77, 768, generator=torch.Generator().manual_seed(3)) * 0.05
cond_fn = HB.make_cond_fn_from_reference_hook(hook, base, token_start=24, is_face=False)
ctx, _, extra = cond_fn({"zs_clip_features": feats})
assert tuple(ctx.shape) == (32, 77, 768)
got = ctx.view(2, 16, 77, 768)[:, :, 24:28]
res["out_err"] = float((got.detach() - torch.from_numpy(fix["out"])).norm() / np.linalg.norm(fix["out"]))
res["untouched_rows_equal_base"] = bool(torch.equal(ctx.view(2, 16, 77, 768)[:, :, :24], base[None, :, :24].expand(2, -1, -1, -1)))
res["subj_indices"] = [extra["subj_indices"][0].tolist(), extra["subj_indices"][1].tolist()]
gctx = torch.zeros(2, 16, 77, 768)
gctx[:, :, 24:28] = U                              # what the UNet's backward would deliver at those rows
gctx[:, :, :24] = 7.0                              # gradient at frozen rows must not reach the hook
ctx.backward(gctx.view(32, 77, 768))
errs, worst = [], 0.0
for n, p in hook.named_parameters():
    if p.grad is None:
        continue
    want_norm, want_s = float(fix["gnorm/" + n]), fix["gsamp/" + n]
    e = abs(float(p.grad.double().norm()) - want_norm) / (want_norm + 1e-30)
    es = float(np.abs(G.sample(p.grad).numpy() - want_s).max() / (np.abs(want_s).max() + 1e-30))
    worst = max(worst, e, es)
res["param_grad_worst_rel"] = worst
res["n_grad_params"] = sum(1 for _, p in hook.named_parameters() if p.grad is not None)
res["feat_grad_err"] = float(abs(float(feats.grad.double().norm()) - float(fix["grad_clip_features_norm"])) / float(fix["grad_clip_features_norm"]))
# ---- (2) optimiser groups in the embedding manager's shape
groups = HB.hook_optimized_parameters(hook)
res["groups"] = [[len(g["params"]), g["lr_ratio"], g["excluded_from_prodigy"]] for g in groups]
res["n_params"] = sum(p.numel() for g in groups for p in g["params"])
print("HOOKRESULT " + json.dumps(res))
'''


def test_fixture_integrity():
    z = np.load(FIX)
    assert z["out"].shape == (2, 16, 4, 768) and int(z["n_params"]) == 2811648
    assert len(z["param_names"]) == 18 and all(("gnorm/" + n) in z.files for n in z["param_names"])
    assert np.isfinite(z["out"]).all() and float(np.linalg.norm(z["out"])) > 1.0


@needs_ref
def test_reference_bg_hook_through_the_bridge_matches_the_direct_call():
    out = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT, "fix": FIX}], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("HOOKRESULT ")][0].split(" ", 1)[1])
    assert res["out_err"] < 1e-6, res
    assert res["untouched_rows_equal_base"]
    assert res["subj_indices"][0] == [0] * 4 + [1] * 4 and res["subj_indices"][1] == [24, 25, 26, 27] * 2
    assert res["n_grad_params"] == 18 and res["param_grad_worst_rel"] < 1e-5, res
    assert res["feat_grad_err"] < 1e-6, res
    assert res["groups"] == [[res["groups"][0][0], 1, False]] and res["n_params"] == 2811648

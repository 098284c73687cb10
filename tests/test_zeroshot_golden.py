"""The zero-shot feature front end's host side: ``encode_zero_shot_image_features`` of the mirror against
the reference's own method (ddpm.py:2322-2471) run on the same fake encoders (tests/golden/make_golden_zeroshot.py)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_zeroshot as Z          # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "zeroshot_frontend.npz"))


@pytest.mark.parametrize("case", Z.CASES, ids=[c["name"] for c in Z.CASES])
def test_encode_zero_shot_image_features_matches_the_reference_method(case):
    from adaprompt_amd.ldm.models.diffusion.conditioning import ConditioningMixin
    obj = types.SimpleNamespace(device=torch.device("cpu"))
    Z.attach_fakes(obj, case)
    img, masks, paths = Z.case_args(case)
    torch.manual_seed(100 + case["seed"])
    feats, ids, faceless = ConditioningMixin.encode_zero_shot_image_features(obj, img, masks, image_paths=paths, **case["kw"])
    n = case["name"]
    assert faceless == int(GOLD[n + ".faceless"])
    assert repr(obj.clip_image_encoder.calls) == str(GOLD[n + ".encoder_calls"])
    assert torch.allclose(feats, torch.from_numpy(GOLD[n + ".clip_features"]), atol=1e-6, rtol=1e-6)
    if ids is None:
        assert n + ".id_embs" not in GOLD.files
    else:
        assert torch.allclose(ids, torch.from_numpy(GOLD[n + ".id_embs"]), atol=1e-6, rtol=1e-6)


def test_front_end_needs_encoders_handed_over():
    from adaprompt_amd.ldm.models.diffusion.conditioning import ConditioningMixin
    obj = types.SimpleNamespace(device=torch.device("cpu"), zs_image_encoders_instantiated=False,
                                instantiate_zero_shot_image_encoders=lambda: ConditioningMixin.instantiate_zero_shot_image_encoders(obj))
    with pytest.raises(RuntimeError, match="set_zero_shot_image_encoders"):
        ConditioningMixin.encode_zero_shot_image_features(obj, torch.zeros(1, 3, 8, 8, dtype=torch.uint8), None)

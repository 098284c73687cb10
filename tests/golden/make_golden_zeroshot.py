"""Golden vectors of the zero-shot feature front end's host side: the reference's own
``LatentDiffusion.encode_zero_shot_image_features`` (ddpm.py:2322-2471) called unbound on a fake ``self`` whose encoders are the
deterministic fakes below -- what is pinned is the method's logic (per-image loop, face selection, faceless / skipped images,
mask resizing, the two masked encoder passes, the cached zero-image features, concatenation, averaging), not third-party
arithmetic.  The image encoder itself is pinned separately (make_golden_clip_vision.py).

    python tests/golden/make_golden_zeroshot.py        # writes tests/golden/zeroshot_frontend.npz"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "zeroshot_frontend.npz")
D_FEAT = 8


# ---- deterministic fakes of the third-party encoders (shared with tests/test_zeroshot_golden.py) -----------------------------
class FakePreprocessor:
    """stands where HF ``CLIPImageProcessor`` stands: [3,H,W] uint8 tensor or HWC array -> pixel_values [1,3,224,224]"""

    def __call__(self, images, return_tensors="pt"):
        t = torch.as_tensor(np.asarray(images.cpu() if torch.is_tensor(images) else images))
        if t.shape[-1] == 3:
            t = t.permute(2, 0, 1)
        pv = F.interpolate(t.float()[None] / 255.0, size=(224, 224), mode="bilinear", align_corners=False)
        return types.SimpleNamespace(pixel_values=(pv - 0.45) / 0.27)


class Face(dict):
    __getattr__ = dict.get


class FakeInsightFace:
    """``get(bgr)``: no face in a dark image, else two boxes -- the reference picks by its own size key"""

    def get(self, bgr):
        assert bgr.dtype == np.uint8 and bgr.shape == (512, 512, 3), (bgr.dtype, bgr.shape)
        means = bgr.reshape(-1, 3).mean(0)                         # B, G, R
        if means.mean() < 40:
            return []
        def emb(k):
            v = np.cos(np.arange(512) * (0.01 + 0.001 * k) + means[0] / 50.0) + np.sin(np.arange(512) * 0.02 * (k + 1) + means[2] / 70.0)
            return (v / np.linalg.norm(v)).astype(np.float32)
        return [Face(bbox=np.array([10., 300., 60., 400.]), normed_embedding=emb(0)),
                Face(bbox=np.array([100., 20., 300., 250.]), normed_embedding=emb(1)),
                Face(bbox=np.array([5., 5., 400., 30.]), normed_embedding=emb(2))]


class FakeClipEncoder:
    """the ``CLIPVisionModelWithMask`` contract: hidden_states[-2] [B,257,D] and attn_mask [B,257,1]"""

    def __init__(self):
        self.proj = torch.randn(3, D_FEAT, generator=torch.Generator().manual_seed(21))
        self.calls = []

    def __call__(self, pixel_values, attn_mask=None, output_hidden_states=True, **kw):
        assert pixel_values.dtype == torch.float16 and (attn_mask is None or attn_mask.dtype == torch.float16)
        self.calls.append((tuple(pixel_values.shape), None if attn_mask is None else tuple(attn_mask.shape)))
        x = pixel_values.float()
        B = x.shape[0]
        patches = F.avg_pool2d(x, 14).flatten(2).transpose(1, 2)                  # [B,256,3]
        tok = torch.cat([patches.mean(1, keepdim=True), patches], dim=1) @ self.proj + 0.1           # [B,257,D]
        tm = None
        if attn_mask is not None:
            m = F.interpolate(attn_mask.float().unsqueeze(1), size=(16, 16), mode="nearest").flatten(2)
            tm = torch.cat([torch.ones_like(m[:, :, :1]), m], dim=-1).permute(0, 2, 1)               # [B,257,1]
            tok = tok * (1 + 0.5 * tm) + tm.mean(dim=1, keepdim=True)
        return types.SimpleNamespace(hidden_states=(tok * 0, tok, tok + 1), last_hidden_state=tok + 1, attn_mask=tm)


class DinoInput(dict):
    def to(self, device):
        return self


def fake_dino_preprocess(images, return_tensors="pt"):
    t = torch.as_tensor(np.asarray(images.cpu() if torch.is_tensor(images) else images)).float()
    return DinoInput(pixel_values=t.mean().view(1, 1))


def fake_dino_encoder(pixel_values):
    h = torch.cos(torch.arange(197 * 384).view(1, 197, 384) * 1e-3 + pixel_values.view(1, 1, 1) / 40.0)
    return types.SimpleNamespace(last_hidden_state=h)


def images_case(seed, B=3, hw=64, dark=()):
    img = (torch.rand(B, 3, hw, hw, generator=torch.Generator().manual_seed(seed)) * 200 + 55).to(torch.uint8)
    for b in dark:
        img[b] = (img[b].float() * 0.1).to(torch.uint8)
    mask = (torch.rand(B, hw, hw, generator=torch.Generator().manual_seed(seed + 1)) > 0.5).float()
    return img, mask


CASES = [dict(name="faces", seed=1, kw={}), dict(name="faces_avg", seed=2, kw={"calc_avg": True}),
         dict(name="one_faceless", seed=3, dark=(1,), kw={}), dict(name="skip_non_faces", seed=4, dark=(0,), kw={"skip_non_faces": True}),
         dict(name="mask_list", seed=5, kw={}, mask_list=True), dict(name="no_mask", seed=6, kw={}, no_mask=True),
         dict(name="non_face_dino", seed=7, kw={"is_face": False, "calc_avg": True}),
         dict(name="no_insightface", seed=8, kw={}, no_insightface=True)]


def attach_fakes(obj, case):
    obj.clip_preprocessor, obj.clip_image_encoder = FakePreprocessor(), FakeClipEncoder()
    obj.insightface_app = None if case.get("no_insightface") else FakeInsightFace()
    obj.dino_preprocess, obj.dino_encoder = fake_dino_preprocess, fake_dino_encoder
    obj.neg_image_features = None
    obj.zs_image_encoders_instantiated = True


def case_args(case):
    img, mask = images_case(case["seed"], dark=case.get("dark", ()))
    masks = mask
    if case.get("mask_list"):
        masks = [mask[0].numpy(), F.interpolate(mask[1][None, None], size=(48, 80))[0, 0].numpy(), mask[2].numpy()]
    if case.get("no_mask"):
        masks = None
    return img, masks, [f"/data/{i}.jpg" for i in range(img.shape[0])]


def main():
    sys.path.insert(0, HERE)
    import make_golden_ddpm as G
    D = G.import_reference_ddpm()
    D.cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])       # the one cv2 call on this path (ddpm.py:2349)
    D.cv2.COLOR_RGB2BGR = 4
    rec = {}
    for case in CASES:
        fake = types.SimpleNamespace(device=torch.device("cpu"))
        attach_fakes(fake, case)
        img, masks, paths = case_args(case)
        torch.manual_seed(100 + case["seed"])                      # the faceless image's random embedding
        feats, ids, faceless = D.LatentDiffusion.encode_zero_shot_image_features(fake, img, masks, image_paths=paths, **case["kw"])
        rec[case["name"] + ".clip_features"] = feats.numpy()
        if ids is not None:
            rec[case["name"] + ".id_embs"] = ids.numpy()
        rec[case["name"] + ".faceless"] = np.int64(faceless)
        rec[case["name"] + ".encoder_calls"] = np.array(repr(fake.clip_image_encoder.calls))
        print(case["name"], tuple(feats.shape), None if ids is None else tuple(ids.shape), faceless)
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()

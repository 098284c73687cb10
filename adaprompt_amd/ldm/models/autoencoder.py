"""``AutoencoderKL`` first stage (reference ldm/models/autoencoder.py:285-328): ``encode`` on the
MI355X kernels.  ``encode(x, mask) -> DiagonalGaussianDistribution`` keeps the reference
signature; ``encode_moments_nhwc`` is the pixel-major fast path the training step uses.
The decoder (used only by CLIP filtering / inference) is a "next" row of SURVEY.md 8(f) and is not
built: ``decode`` raises."""
import torch
import torch.nn as nn

from ... import functional as HF
from ... import ops
from ..modules.diffusionmodules.model import Encoder
from ..modules.distributions.distributions import DiagonalGaussianDistribution


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        if monitor is not None:
            self.monitor = monitor
        self._wc = HF.WeightCache()
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    @torch.no_grad()
    def encode_moments_nhwc(self, x_hwc, mask=None):
        """x [B,H,W,3] f32 in [-1,1] -> moments pixel-major [B,H/8,W/8,2*embed_dim] f32."""
        h = self.encoder.forward_nhwc(x_hwc, mask)
        qc = self._wc.get("quant_conv", self.quant_conv.weight, self.quant_conv.bias)
        B, H, W, C = h.shape
        m, _ = ops.linear(h.view(B, H * W, C), qc.fwd, qc.O4, bias=qc.bias)
        return m.view(B, H, W, -1)

    def encode(self, x, mask=None):
        """reference signature: x NCHW -> posterior over NCHW moments."""
        moments = self.encode_moments_nhwc(x.permute(0, 2, 3, 1), mask).permute(0, 3, 1, 2)
        return DiagonalGaussianDistribution(moments)

    def decode(self, z):
        raise NotImplementedError("VAE decoder is a 'next' row (SURVEY.md 8f-2): not on the training hot path")

"""``AutoencoderKL`` first stage (reference ldm/models/autoencoder.py:285-328): ``encode`` on the
MI355X kernels.  ``encode(x, mask) -> DiagonalGaussianDistribution`` keeps the reference
signature; ``encode_moments_nhwc`` is the pixel-major fast path the training step uses.
``decode(z)`` (autoencoder.py:330-333: post_quant_conv then the Decoder) serves the inference path
(``decode_first_stage`` in the sampler scripts); the decoder is built only when ``with_decoder`` is set or a
decoder checkpoint is loaded, so the training replica does not carry its 49.5 M parameters."""
import torch
import torch.nn as nn

from ... import functional as HF
from ... import ops
from ..modules.diffusionmodules.model import Decoder, Encoder
from ..modules.distributions.distributions import DiagonalGaussianDistribution


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, with_decoder=False):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.ddconfig = dict(ddconfig)
        self.decoder = Decoder(**ddconfig) if with_decoder else None
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

    def build_decoder(self):
        """instantiate the decoder (same ddconfig) after construction, e.g. before loading a full VAE checkpoint."""
        if self.decoder is None:
            self.decoder = Decoder(**self.ddconfig).to(self.quant_conv.weight.device)
        return self.decoder

    @torch.no_grad()
    def decode_nhwc(self, z_hwc):
        """z pixel-major [B,h,w,embed_dim] -> image pixel-major [B,8h,8w,3]."""
        if self.decoder is None:
            raise RuntimeError("AutoencoderKL was built without its decoder: pass with_decoder=True or call build_decoder()")
        if not z_hwc.is_cuda:
            raise RuntimeError("adaprompt_amd AutoencoderKL.decode runs on the MI355X HIP kernels only; got a CPU tensor")
        B, H, W, C = z_hwc.shape
        # the kernels address one operand through a 32-bit buffer descriptor (< 2 GiB): the widest decoder
        # activation is [b, 8H, 8W, 2*ch] f32, so large batches are decoded in slices
        widest = 64 * H * W * 2 * self.ddconfig["ch"] * 4
        per = max(1, int(((1 << 31) - 1) // widest))
        if B > per:
            return torch.cat([self.decode_nhwc(z_hwc[i:i + per]) for i in range(0, B, per)], dim=0)
        pq = self._wc.get("post_quant_conv", self.post_quant_conv.weight, self.post_quant_conv.bias)
        z16 = ops.pad_cast_bf16(z_hwc.contiguous().float(), pq.I8)
        q, _ = ops.linear(z16.view(B, H * W, -1), pq.fwd, pq.O4, bias=pq.bias)
        q = q.view(B, H, W, -1)[..., :self.post_quant_conv.out_channels]
        return self.decoder.forward_nhwc(q)

    def decode(self, z):
        """reference signature (autoencoder.py:330-333): z NCHW -> image NCHW."""
        return self.decode_nhwc(z.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

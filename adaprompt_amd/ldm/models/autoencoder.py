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

    # ---- the same encode through the single C entry adap_vae_encode (vae_encode(x, masks, weights*, noise, z))
    def _c_abi_table(self):
        """(cfg ints, device-pointer table, the tensors kept alive) in the order include/adaprompt_hip.h documents."""
        import ctypes
        enc = self.encoder
        keep, ptrs = [], []

        def conv(mod_wc, key, m, fused=None):
            pk = mod_wc.get(key, m.weight, m.bias) if fused is None else mod_wc.get(key, *fused)
            keep.extend([pk.fwd, pk.bias])
            ptrs.extend([pk.fwd.data_ptr(), pk.bias.data_ptr()])

        def norm(m):
            keep.extend([m.weight, m.bias])
            ptrs.extend([m.weight.data_ptr(), m.bias.data_ptr()])

        def res(b):
            norm(b.norm1)
            conv(b._wc, "conv1", b.conv1)
            norm(b.norm2)
            conv(b._wc, "conv2", b.conv2)
            if b.in_channels != b.out_channels:
                conv(b._wc, "nin", b.nin_shortcut)

        conv(enc._wc, "conv_in", enc.conv_in)
        for i, lvl in enumerate(enc.down):
            for b in lvl.block:
                res(b)
            if i != enc.num_resolutions - 1:
                conv(lvl.downsample._wc, "conv", lvl.downsample.conv)
        res(enc.mid.block_1)
        at = enc.mid.attn_1
        norm(at.norm)
        conv(at._wc, "qkv", None, fused=([at.q.weight, at.k.weight, at.v.weight], [at.q.bias, at.k.bias, at.v.bias]))
        conv(at._wc, "proj_out", at.proj_out)
        res(enc.mid.block_2)
        norm(enc.norm_out)
        conv(enc._wc, "conv_out", enc.conv_out)
        conv(self._wc, "quant_conv", self.quant_conv)
        dd = self.ddconfig
        cfg = [dd["ch"], len(dd["ch_mult"]), dd["num_res_blocks"], enc.conv_out.out_channels, self.quant_conv.out_channels] + \
            [int(m) for m in dd["ch_mult"]]
        return (ctypes.c_int * len(cfg))(*cfg), (ctypes.c_void_p * len(ptrs))(*ptrs), len(ptrs), keep

    @torch.no_grad()
    def encode_c_abi(self, x_hwc, mask=None, noise=None, scale=1.0):
        """``adap_vae_encode``: x [B,H,W,3] f32 -> (moments [B,h,w,2z], z [B,h,w,z] or None) in ONE C-ABI call -- the same
        launches ``encode_moments_nhwc`` + ``ops.posterior_sample`` issue from Python, issued by the library."""
        import ctypes
        from ... import _lib
        from ...ops import _stream, gn_sync_buffer
        from ..modules.diffusionmodules.model import AttnBlock
        if not x_hwc.is_cuda:
            raise RuntimeError("adaprompt_amd AutoencoderKL.encode runs on the MI355X HIP kernels only; got a CPU tensor")
        x = x_hwc.contiguous().float()
        B, H, W, _ = x.shape
        cfg, table, n, keep = self._c_abi_table()
        f = 2 ** (self.encoder.num_resolutions - 1)
        h, w = H // f, W // f
        cls = AttnBlock.pixel_classes(mask, (h, w), x)
        e2 = self.quant_conv.out_channels
        moments = torch.empty(B, h, w, e2, device=x.device)
        z = None
        if noise is not None:
            noise = noise.contiguous().float()
            assert tuple(noise.shape) == (B, h, w, e2 // 2), noise.shape
            z = torch.empty_like(noise)
        nbytes = _lib.call_long("adap_vae_encode_workspace_bytes", ctypes.addressof(cfg), B, H, W)
        ws = torch.empty(nbytes + 256, device=x.device, dtype=torch.uint8)
        base = (ws.data_ptr() + 255) // 256 * 256
        _lib.call("adap_vae_encode", ctypes.addressof(cfg), ctypes.addressof(table), n, x.data_ptr(),
                  0 if cls is None else cls.data_ptr(), 0 if noise is None else noise.data_ptr(), float(scale), moments.data_ptr(),
                  0 if z is None else z.data_ptr(), base, nbytes, gn_sync_buffer(x.device), B, H, W, _stream())
        del keep
        return moments, z

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

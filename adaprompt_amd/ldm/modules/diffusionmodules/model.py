"""VAE ``Encoder`` (reference ldm/modules/diffusionmodules/model.py:408-499) and ``Decoder`` (:502-608) of the
first stage on the MI355X kernels, with ResnetBlock :83-142, Downsample :61-80, Upsample :42-58,
AttnBlock :151-242, Normalize :39-40.  Parameter names follow the reference
(``first_stage_model.encoder.*`` checkpoints load by key).  Forward only: the first stage is
frozen and runs under no_grad on the training path (ddpm.py:1381-1419).

Activations are pixel-major; the 512x512x3 image is consumed in the dataloader's own HWC layout
(ddpm.py:480-481 permutes it to CHW for the reference -- here that permute is the identity).
The mid AttnBlock (single head, N=4096, C=512) is two batched contractions around a softmax
kernel that applies the reference's POST-softmax zero fill of fg/bg hetero pairs."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import functional as HF
from .... import ops


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Downsample(nn.Module):
    """F.pad(x, (0,1,0,1)) + conv3x3 stride 2 pad 0 == zero-predicated taps with pad 0 and Hout = H/2."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)
        self._wc = HF.WeightCache()

    def forward(self, x):
        """x f32 or -- from the level's last ResnetBlock (``out_bf16_only``) -- bf16: the contraction's operand is bf16 either way
        (an f32 input is converted in registers with the same rounding), so the bf16 hand-over is the same arithmetic with half
        the bytes written by the producer and read here (537 MB -> 268 MB each way at 512 x 512, bs 4)."""
        B, H, W, C = x.shape
        pk = self._wc.get("conv", self.conv.weight, self.conv.bias)
        # (its output is the next level's first GroupNorm input: statistics from the epilogue when it is large, as in ResnetBlock)
        y, _ = ops.conv2d(x, pk.fwd, pk.O4, 3, 2, 0, out_hw=(H // 2, W // 2), bias=pk.bias,
                          gn_stats=gn_stats_from_epilogue(B, H // 2, W // 2, C))
        return y


DOWNSAMPLE_BF16 = __import__("os").environ.get("ADAP_VAE_DOWNSAMPLE_BF16", "1") != "0"       # A/B switch (csrc/vae.hip reads the same)
_GN_EPI_MIN_LOG2 = int(__import__("os").environ.get("ADAP_GN_EPILOGUE_MIN_LOG2", "23"))      # (tuning; csrc/vae.hip reads the same)


def gn_stats_from_epilogue(B, H, W, C):
    """A conv output this large (the 128^2 ... 512^2 levels: 64-537 MB) gets its GroupNorm statistics from the conv's own epilogue
    (``ops.conv2d(gn_stats=True)``: the records ride on the output tensor and ``groupnorm_fwd`` skips its statistics pass); below it
    the pass is cheap and the records' finish launch is not.  The same rule as csrc/vae.hip ``stats_from_epilogue``, so that
    ``adap_vae_encode`` and this mirror stay bit-identical."""
    return B * H * W * C >= (1 << _GN_EPI_MIN_LOG2) and (H * W) % 256 == 0 and C % 32 == 0


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        assert not conv_shortcut and temb_channels == 0
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)
        self._wc = HF.WeightCache()

    def forward(self, x, temb=None, out_bf16_only=False):
        """``out_bf16_only``: the block's output is read by a Downsample's contraction alone (the encoder's last block of a
        level): it is written as the bf16 operand that contraction would make of it anyway, and not as f32."""
        assert temb is None
        wc = self._wc
        c1 = wc.get("conv1", self.conv1.weight, self.conv1.bias)
        c2 = wc.get("conv2", self.conv2.weight, self.conv2.bias)
        B, H, W, _ = x.shape
        big = gn_stats_from_epilogue(B, H, W, self.out_channels)
        _, a1, _, _ = ops.groupnorm_fwd(x, self.norm1.weight, self.norm1.bias, 1e-6, 1)
        # the tensor between conv1 and norm2 is block-internal: bf16
        _, h = ops.conv2d(a1, c1.fwd, c1.O4, 3, 1, 1, bias=c1.bias, out_f32=False, out_bf16=True, gn_stats=big)
        _, a2, _, _ = ops.groupnorm_fwd(h, self.norm2.weight, self.norm2.bias, 1e-6, 1)
        if self.in_channels != self.out_channels:
            sk = wc.get("nin", self.nin_shortcut.weight, self.nin_shortcut.bias)
            skip, _ = ops.conv2d(x, sk.fwd, sk.O4, 1, bias=sk.bias)
        else:
            skip = x
        if out_bf16_only:
            _, y = ops.conv2d(a2, c2.fwd, c2.O4, 3, 1, 1, bias=c2.bias, residual=skip, out_f32=False, out_bf16=True)
            return y
        y, _ = ops.conv2d(a2, c2.fwd, c2.O4, 3, 1, 1, bias=c2.bias, residual=skip, gn_stats=big)
        return y


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self._wc = HF.WeightCache()

    @staticmethod
    def pixel_classes(mask, hw, like):
        """mask dict {'fg_mask','aug_mask'} [B,1,H,W] -> uint8 [B, h*w]: 0 outside aug, 1 fg, 2 bg (model.py:196-232)."""
        if mask is None or mask.get("fg_mask") is None:
            return None
        B = like.shape[0]
        fg = mask["fg_mask"].detach().float().contiguous()
        aug = mask["aug_mask"]
        aug = None if aug is None else aug.detach().float().contiguous()
        if fg.dim() != 4 or fg.shape[0] != B or fg.shape[1] != 1 or (aug is not None and aug.shape[:2] != fg.shape[:2]):
            raise ValueError(f"fg_mask / aug_mask must be [B,1,H,W], got {tuple(fg.shape)}")
        cls = torch.empty(B, hw[0] * hw[1], device=like.device, dtype=torch.uint8)
        ops._lib.call("adap_pixel_classes", fg.data_ptr(), fg.shape[2], fg.shape[3], 0 if aug is None else aug.data_ptr(),
                      0 if aug is None else aug.shape[2], 0 if aug is None else aug.shape[3], cls.data_ptr(), B, hw[0], hw[1],
                      ops._stream())
        return cls

    def forward(self, x, mask=None):
        B, H, W, C = x.shape
        N = H * W
        wc = self._wc
        _, hn, _, _ = ops.groupnorm_fwd(x, self.norm.weight, self.norm.bias, 1e-6, 0)
        qkv = wc.get("qkv", [self.q.weight, self.k.weight, self.v.weight], [self.q.bias, self.k.bias, self.v.bias])
        _, y = ops.linear(hn.view(B, N, C), qkv.fwd, 3 * C, bias=qkv.bias, out_f32=False, out_bf16=True)
        q = y[..., :C].contiguous()
        k = y[..., C:2 * C].contiguous()
        vT = ops.transpose_bf16(y[..., 2 * C:].contiguous())                 # [B, C, N]
        S = ops.batched_matmul_nt(q, k)                                       # [B, N, N] f32, unscaled
        P = ops.vae_softmax(S, float(int(C) ** -0.5), self.pixel_classes(mask, (H, W), x))
        o = ops.batched_matmul_nt(P, vT, out_dtype=torch.bfloat16)           # [B, N, C]
        po = wc.get("proj_out", self.proj_out.weight, self.proj_out.bias)
        out, _ = ops.linear(o, po.fwd, C, bias=po.bias, residual=x.view(B, N, C))
        return out.view(B, H, W, C)


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type == "vanilla"
    return AttnBlock(in_channels)


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert not use_linear_attn and len(list(attn_resolutions)) == 0, "SD-1.5 VAE: attention only in mid"
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            down = nn.Module()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1,
                                  padding=1)
        self._wc = HF.WeightCache()

    def forward_nhwc(self, x_hwc, mask=None):
        """x_hwc [B,H,W,3] f32 (the dataloader's layout) -> pixel-major h [B,H/8,W/8,2*z]."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and x_hwc.requires_grad:
            raise NotImplementedError("the first stage is frozen: encode runs under no_grad (ddpm.py:1381-1419)")
        if not x_hwc.is_cuda:
            raise RuntimeError("adaprompt_amd Encoder runs on the MI355X HIP kernels only; got a CPU tensor")
        wc = self._wc
        cin = wc.get("conv_in", self.conv_in.weight, self.conv_in.bias)
        xf = x_hwc.contiguous().float()
        if cin.O4 in (32, 64, 128) and cin.I == 3:
            # the 3 x 3 x 3 patch as one K step, straight from the f32 image (adap_conv3x3_rgb; the same rule in csrc/vae.hip)
            h = ops.conv3x3_rgb(xf, cin.fwd, cin.O4, bias=cin.bias, gn_stats=gn_stats_from_epilogue(*xf.shape[:3], cin.O4))
        else:
            x16 = ops.pad_cast_bf16(xf, cin.I8)
            h, _ = ops.conv2d(x16, cin.fwd, cin.O4, 3, 1, 1, bias=cin.bias, gn_stats=gn_stats_from_epilogue(*x16.shape[:3], cin.O4))
        for i_level in range(self.num_resolutions):
            last = i_level == self.num_resolutions - 1
            blocks = list(self.down[i_level].block)
            for j, blk in enumerate(blocks):
                # (a level's last block feeds the Downsample alone: handed over as that contraction's bf16 operand)
                h = blk(h, out_bf16_only=DOWNSAMPLE_BF16 and not last and j == len(blocks) - 1)
            if not last:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h, mask)
        h = self.mid.block_2(h)
        _, a, _, _ = ops.groupnorm_fwd(h, self.norm_out.weight, self.norm_out.bias, 1e-6, 1)
        co = wc.get("conv_out", self.conv_out.weight, self.conv_out.bias)
        y, _ = ops.conv2d(a, co.fwd, co.O4, 3, 1, 1, bias=co.bias)
        return y

    def forward(self, x, mask=None):
        """reference signature: x NCHW -> NCHW (a view of the pixel-major result)."""
        with torch.no_grad():
            return self.forward_nhwc(x.permute(0, 2, 3, 1), mask).permute(0, 3, 1, 2)


UPSAMPLE_BF16 = os.environ.get("ADAP_VAE_UPSAMPLE_BF16", "1") != "0"       # A/B switch


class Upsample(nn.Module):
    """nearest x2 then conv3x3 (model.py:42-58).  The x2 copy is made once, as the conv's bf16 operand (``ops.upsample2x_bf16``), so
    that the conv takes the stencil-window kernel with its statistics epilogue; folded into the conv's gather (``up=1``: every tap
    re-reads and re-converts the f32 source) the three decoder Upsamples were the slowest launches of a decode (483 us average)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self._wc = HF.WeightCache()

    def forward(self, x):
        pk = self._wc.get("conv", self.conv.weight, self.conv.bias)
        B, H, W, C = x.shape
        if x.is_cuda and x.dtype == torch.float32 and not HF.F32_STORAGE and UPSAMPLE_BF16:
            y, _ = ops.conv2d(ops.upsample2x_bf16(x), pk.fwd, pk.O4, 3, 1, 1, bias=pk.bias,
                              gn_stats=gn_stats_from_epilogue(B, 2 * H, 2 * W, C))
            return y
        y, _ = ops.conv2d(x, pk.fwd, pk.O4, 3, 1, 1, up=1, bias=pk.bias, gn_stats=gn_stats_from_epilogue(B, 2 * H, 2 * W, C))
        return y


class Decoder(nn.Module):
    """model.py:502-608.  SD-1.5: z [B,4,64,64] -> conv_in 4->512, mid (res, attn, res), four levels of three
    ResnetBlocks (512,512,256,128) with nearest-x2 + conv between them, GroupNorm + swish + conv_out 128->3.
    Inference only (``decode_first_stage`` runs under no_grad in the sampler scripts)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert not use_linear_attn and len(list(attn_resolutions)) == 0, "SD-1.5 VAE: attention only in mid"
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.give_pre_end = give_pre_end
        self.tanh_out = tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            up = nn.Module()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)          # prepend to get consistent order
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)
        self._wc = HF.WeightCache()

    def forward_nhwc(self, z_hwc):
        """z pixel-major [B,h,w,z_channels] f32 -> image pixel-major [B,8h,8w,out_ch] f32."""
        if not z_hwc.is_cuda:
            raise RuntimeError("adaprompt_amd Decoder runs on the MI355X HIP kernels only; got a CPU tensor")
        wc = self._wc
        cin = wc.get("conv_in", self.conv_in.weight, self.conv_in.bias)
        z16 = ops.pad_cast_bf16(z_hwc.contiguous().float(), cin.I8)
        h, _ = ops.conv2d(z16, cin.fwd, cin.O4, 3, 1, 1, bias=cin.bias)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        for i_level in reversed(range(self.num_resolutions)):
            for blk in self.up[i_level].block:
                h = blk(h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        if self.give_pre_end:
            return h
        _, a, _, _ = ops.groupnorm_fwd(h, self.norm_out.weight, self.norm_out.bias, 1e-6, 1)
        co = wc.get("conv_out", self.conv_out.weight, self.conv_out.bias)
        y, _ = ops.conv2d(a, co.fwd, co.O4, 3, 1, 1, bias=co.bias)
        y = y[..., :self.conv_out.out_channels]
        if self.tanh_out:
            y = torch.tanh(y)
        return y

    def forward(self, z):
        """reference signature: z NCHW -> image NCHW (a view of the pixel-major result)."""
        with torch.no_grad():
            return self.forward_nhwc(z.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

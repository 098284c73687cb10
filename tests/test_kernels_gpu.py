"""Per-kernel parity on the MI355X: every C-ABI entry point against a plain PyTorch fp32
reference of the same op (run on the same device), on seeded inputs.  Inputs of the bf16
matrix-core kernels are pre-rounded to bf16 so the only difference is accumulation order."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from adaprompt_amd import ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def bf(x):
    return x.to(torch.bfloat16).float()


def rel(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x):     # [B,C,H,W] -> [B,H,W,C] contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------------------------
# conv / linear
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, Cin, Cout, H, W, K, stride, pad, up, x bf16?
    (2, 320, 320, 16, 16, 3, 1, 1, 0, True),
    (2, 320, 320, 16, 16, 3, 1, 1, 0, False),
    (1, 640, 320, 8, 8, 3, 1, 1, 0, True),
    (2, 128, 128, 20, 12, 3, 1, 1, 0, True),       # non-square, Cout tile 128
    (2, 64, 256, 9, 7, 3, 1, 1, 0, False),         # ragged pixel tile
    (2, 8, 320, 16, 16, 3, 1, 1, 0, True),         # padded 4->8 input channels
    (2, 320, 4, 16, 16, 3, 1, 1, 0, True),         # final conv 320->4
    (2, 320, 320, 16, 16, 3, 2, 1, 0, False),      # UNet downsample
    (2, 128, 128, 16, 16, 3, 2, 0, 0, False),      # VAE downsample: pad (0,1,0,1)
    (2, 320, 320, 8, 8, 3, 1, 1, 1, False),        # nearest x2 upsample fused
    (2, 960, 640, 8, 8, 1, 1, 0, 0, False),        # 1x1 skip
    (1, 2560, 1280, 8, 8, 3, 1, 1, 0, True),       # long K
    # the 256-pixel, 3-stage LDS-DMA variant (bf16 activations, M >= 4096)
    (2, 320, 320, 64, 64, 3, 1, 1, 0, True),       # UNet 64x64 level, split-K 2
    (1, 128, 128, 128, 128, 3, 1, 1, 0, True),     # VAE-like, Cout tile 128
    (1, 64, 256, 70, 66, 3, 1, 1, 0, True),        # ragged pixel tile (4620 pixels), short K
    (4, 640, 640, 32, 32, 3, 1, 1, 0, True),       # 32x32 level: split-K 4
    (2, 320, 2560, 64, 64, 1, 1, 0, 0, True),      # GEGLU projection shape
    (4, 128, 128, 64, 64, 3, 2, 0, 0, True),       # stride 2, asymmetric pad, big tile
    (2, 320, 320, 32, 32, 3, 1, 1, 1, True),       # fused nearest x2 upsample, big tile
    # stencil-window kernel, 16 x 16 patches (rows too short for the 8 x 32 patch): ping-pong 128-wide tile + split-K
    (4, 1280, 1280, 16, 16, 3, 1, 1, 0, True),     # UNet 16x16 level
    (1, 128, 256, 32, 16, 3, 1, 1, 0, True),       # two square patches stacked, no split
    (2, 2560, 1280, 16, 16, 3, 1, 1, 0, True),     # concat input, long K
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case):
    B, Cin, Cout, H, W, K, stride, pad, up, xb = case
    x = bf(rnd(B, Cin, H, W, seed=1))
    w = bf(rnd(Cout, Cin, K, K, seed=2, scale=(Cin * K * K) ** -0.5))
    bias = rnd(Cout, seed=3)
    xr = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    if stride == 2 and pad == 0:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), w, bias, stride=2)
        out_hw = (H // 2, W // 2)
    else:
        ref = F.conv2d(xr, w, bias, stride=stride, padding=pad)
        out_hw = None
    pk = ops.PackedConv(w, bias)
    xin = nhwc(x)
    if xb:
        xin = xin.to(torch.bfloat16)
    y32, y16 = ops.conv2d(xin, pk.fwd, pk.O4, K, stride, pad, up, out_hw=out_hw, bias=pk.bias, out_f32=True,
                          out_bf16=True)
    got = nchw(y32)[:, :Cout]
    assert got.shape == ref.shape
    assert rel(got, ref) < 2e-5, rel(got, ref)
    assert rel(nchw(y16.float())[:, :Cout], ref) < 4e-3


@pytest.mark.parametrize("B,H,W,Cout", [(2, 64, 64, 128), (1, 24, 40, 64), (3, 16, 16, 32), (1, 512, 512, 128)])
def test_conv3x3_rgb_single_k_step(B, H, W, Cout):
    """``adap_conv3x3_rgb`` (the VAE encoder's conv_in, model.py:426, 468: the 3 x 3 x 3 patch as ONE K step, straight from the f32
    image) against F.conv2d on the bf16-rounded operands, and against the general path (pad 3 -> 8 channels, nine K steps): same
    products, another summation order.  With Cout = 128 the epilogue's GroupNorm statistics records against the output's sums."""
    x = rnd(B, H, W, 3, seed=1)
    w = rnd(Cout, 3, 3, 3, seed=2, scale=27 ** -0.5)
    bias = rnd(Cout, seed=3)
    pk = ops.PackedConv(w, bias)
    stats = Cout == 128 and (H * W) % 256 == 0
    y = ops.conv3x3_rgb(x, pk.fwd, Cout, bias=pk.bias, gn_stats=stats)
    ref = F.conv2d(bf(x).permute(0, 3, 1, 2), bf(w), bias, padding=1).permute(0, 2, 3, 1)
    assert rel(y, ref) < 2e-5, rel(y, ref)
    y_gen, _ = ops.conv2d(ops.pad_cast_bf16(x, 8), pk.fwd, Cout, 3, 1, 1, bias=pk.bias)
    assert rel(y, y_gen) < 2e-6
    if stats:
        part, chunks = getattr(y, ops.GN_STATS_ATTR)[:2]
        assert chunks == H * W // 64
        g = y.double().view(B, H * W // 64, 64, 32, Cout // 32)
        want = torch.stack([g.sum(dim=(2, 4)), (g * g).sum(dim=(2, 4))], dim=-1)
        assert rel(part.double(), want) < 1e-5
    else:
        assert getattr(y, ops.GN_STATS_ATTR, None) is None


def test_upsample2x_bf16_and_upsample_conv_paths_agree():
    """``adap_upsample2x_bf16`` = F.interpolate(nearest, x2) + bf16 rounding, exactly; and Upsample's conv on that image (the
    stencil-window kernel) against the fused-gather form ``conv2d(up=1)`` on the f32 tensor: same operands, another summation order."""
    from adaprompt_amd import functional as HF
    x = rnd(2, 16, 16, 640, seed=1)
    up = ops.upsample2x_bf16(x)
    ref = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(up, ref)
    pk = ops.PackedConv(rnd(640, 640, 3, 3, seed=2, scale=(9 * 640) ** -0.5), rnd(640, seed=3))
    y_new, _ = ops.conv2d(up, pk.fwd, pk.O4, 3, 1, 1, bias=pk.bias)
    y_old, _ = ops.conv2d(x, pk.fwd, pk.O4, 3, 1, 1, up=1, bias=pk.bias)
    assert rel(y_new, y_old) < 2e-6
    # and through the block Function, both switches
    outs = []
    for flag in (True, False):
        old, HF.UPSAMPLE_BF16 = HF.UPSAMPLE_BF16, flag
        try:
            xi = x.clone().requires_grad_(True)
            y = HF.ConvFn.apply(xi, pk, "up")
            y.backward(rnd(*y.shape, seed=4))
            outs.append((y.detach(), xi.grad.detach()))
        finally:
            HF.UPSAMPLE_BF16 = old
    assert rel(outs[0][0], outs[1][0]) < 2e-6 and rel(outs[0][1], outs[1][1]) < 2e-6


def test_conv2d_epilogue_and_splitk():
    B, Cin, Cout, H = 2, 640, 320, 8
    x = bf(rnd(B, Cin, H, H, seed=1))
    w = bf(rnd(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5))
    bias, emb, res = rnd(Cout, seed=3), rnd(B, Cout, seed=4), rnd(B, Cout, H, H, seed=5)
    ref = 0.5 * F.conv2d(x, w, None, padding=1) + bias[None, :, None, None] + emb[:, :, None, None] + res
    pk = ops.PackedConv(w, bias)
    xin = nhwc(x).to(torch.bfloat16)
    y32, _ = ops.conv2d(xin, pk.fwd, Cout, 3, 1, 1, bias=pk.bias, chan_add=emb, residual=nhwc(res), alpha=0.5)
    assert rel(nchw(y32), ref) < 2e-5
    for ks in (4, 0):          # forced and automatic split-K (slab reduce: bit-reproducible)
        y32s, y16s = ops.conv2d(xin, pk.fwd, Cout, 3, 1, 1, bias=pk.bias, chan_add=emb, residual=nhwc(res), alpha=0.5,
                                ksplit=ks, out_bf16=True)
        assert rel(nchw(y32s), ref) < 2e-5
        assert rel(nchw(y16s.float()), ref) < 4e-3
        y32t, _ = ops.conv2d(xin, pk.fwd, Cout, 3, 1, 1, bias=pk.bias, chan_add=emb, residual=nhwc(res), alpha=0.5, ksplit=ks)
        assert torch.equal(y32s, y32t)


@pytest.mark.parametrize("case", [(2, 320, 320, 16, 3, 1, 1, 0), (2, 640, 320, 8, 1, 1, 0, 0),
                                  (2, 320, 320, 16, 3, 2, 1, 0), (2, 320, 320, 8, 3, 1, 1, 1),
                                  (2, 128, 256, 8, 3, 1, 1, 0)])
def test_conv2d_data_grad(case):
    """dX of the conv through the same kernel with the mode-1 weight pack."""
    B, Cin, Cout, H, K, stride, pad, up = case
    x = bf(rnd(B, Cin, H, H, seed=1)).requires_grad_(True)
    w = bf(rnd(Cout, Cin, K, K, seed=2, scale=(Cin * K * K) ** -0.5))
    xr = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    y = F.conv2d(xr, w, None, stride=stride, padding=pad)
    gy = bf(rnd(*y.shape, seed=7))
    (gx_ref,) = torch.autograd.grad(y, x, gy)
    pk = ops.PackedConv(w)
    g = nhwc(gy)
    if stride == 2:
        # transposed conv: zero-insert gather (up=2), pad K-1-pad, output = input size
        gx, _ = ops.conv2d(g, pk.bwd, pk.bwd.shape[1], K, 1, K - 1 - pad, up=2, out_hw=(H, H))
    else:
        gx, _ = ops.conv2d(g, pk.bwd, pk.bwd.shape[1], K, 1, K - 1 - pad)
        if up:
            gx = ops.sumpool2x2(gx)
    assert rel(nchw(gx)[:, :Cin], gx_ref) < 2e-5


def test_linear_and_batched_matmul():
    x = bf(rnd(3, 77, 768, seed=1))
    w = bf(rnd(320, 768, seed=2, scale=768 ** -0.5))
    b = rnd(320, seed=3)
    r = rnd(3, 77, 320, seed=4)
    pk = ops.PackedConv(w, b)
    y32, y16 = ops.linear(x, pk.fwd, 320, bias=pk.bias, residual=r, out_bf16=True)
    ref = F.linear(x, w, b) + r
    assert rel(y32, ref) < 2e-5 and rel(y16.float(), ref) < 4e-3
    a = bf(rnd(2, 200, 64, seed=5)).to(torch.bfloat16)
    c = bf(rnd(2, 136, 64, seed=6)).to(torch.bfloat16)
    got = ops.batched_matmul_nt(a, c, alpha=0.25)
    assert rel(got, 0.25 * torch.einsum("gmk,gnk->gmn", a.float(), c.float())) < 2e-5


# ---------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------
def _gn_variant():
    from adaprompt_amd import _lib
    return _lib.call_long("adap_groupnorm_last_variant")


@pytest.mark.parametrize("B,C,H,rows_fwd,rows_bwd", [
    (4, 320, 64, 6, 6),        # the 64x64 level of the UNet at the training batch size
    (4, 640, 64, 11, 0),       # fwd: 12-row variant; bwd: two-pass (12 rows of x + dy do not fit the register file)
    (4, 960, 64, 16, 0),
    (4, 1920, 32, 8, 8),
    (4, 1280, 16, 2, 2),
    (4, 2560, 8, 1, 1),
    (1, 320, 64, 3, 3),        # one instance (the distillation mix): 114 slabs of 36 rows
    (7, 640, 32, 5, 5),        # the student's batched passes
    (3, 352, 24, 1, 1),        # odd sizes: 11 channels per group, 576 pixels = 52 slabs of 11 rows + a ragged one of 4
    (2, 128, 24, 0, 0),        # 4 channels per group: an 8-channel oct would span three groups -> two-pass
    (16, 320, 64, 0, 0),       # too large for the register-resident path: two-pass
])
def test_groupnorm_single_launch_path(B, C, H, rows_fwd, rows_bwd):
    """the single-launch GroupNorm (one read of x, workgroups of a sample meeting at an arrival counter) against the
    two-pass kernels on the same inputs -- same fixed-order fp64 finish, so results agree to f32 round-off -- and against
    torch in f64; which path ran is read back from the library.  Repeated launches reuse the self-resetting counters."""
    import os
    x = torch.randn(B, H, H, C, generator=torch.Generator().manual_seed(1)) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(2))
    beta = 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(3))
    gy = torch.randn(B, H, H, C, generator=torch.Generator().manual_seed(4))
    addend = torch.randn(B, H, H, C, generator=torch.Generator().manual_seed(5))
    xd, gyd, ad = x.to(dev()), gy.to(dev()), addend.to(dev())
    ref_in = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    for act, eps in ((1, 1e-5), (0, 1e-6)):
        ref = F.group_norm(ref_in, 32, gamma.double(), beta.double(), eps)
        if act:
            ref = F.silu(ref)
        (gref,) = torch.autograd.grad(ref, ref_in, gy.double().permute(0, 3, 1, 2))
        ref_nhwc, gref_nhwc = ref.detach().permute(0, 2, 3, 1), gref.permute(0, 2, 3, 1)
        for xdt, gdt in ((torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)):
            xi, gi = xd.to(xdt), gyd.to(gdt)
            outs = {}
            for mode in ("single", "two_pass"):
                if mode == "two_pass":
                    ops.gn_two_pass(True)
                try:
                    for rep in range(3):                                    # back-to-back launches: counters must reset
                        y32, y16, mean, rstd = ops.groupnorm_fwd(xi, gamma.to(dev()), beta.to(dev()), eps, act, out_f32=True,
                                                                 out_bf16=True)
                    vf = _gn_variant()
                    dx32, dx16 = ops.groupnorm_bwd(gi, xi, gamma.to(dev()), beta.to(dev()), mean, rstd, act, out_bf16=True,
                                                   add_from=ad)
                    vb = _gn_variant()
                    _, dx16_only = ops.groupnorm_bwd(gi, xi, gamma.to(dev()), beta.to(dev()), mean, rstd, act, out_f32=False,
                                                     out_bf16=True)
                finally:
                    if mode == "two_pass":
                        ops.gn_two_pass(False)
                assert (vf, vb) == ((rows_fwd, rows_bwd) if mode == "single" else (0, 0)), (mode, vf, vb)
                outs[mode] = (y32, y16, mean, rstd, dx32, dx16, dx16_only)
            s, t = outs["single"], outs["two_pass"]
            assert rel(s[2], t[2]) < 1e-6 and rel(s[3], t[3]) < 1e-6            # mean / rstd: same partial order per slab? no -- same math
            assert rel(s[0], t[0]) < 2e-6 and rel(s[4], t[4]) < 2e-5
            assert rel(s[1].float(), t[1].float()) < 2e-3 and rel(s[5].float(), t[5].float()) < 2e-3
            if xdt == torch.float32 and gdt == torch.float32:
                assert rel(s[0].cpu(), ref_nhwc) < 1e-5
                assert rel((s[4] - ad).cpu(), gref_nhwc) < 3e-5
            assert rel(s[6].float(), (s[4] - ad)) < 5e-3
    assert not ops.gn_sync_poisoned()


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,bf16_out,kind", [
    (2, 64, 64, 128, 128, True, False, "halo"), (3, 32, 64, 64, 256, False, True, "halo"),
    (4, 64, 64, 256, 512, True, False, "halo"), (4, 128, 128, 128, 128, False, True, "halo"),
    (4, 256, 256, 128, 128, False, False, "down"),        # f32 operand, stride 2: conv_gemm_kernel<128, true>
    (4, 256, 256, 8, 128, False, False, "cin8"),          # the VAE's conv_in: the 256-row ring kernel
    (4, 64, 64, 128, 128, False, False, "up")])           # nearest x2 + conv (the decoder's Upsample)
def test_groupnorm_statistics_from_the_conv_epilogue(B, H, W, Cin, Cout, res, bf16_out, kind):
    """``conv2d(gn_stats=True)``: the contraction's epilogue leaves (sum, sum of squares) records per tile and group, and
    ``groupnorm_fwd`` on that output takes mean / rstd from them instead of reading the tensor a second time.  The conv's own
    output must not change by a bit; the statistics must be those of the two-pass kernels up to summation order (for a bf16-only
    output: up to the rounding the records do not see); the records are bit-reproducible."""
    from adaprompt_amd import _lib
    x = bf(rnd(B, H, W, Cin, seed=1))
    x = x if kind == "down" else x.to(torch.bfloat16)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    bias = rnd(Cout, seed=3)
    Ho, Wo = (H // 2, W // 2) if kind == "down" else ((2 * H, 2 * W) if kind == "up" else (H, W))
    r = (rnd(B, Ho, Wo, Cout, seed=4) + 3.0) if res else None        # a mean well away from 0: sq/n - mean^2 must survive it
    pk = ops.PackedConv(w, bias)
    gw, gb = rnd(Cout, seed=5) + 1.0, rnd(Cout, seed=6)
    geo = {"halo": dict(stride=1, pad=1), "cin8": dict(stride=1, pad=1), "down": dict(stride=2, pad=0, out_hw=(Ho, Wo)),
           "up": dict(stride=1, pad=1, up=1)}[kind]

    def run(stats):
        y32, y16 = ops.conv2d(x, pk.fwd, pk.O4, 3, bias=pk.bias, residual=r, out_f32=not bf16_out, out_bf16=bf16_out,
                              gn_stats=stats, **geo)
        y = y16 if bf16_out else y32
        _, a16, mean, rstd = ops.groupnorm_fwd(y, gw, gb, 1e-6, 1)
        return y, a16, mean, rstd, getattr(y, ops.GN_STATS_ATTR, None)
    y0, a0, m0, r0, s0 = run(False)
    y1, a1, m1, r1, s1 = run(True)
    variant = _lib.call_long("adap_conv2d_last_variant")
    assert variant == {"halo": 4128, "down": 128, "cin8": 2128, "up": 1128}[kind], variant
    assert s0 is None and s1 is not None and s1[1] == Ho * Wo // 64
    assert torch.equal(y0, y1)
    tol = 2e-3 if bf16_out else 2e-5
    assert float((m1 - m0).abs().max()) < tol * float(m0.abs().max() + 1) and rel(r1, r0) < tol, (m0, m1, r0, r1)
    assert rel(a1.float(), a0.float()) < (4e-3 if bf16_out else 1e-4)
    _, _, m2, r2, s2 = run(True)
    assert torch.equal(s1[0], s2[0]) and torch.equal(m1, m2) and torch.equal(r1, r2)
    for _ in range(4):                    # (the output itself stays bit-reproducible with the statistics epilogue in place)
        assert torch.equal(run(True)[0], y0)
    # a kernel without that epilogue (a 160-wide channel tile) leaves no records and the normalisation falls back to its own pass
    pk1 = ops.PackedConv(rnd(320, Cin, 1, 1, seed=7, scale=Cin ** -0.5), None)
    y32, _ = ops.conv2d(x.to(torch.bfloat16), pk1.fwd, pk1.O4, 1, gn_stats=True)
    assert getattr(y32, ops.GN_STATS_ATTR, None) is None and _lib.call_long("adap_conv2d_last_gn_chunks") == 0


@pytest.mark.parametrize("shape", [(3, 8, 4096), (4, 11520), (2, 16, 77, 768), (1, 5, 33), (4, 72000), (1, 8193)])
def test_ortho_subtract_fused_rows(shape):
    """``ortho_subtract`` over the last dim (ldm/util.py:280) as one launch each way (``OrthoRowsFn``) against the torch
    expressions and their autograd: a - <a,b>/(<b,b>+1e-6) b, gradients into both operands."""
    from adaprompt_amd.ldm.util import ortho_subtract
    a0, b0, g = rnd(*shape, seed=1), rnd(*shape, seed=2) + 0.3, rnd(*shape, seed=3)

    def torch_form(a, b):
        coeff = (a * b).sum(dim=-1) / ((b * b).sum(dim=-1) + 1e-6)
        return a - coeff[..., None] * b
    a1, b1 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    out = ortho_subtract(a1, b1)
    assert type(out.grad_fn).__name__ == "OrthoRowsFnBackward"
    out.backward(g)
    a2, b2 = a0.double().requires_grad_(True), b0.double().requires_grad_(True)
    ref = torch_form(a2, b2)
    ref.backward(g.double())
    assert rel(out, ref) < 1e-6 and rel(a1.grad, a2.grad) < 1e-5 and rel(b1.grad, b2.grad) < 1e-5
    # only one operand differentiable (gradient scalers / detached references)
    a3 = a0.clone().requires_grad_(True)
    ortho_subtract(a3, b0).backward(g)
    assert rel(a3.grad, a2.grad) < 1e-5


@pytest.mark.parametrize("B,H,Cin,Cout", [(4, 64, 320, 320), (2, 32, 320, 640), (4, 16, 1280, 1280), (3, 8, 2560, 1280),
                                           (1, 32, 960, 640)])
def test_resblock_one_c_call_equals_the_per_op_sequence(B, H, Cin, Cout):
    """``adap_resblock_fwd`` / ``_bwd`` (csrc/blocks.hip: the frozen ResBlock's launches issued from one C call) against the same
    block issued op by op from Python (``functional.RESBLOCK_C`` off): output, saved statistics and the data gradient must be
    bit-identical -- the same kernels with the same arguments, only the host side differs."""
    from adaprompt_amd import functional as HF
    x = rnd(B, H, H, Cin, seed=1)
    emb = rnd(B, Cout, seed=2)
    g = rnd(B, H, H, Cout, seed=3)
    P = {"gn1": (rnd(Cin, seed=4) + 1, rnd(Cin, seed=5)), "gn2": (rnd(Cout, seed=6) + 1, rnd(Cout, seed=7)),
         "conv1": ops.PackedConv(rnd(Cout, Cin, 3, 3, seed=8, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=9)),
         "conv2": ops.PackedConv(rnd(Cout, Cout, 3, 3, seed=10, scale=(9 * Cout) ** -0.5), rnd(Cout, seed=11)),
         "skip": None if Cin == Cout else ops.PackedConv(rnd(Cout, Cin, 1, 1, seed=12, scale=Cin ** -0.5), rnd(Cout, seed=13)),
         "train": None}

    def run(c_call):
        old = HF.RESBLOCK_C
        HF.RESBLOCK_C = c_call
        try:
            xi = x.clone().requires_grad_(True)
            out = HF.ResBlockFn.apply(xi, emb, P)
            out.backward(g)
            return out.detach(), xi.grad.detach()
        finally:
            HF.RESBLOCK_C = old
    o0, g0 = run(False)
    o1, g1 = run(True)
    assert torch.equal(o0, o1) and torch.equal(g0, g1)
    assert rel(o1, o0) == 0.0 and torch.isfinite(g1).all()


@pytest.mark.parametrize("B,H,C,heads,masked,split_ctx,capture", [
    (2, 32, 320, 8, True, False, True),        # N = 1024: key compaction; token-map capture + its gradient
    (2, 16, 640, 8, True, False, False),       # N = 256: the mask as key bias
    (1, 64, 320, 8, False, False, True),       # the 64 x 64 level, no mask
    (2, 8, 1280, 8, False, True, False),       # d = 160; split K / V context (mix_hijk form, 154 tokens)
    (3, 16, 64, 8, True, True, True)])         # narrow (the test models' width), everything on
def test_transformer_block_one_c_call_equals_the_per_op_sequence(B, H, C, heads, masked, split_ctx, capture):
    """``adap_stblock_fwd`` / ``_bwd`` (csrc/blocks.hip: the frozen SpatialTransformer block's launches, side lane included,
    issued from one C call each way) against the same block issued op by op from Python (``functional.STBLOCK_C`` off) through
    the real module: output, token maps, the input gradient (f32 and its bf16 side copy) and the context gradients must be
    bit-identical -- the same kernels with the same arguments, only the host side differs."""
    from adaprompt_amd import functional as HF
    from adaprompt_amd.ldm.modules.attention import SpatialTransformer, KeyMasks
    torch.manual_seed(7)
    Cctx, M = 96, (154 if split_ctx else 77)
    st = SpatialTransformer(C, heads, C // heads, depth=1, context_dim=Cctx).to(dev())
    with torch.no_grad():
        for p_ in st.parameters():
            p_.copy_(torch.randn_like(p_) * (0.5 if p_.dim() == 1 else p_.shape[1] ** -0.5))
    for p_ in st.parameters():
        p_.requires_grad_(False)
    x = rnd(B, H, H, C, seed=1)
    ck = rnd(B, M, Cctx, seed=2)
    cv = rnd(B, M, Cctx, seed=3)
    g = rnd(B, H, H, C, seed=4)
    img_mask = None
    if masked:
        img_mask = torch.ones(B, 1, 64, 64, device=dev())
        img_mask[:, :, :7] = 0
        img_mask[0, :, :, 50:] = 0
    G = 2
    tok_w = torch.zeros(B, M, G, device=dev())
    tok_w[:, 4:20, 0] = 1.0
    tok_w[:, 20:24, 1] = 0.25
    gt = rnd(B, heads, H * H, G, seed=5)
    blk = st.transformer_blocks[0]

    def run(c_call, x_grad=True):
        old = HF.STBLOCK_C
        HF.STBLOCK_C = c_call
        try:
            blk.attn2.save_attn_vars = capture
            blk.attn2.token_weights = tok_w if capture else None
            blk.attn2.tokmap_only = capture
            xi = x.clone().requires_grad_(x_grad)
            cki, cvi = ck.clone().requires_grad_(True), cv.clone().requires_grad_(True)
            ctx_arg = (cvi, cki) if split_ctx else cki
            HF.MODEL_STAMP = None
            out = st(xi, ctx_arg, None if img_mask is None else KeyMasks(img_mask))
            roots, grads = [out], [g]
            tm = None
            if capture:
                tm = blk.attn2.cached_activations["attnscore_tokmap"]
                HF.join_side_lane()
                roots.append(tm)
                grads.append(gt)
            torch.autograd.backward(roots, grads)
            torch.cuda.synchronize()
            if not x_grad:
                assert xi.grad is None
                return (out.detach(), None if tm is None else tm.detach().clone(), None, None, cki.grad.detach().clone(),
                        None if cvi.grad is None else cvi.grad.detach().clone())
            gx16 = HF._operand(xi.grad)           # (the bf16 side copy the block left for its predecessor)
            return (out.detach(), None if tm is None else tm.detach().clone(), xi.grad.detach().clone(),
                    None if gx16 is xi.grad else gx16.clone(), cki.grad.detach().clone(),
                    None if cvi.grad is None else cvi.grad.detach().clone())
        finally:
            HF.STBLOCK_C = old
    names_all = ("out", "tokmap", "gx", "gx16", "g_ck", "g_cv")
    a = run(False)
    n0 = list(HF.STB_CALLS)
    b = run(True)
    assert HF.STB_CALLS == [n0[0] + 1, n0[1] + 1, n0[2]], "the C path was not taken"
    # the block's input needs no gradient (the UNet's first transformer block): the call stops after the cross attention
    # (ADAP_STB_NO_GX) -- same output, same token maps, bit-identical context gradients
    c = run(True, x_grad=False)
    assert HF.STB_CALLS == [n0[0] + 2, n0[1] + 2, n0[2] + 1], "the pruned backward was not taken"
    for i in (0, 1, 4, 5):
        assert (b[i] is None) == (c[i] is None) and (b[i] is None or torch.equal(b[i], c[i])), names_all[i]
    names = ("out", "tokmap", "gx", "gx16", "g_ck", "g_cv")
    for n, u, v in zip(names, a, b):
        assert (u is None) == (v is None), n
        if u is not None:
            assert torch.isfinite(u.float()).all() and torch.equal(u, v), (n, rel(v.float(), u.float()))
    assert float(a[2].abs().max()) > 0 and float(a[4].abs().max()) > 0
    assert not ops.gn_sync_poisoned()


def test_groupnorm_single_launch_only_on_one_stream():
    """the workgroups of a single-launch GroupNorm wait for each other, so only ONE stream per device may issue them (two
    such kernels in flight could each hold part of the chip and wait for the rest): the default stream takes that path, a
    side stream (the VAE / teacher prefetchers) gets the two-launch kernels; both in flight together, results exact."""
    B, C, H = 4, 320, 64
    g, b = (1 + 0.1 * rnd(C, seed=2)).to(dev()), (0.1 * rnd(C, seed=3)).to(dev())
    xs = [torch.randn(B, H, H, C, device=dev(), generator=torch.Generator(device=dev()).manual_seed(i)) for i in range(2)]
    want = [ops.groupnorm_fwd(x, g, b, 1e-5, 1, out_f32=True)[0].clone() for x in xs]
    assert _gn_variant() == 6
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    got = [[], []]
    for it in range(20):
        with torch.cuda.stream(side):
            got[1].append(ops.groupnorm_fwd(xs[1], g, b, 1e-5, 1, out_f32=True)[0])
            assert _gn_variant() == 0
        got[0].append(ops.groupnorm_fwd(xs[0], g, b, 1e-5, 1, out_f32=True)[0])
        assert _gn_variant() == 6
    torch.cuda.synchronize()
    for y in got[0]:
        assert torch.equal(y, want[0])
    for y in got[1]:
        assert rel(y, want[1]) < 2e-6
    assert not ops.gn_sync_poisoned()


@pytest.mark.parametrize("B,C,H,eps,act", [(2, 320, 16, 1e-5, 1), (2, 1920, 8, 1e-5, 1), (1, 2560, 8, 1e-5, 1),
                                           (2, 640, 8, 1e-6, 0), (2, 128, 40, 1e-6, 1), (2, 32, 8, 1e-5, 1),
                                           (4, 960, 32, 1e-5, 1)])
def test_groupnorm_fwd_bwd(B, C, H, eps, act):
    x = (rnd(B, C, H, H, seed=1) * 1.5 + 0.3).requires_grad_(True)
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if act:
        ref = F.silu(ref)
    y32, y16, mean, rstd = ops.groupnorm_fwd(nhwc(x.detach()), gamma, beta, eps, act, out_f32=True, out_bf16=True)
    assert rel(nchw(y32), ref) < 1e-5
    assert rel(nchw(y16.float()), ref) < 4e-3
    gy = rnd(*ref.shape, seed=4)
    (gx_ref,) = torch.autograd.grad(ref, x, gy)
    dx32, dx16 = ops.groupnorm_bwd(nhwc(gy), nhwc(x.detach()), gamma, beta, mean, rstd, act, out_f32=True, out_bf16=True)
    assert rel(nchw(dx32), gx_ref) < 2e-5
    assert rel(nchw(dx16.float()), gx_ref) < 4e-3
    # bf16 input tensor (block-internal activations): same statistics in fp32
    xb = bf(x.detach()).requires_grad_(True)
    refb = F.group_norm(xb, 32, gamma, beta, eps)
    if act:
        refb = F.silu(refb)
    yb32, yb16, meanb, rstdb = ops.groupnorm_fwd(nhwc(xb.detach()).to(torch.bfloat16), gamma, beta, eps, act, out_f32=True,
                                                 out_bf16=True)
    assert rel(nchw(yb32), refb) < 1e-5 and rel(nchw(yb16.float()), refb) < 4e-3
    (gxb_ref,) = torch.autograd.grad(refb, xb, gy)
    _, dxb16 = ops.groupnorm_bwd(nhwc(gy).to(torch.bfloat16), nhwc(xb.detach()).to(torch.bfloat16), gamma, beta, meanb, rstdb,
                                 act, out_f32=False, out_bf16=True)
    assert rel(nchw(dxb16.float()), gxb_ref) < 6e-3
    base = rnd(B, H, H, C, seed=9)
    acc = base.clone()
    ops.groupnorm_bwd(nhwc(gy).to(torch.bfloat16), nhwc(x.detach()), gamma, beta, mean, rstd, act, accumulate_into=acc)
    assert rel(nchw(acc - base), gx_ref) < 4e-3
    # out-of-place form: dx + addend into a new tensor, the addend untouched; the bf16 copy holds the same sum
    keep = base.clone()
    out32, out16 = ops.groupnorm_bwd(nhwc(gy).to(torch.bfloat16), nhwc(x.detach()), gamma, beta, mean, rstd, act,
                                     out_bf16=True, add_from=base)
    assert torch.equal(base, keep) and out32.data_ptr() != base.data_ptr()
    assert torch.equal(out32, acc)
    assert rel(out16.float(), out32) < 3e-3


@pytest.mark.parametrize("rows,D", [(512, 320), (100, 640), (64, 1280), (33, 32)])
def test_layernorm_fwd_bwd(rows, D):
    x = (rnd(rows, D, seed=1) * 2 + 0.5).requires_grad_(True)
    gamma, beta = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    ref = F.layer_norm(x, (D,), gamma, beta, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.detach(), gamma, beta)
    assert rel(y.float(), ref) < 4e-3
    gy = rnd(rows, D, seed=4)
    (gx_ref,) = torch.autograd.grad(ref, x, gy)
    dx = ops.layernorm_bwd(gy, x.detach(), gamma, mean, rstd)
    assert rel(dx, gx_ref) < 2e-5
    acc = torch.ones_like(dx)
    acc2, acc16 = ops.layernorm_bwd(gy, x.detach(), gamma, mean, rstd, accumulate_into=acc, want_bf16=True)
    assert acc2 is acc and rel(acc - 1, gx_ref) < 2e-5
    assert torch.equal(acc16, acc.to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def ref_attention(q, k, v, heads, mask=None):
    B, N, C = q.shape
    d = C // heads

    def sp(t):
        return t.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    qq, kk, vv = sp(q), sp(k), sp(v)
    sim = torch.einsum("bhid,bhjd->bhij", qq, kk) * d ** -0.5
    if mask is not None:
        sim = sim.masked_fill(~mask.bool()[:, None, None, :], -torch.finfo(sim.dtype).max)
    attn = sim.softmax(-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, vv)
    return out.permute(0, 2, 1, 3).reshape(B, N, C), sim, attn


ATTN_CASES = [
    # B, heads, N, M, d, mask
    (2, 8, 256, 256, 40, False),
    (2, 8, 256, 256, 40, True),
    (1, 8, 1024, 1024, 40, False),
    (2, 8, 200, 200, 80, True),          # ragged N and M
    (2, 8, 64, 64, 160, False),
    (2, 8, 256, 77, 40, False),          # cross attention
    (2, 8, 64, 77, 160, False),
    (2, 8, 128, 77, 80, False),
    (2, 8, 96, 40, 8, False),            # narrow test model dims
    (2, 8, 96, 96, 16, True),
    (1, 8, 64, 64, 32, False),
    # the 64x64 level at full size: enough workgroups for the two-query-blocks-per-wave forward variant
    (4, 8, 4096, 4096, 40, True),
    (4, 8, 4096, 77, 40, False),
]


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("case", [(2, 8, 300, 333, 40, True), (1, 8, 1024, 1024, 40, False), (1, 8, 512, 257, 40, True),
                                  (2, 8, 260, 256, 40, False), (4, 8, 4096, 4096, 40, True), (1, 8, 700, 700, 24, True)])
def test_attention_fwd_interleaved_kernel(case, mode):
    """``attn_fwd_il_kernel`` (opt-in: the wave software-pipelined over the key tiles, P V of tile t-1 / softmax of t / Q K^T of
    t+1 in one instruction stream, staging two steps deep; 8 or 4 waves per workgroup): ragged, masked, odd and even tile
    counts, full size -- output and LSE against fp32 torch and against the default kernel.  The keys' magnitude grows along the
    sequence, so the running reference point moves in LATE tiles too (the rescale of O deferred to the end of a step)."""
    from adaprompt_amd import _lib
    B, H, N, M, d, use_mask = case
    C = H * d
    q, k, v = bf(rnd(B, N, C, seed=1)) * 2.0, bf(rnd(B, M, C, seed=2)), bf(rnd(B, M, C, seed=3))
    k = k * torch.linspace(0.5, 3.0, M, device=k.device).view(1, M, 1)
    mask = None
    if use_mask:
        mask = (torch.rand(B, M, generator=torch.Generator().manual_seed(5)) > 0.3).to(dev())
        mask[:, 0] = True
    km = mask.to(torch.uint8).contiguous() if mask is not None else None
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    q, k, v = qb.float(), kb.float(), vb.float()
    _lib.call("adap_attention_set_debug", mode, -1, -1, -1)
    try:
        out, lse = ops.attention_fwd(qb, kb, vb, H, km)
        assert _lib.call_long("adap_attention_fwd_last_variant") == {3: 6, 4: 4}[mode]
    finally:
        _lib.call("adap_attention_set_debug", 0, -1, -1, -1)
    out0, lse0 = ops.attention_fwd(qb, kb, vb, H, km)
    assert _lib.call_long("adap_attention_fwd_last_variant") in (1, 2)
    ref, sim, _ = ref_attention(q, k, v, H, mask)
    assert rel(out.float(), ref) < 6e-3, rel(out.float(), ref)
    assert rel(lse, torch.logsumexp(sim, dim=-1)) < 1e-4
    assert rel(out.float(), out0.float()) < 6e-3 and rel(lse, lse0) < 1e-4


@pytest.mark.parametrize("case", [(2, 8, 300, 333, 40, True), (1, 8, 1024, 1024, 40, False), (2, 8, 520, 77, 40, False),
                                  (1, 8, 64, 64, 32, False), (2, 8, 96, 40, 8, False), (1, 8, 700, 700, 64, True),
                                  (4, 8, 4096, 4096, 40, False)])
def test_attention_fwd_ping_pong_kernel(case):
    """the ping-pong forward (8-wave workgroups, SIMD partners half a tile apart, thresholded running max) on ragged,
    masked, single-tile and full-size shapes: output and LSE against fp32 torch, and against the query-stationary kernel."""
    import os
    from adaprompt_amd import _lib
    B, H, N, M, d, use_mask = case
    C = H * d
    q, k, v = bf(rnd(B, N, C, seed=1)), bf(rnd(B, M, C, seed=2)), bf(rnd(B, M, C, seed=3))
    mask = None
    if use_mask:
        mask = (torch.rand(B, M, generator=torch.Generator().manual_seed(5)) > 0.3).to(dev())
        mask[:, 0] = True
    km = mask.to(torch.uint8).contiguous() if mask is not None else None
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    _lib.call("adap_attention_set_debug", 2, -1, -1, -1)          # ping-pong forward always
    try:
        out, lse = ops.attention_fwd(qb, kb, vb, H, km)
        assert _lib.call_long("adap_attention_fwd_last_variant") == 3
    finally:
        _lib.call("adap_attention_set_debug", 0, -1, -1, -1)
    out0, lse0 = ops.attention_fwd(qb, kb, vb, H, km)
    assert _lib.call_long("adap_attention_fwd_last_variant") in (1, 2)
    ref, sim, _ = ref_attention(q, k, v, H, mask)
    assert rel(out.float(), ref) < 6e-3, rel(out.float(), ref)
    assert rel(lse, torch.logsumexp(sim, dim=-1)) < 1e-4
    assert rel(out.float(), out0.float()) < 6e-3 and rel(lse, lse0) < 1e-4       # two roundings of the same bf16 P sums


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(case):
    B, H, N, M, d, use_mask = case
    C = H * d
    q = bf(rnd(B, N, C, seed=1)).requires_grad_(True)
    k = bf(rnd(B, M, C, seed=2)).requires_grad_(True)
    v = bf(rnd(B, M, C, seed=3)).requires_grad_(True)
    mask = None
    if use_mask:
        g = torch.Generator().manual_seed(5)
        mask = (torch.rand(B, M, generator=g) > 0.3).to(dev())
        mask[:, 0] = True
    ref, sim, attn = ref_attention(q, k, v, H, mask)
    km = mask.to(torch.uint8).contiguous() if mask is not None else None
    qb, kb, vb = (t.detach().to(torch.bfloat16) for t in (q, k, v))
    out, lse = ops.attention_fwd(qb, kb, vb, H, km)
    assert rel(out.float(), ref) < 6e-3, rel(out.float(), ref)
    lse_ref = torch.logsumexp(sim, dim=-1)
    assert rel(lse, lse_ref) < 1e-4
    do = bf(rnd(B, N, C, seed=4))
    gq, gk, gv = torch.autograd.grad(ref, (q, k, v), do)
    dq, dk, dv = ops.attention_bwd(qb, kb, vb, out, do.to(torch.bfloat16), lse, H, km, out_dtype=torch.float32)
    assert rel(dv, gv) < 1e-2, ("dv", rel(dv, gv))
    assert rel(dq, gq) < 1e-2, ("dq", rel(dq, gq))
    assert rel(dk, gk) < 1e-2, ("dk", rel(dk, gk))


def test_attention_online_softmax_rescale():
    """one key whose score dwarfs the others appears late: forces the running-max rescale branch."""
    B, H, N, M, d = 1, 8, 64, 256, 40
    C = H * d
    q = bf(rnd(B, N, C, seed=1))
    k = bf(rnd(B, M, C, seed=2))
    v = bf(rnd(B, M, C, seed=3))
    k[:, 200] = q[:, 5] * 4.0          # huge score for query 5 (and large for others) in the 4th key tile
    ref, _, _ = ref_attention(q, k, v, H)
    out, _ = ops.attention_fwd(q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), H)
    assert rel(out.float(), ref) < 6e-3


@pytest.mark.parametrize("B,N,M,d", [(2, 256, 77, 40), (1, 64, 77, 160), (2, 100, 154, 80)])
def test_attention_capture(B, N, M, d):
    H = 8
    C = H * d
    q, k = bf(rnd(B, N, C, seed=1)), bf(rnd(B, M, C, seed=2))
    _, sim, attn = ref_attention(q, k, k, H)
    score, prob, qs = ops.attention_capture(q.to(torch.bfloat16), k.to(torch.bfloat16), H)
    assert rel(score, sim) < 1e-5 and rel(prob, attn) < 1e-5
    qref = q.reshape(B, N, H, d).permute(0, 2, 1, 3) * math.sqrt(d ** -0.5)
    assert rel(qs, qref) < 1e-6


# ---------------------------------------------------------------------------------------------
# small kernels
# ---------------------------------------------------------------------------------------------
def test_geglu_fwd_bwd():
    h = bf(rnd(300, 2 * 1280, seed=1)).requires_grad_(True)
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    out = ops.geglu_fwd(h.detach().to(torch.bfloat16))
    assert rel(out.float(), ref) < 4e-3
    do = bf(rnd(300, 1280, seed=2))
    (gh,) = torch.autograd.grad(ref, h, do)
    dh = ops.geglu_bwd(do.to(torch.bfloat16), h.detach().to(torch.bfloat16))
    assert rel(dh.float(), gh) < 4e-3


def test_linear_small_and_timestep_embedding():
    t = torch.tensor([0, 1, 500, 999], device=dev())
    emb = ops.timestep_embedding(t, 320)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev()) / half)
    args = t[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    assert float((emb - ref).abs().max()) < 2e-4          # sin/cos of arguments up to 999 rad in fp32
    w1, b1 = rnd(1280, 320, seed=1, scale=320 ** -0.5), rnd(1280, seed=2)
    y = ops.linear_small(ref, w1, b1, post_silu=True)
    assert rel(y, F.silu(F.linear(ref, w1, b1))) < 1e-5
    w2 = rnd(640, 1280, seed=3, scale=1280 ** -0.5)
    y2 = ops.linear_small(y, w2, None, pre_silu=True)
    assert rel(y2, F.linear(F.silu(y), w2)) < 1e-5


def test_q_sample_posterior_mse():
    B = 4
    x0, n = rnd(B, 64, 64, 4, seed=1), rnd(B, 64, 64, 4, seed=2)
    t = torch.tensor([0, 10, 500, 999], device=dev())
    sa, sb = torch.rand(1000, device=dev()), torch.rand(1000, device=dev())
    got = ops.q_sample(x0, n, t, sa, sb)
    ref = sa[t].view(B, 1, 1, 1) * x0 + sb[t].view(B, 1, 1, 1) * n
    assert rel(got, ref) < 1e-6
    # a timestep outside the schedule is not allowed to read past the tables: that sample comes back NaN, the others
    # are untouched (the reference raises an IndexError at this point)
    bad = ops.q_sample(x0, n, torch.tensor([0, 1000, -3, 999], device=dev()), sa, sb)
    assert torch.isnan(bad[1]).all() and torch.isnan(bad[2]).all()
    assert torch.equal(bad[0], got[0]) and torch.equal(bad[3], got[3])
    mom = rnd(B, 64, 64, 8, seed=3) * 3
    mom[..., 4:] *= 10
    z = ops.posterior_sample(mom, n, 0.18215)
    refz = 0.18215 * (mom[..., :4] + torch.exp(0.5 * mom[..., 4:].clamp(-30, 20)) * n)
    assert rel(z, refz) < 1e-5
    out = rnd(B, 64, 64, 4, seed=4).requires_grad_(True)
    im = (torch.rand(B, 64, 64, device=dev()) > 0.2).float()
    fg = (torch.rand(B, 64, 64, device=dev()) > 0.6).float()
    pix = F.mse_loss(n * im[..., None], out * im[..., None], reduction="none")
    wfg = (fg * im)[..., None].expand_as(pix)
    wbg = ((1 - fg) * im * 0.1)[..., None].expand_as(pix)
    ref_loss = ((pix * wfg).sum() + (pix * wbg).sum()) / (wfg.sum() + wbg.sum() + 1e-6)
    (gref,) = torch.autograd.grad(ref_loss, out)
    loss, grad = ops.masked_mse(out.detach(), n, im, fg, 1.0, 0.1)
    assert abs(float(loss) - float(ref_loss)) / float(ref_loss) < 1e-5
    assert rel(grad, gref) < 1e-5


def test_concat_pool_pad_transpose_softmax_axpy():
    a, b = rnd(2, 8, 8, 320, seed=1), rnd(2, 8, 8, 640, seed=2)
    assert torch.equal(ops.concat2(a, b), torch.cat([a, b], dim=-1))
    x = rnd(2, 16, 16, 64, seed=3)
    ref = x.reshape(2, 8, 2, 8, 2, 64).sum(dim=(2, 4))
    assert rel(ops.sumpool2x2(x), ref) < 1e-6
    img = rnd(2, 32, 32, 3, seed=4)
    p = ops.pad_cast_bf16(img, 8)
    assert p.shape[-1] == 8 and torch.equal(p[..., :3], img.to(torch.bfloat16)) and float(p[..., 3:].abs().max()) == 0
    t = rnd(2, 100, 72, seed=5).to(torch.bfloat16)
    assert torch.equal(ops.transpose_bf16(t), t.transpose(1, 2).contiguous())
    S = rnd(2, 256, 256, seed=6) * 8
    cls = torch.randint(0, 3, (2, 256), device=dev(), dtype=torch.uint8)
    P = ops.vae_softmax(S, 0.125, cls)
    ref = (S * 0.125).softmax(-1)
    allowed = (cls[:, :, None] == cls[:, None, :]) & (cls[:, :, None] != 0)
    assert rel(P.float(), ref * allowed) < 4e-3
    assert rel(ops.vae_softmax(S, 0.125).float(), ref) < 4e-3
    y, xx = rnd(1024, seed=7), rnd(1024, seed=8)
    y0 = y.clone()
    ops.axpy_(y, xx, 0.5)
    assert rel(y, y0 + 0.5 * xx) < 1e-6


def test_errors_are_loud():
    from adaprompt_amd._lib import HipError
    x = rnd(1, 4, 4, 12).to(torch.bfloat16)       # Cin not a multiple of 8
    w = torch.zeros(1, 8, 12, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(HipError):
        ops.conv2d(x, w, 8, 1)


def test_add2_strided_and_bf16_copy():
    """adap_add2: the meeting point of a skip connection's two gradients -- one contiguous, one a channel slice of a
    wider tensor -- exact in f32, with the bf16 copy of the sum."""
    B, H, C = 2, 16, 320
    a = rnd(B, H, H, C, seed=1)
    wide = rnd(B, H, H, C + 640, seed=2)
    b = wide[..., 640:]                                    # rows 960 floats apart
    y32, y16 = ops.add2(a, b)
    assert torch.equal(y32, a + b)
    assert torch.equal(y16, (a + b).to(torch.bfloat16))
    y32b, none = ops.add2(b, a, want_bf16=False)
    assert none is None and torch.equal(y32b, y32)


def test_linear_small_more_than_eight_rows():
    """the timestep-embedding MLP at the sampler's batch of 16: slices of 8 rows through the register-resident kernel."""
    x = rnd(19, 320, seed=3)
    w, bias = rnd(1280, 320, seed=4) * 0.05, rnd(1280, seed=5) * 0.1
    got = ops.linear_small(x, w, bias, post_silu=True)
    ref = F.silu(x @ w.t() + bias)
    assert rel(got, ref) < 1e-5


@pytest.mark.parametrize("B,N,M,d", [(2, 256, 77, 160), (1, 1024, 77, 80), (2, 100, 154, 40), (1, 64, 5, 40)])
def test_attention_capture_bwd(B, N, M, d):
    """gradients of the captured attnscore / q*d^-1/4 side outputs added into the layer's bf16 dq / dk
    (ddpm.py:3246-3270 reads attnscore with gradient); reference: the same contraction in fp64."""
    H = 8
    g = torch.Generator().manual_seed(21)
    q = torch.randn(B, N, H * d, generator=g).bfloat16()
    k = torch.randn(B, M, H * d, generator=g).bfloat16()
    ds = torch.randn(B, H, N, M, generator=g) * 0.1
    dqs = torch.randn(B, H, N, d, generator=g)
    dq0 = torch.randn(B, N, H * d, generator=g).bfloat16()
    dk0 = torch.randn(B, M, H * d, generator=g).bfloat16()
    scale = d ** -0.5
    qh = q.double().view(B, N, H, d).permute(0, 2, 1, 3)
    kh = k.double().view(B, M, H, d).permute(0, 2, 1, 3)
    dq_ref = scale * (ds.double() @ kh) + d ** -0.25 * dqs.double()                    # [B,H,N,d]
    dk_ref = scale * (ds.double().transpose(2, 3) @ qh)                                  # [B,H,M,d]
    dq_ref = dq0.double() + dq_ref.permute(0, 2, 1, 3).reshape(B, N, H * d)
    dk_ref = dk0.double() + dk_ref.permute(0, 2, 1, 3).reshape(B, M, H * d)
    dev = torch.device("cuda:0")
    dq, dk = dq0.to(dev).clone(), dk0.to(dev).clone()
    ops.attention_capture_bwd(ds.to(dev), dqs.to(dev), q.to(dev), k.to(dev), dq, dk, H)
    # the sum is rounded to bf16 once: half an ulp of the result
    assert rel(dq.float().cpu(), dq_ref.float()) < 3e-3
    assert rel(dk.float().cpu(), dk_ref.float()) < 3e-3
    # q-only gradient (stage 2 reads q without attnscore in some losses): dk untouched
    dq2, dk2 = dq0.to(dev).clone(), dk0.to(dev).clone()
    ops.attention_capture_bwd(None, dqs.to(dev), q.to(dev), k.to(dev), dq2, None, H)
    ref2 = dq0.double() + (d ** -0.25 * dqs.double()).permute(0, 2, 1, 3).reshape(B, N, H * d)
    assert rel(dq2.float().cpu(), ref2.float()) < 3e-3


@pytest.mark.parametrize("B,N,M,d,G", [(2, 256, 77, 160, 2), (1, 1024, 77, 80, 1), (2, 4096, 77, 40, 2), (1, 100, 154, 40, 3)])
def test_attention_token_maps_fwd_bwd(B, N, M, d, G):
    """token maps T = attnscore . w (per head) from the capture kernel, and their gradient applied to dq / dk without the
    dense d attnscore: both against the dense formulation in fp64."""
    H = 8
    g = torch.Generator().manual_seed(33)
    q = torch.randn(B, N, H * d, generator=g).bfloat16()
    k = torch.randn(B, M, H * d, generator=g).bfloat16()
    w = (torch.rand(B, M, G, generator=g) < 0.2).float() * torch.randint(1, 3, (B, M, G), generator=g).float()
    dt = torch.randn(B, H, N, G, generator=g)
    dq0 = torch.randn(B, N, H * d, generator=g).bfloat16()
    dk0 = torch.randn(B, M, H * d, generator=g).bfloat16()
    scale = d ** -0.5
    qh = q.double().view(B, N, H, d).permute(0, 2, 1, 3)
    kh = k.double().view(B, M, H, d).permute(0, 2, 1, 3)
    score = scale * qh @ kh.transpose(2, 3)                                  # [B,H,N,M]
    t_ref = score @ w.double()[:, None]                                      # [B,H,N,G]
    ds = dt.double() @ w.double()[:, None].transpose(2, 3)                   # dense d attnscore [B,H,N,M]
    dq_ref = dq0.double() + (scale * ds @ kh).permute(0, 2, 1, 3).reshape(B, N, H * d)
    dk_ref = dk0.double() + (scale * ds.transpose(2, 3) @ qh).permute(0, 2, 1, 3).reshape(B, M, H * d)
    dev = torch.device("cuda:0")
    sc, pr, qs, tm = ops.attention_capture(q.to(dev), k.to(dev), H, tok_w=w.to(dev))
    assert rel(tm.cpu(), t_ref.float()) < 1e-5
    assert rel(sc.cpu(), score.float()) < 1e-5
    # the maps alone (what the recon iteration asks for): formed as <q, scale * sum_m w k> without any score row
    none_sc, none_pr, none_qs, tm_only = ops.attention_capture(q.to(dev), k.to(dev), H, tok_w=w.to(dev), dense=False)
    assert none_sc is None and none_pr is None and none_qs is None
    assert rel(tm_only.cpu(), t_ref.float()) < 1e-5 and rel(tm_only.cpu(), tm.cpu()) < 1e-5
    dq, dk = dq0.to(dev).clone(), dk0.to(dev).clone()
    ops.attention_tokmap_bwd(dt.to(dev), w.to(dev), q.to(dev), k.to(dev), dq, dk, H)
    assert rel(dq.float().cpu(), dq_ref.float()) < 3e-3
    assert rel(dk.float().cpu(), dk_ref.float()) < 3e-3
    # keys that no group lists keep their gradient bit for bit
    untouched = (w.sum(-1) == 0)
    assert torch.equal(dk.cpu()[untouched], dk0[untouched])
    # fixed-order two-stage reduction over the queries: a repeat is bit-identical
    dq_b, dk_b = dq0.to(dev).clone(), dk0.to(dev).clone()
    ops.attention_tokmap_bwd(dt.to(dev), w.to(dev), q.to(dev), k.to(dev), dq_b, dk_b, H)
    assert torch.equal(dq_b, dq) and torch.equal(dk_b, dk)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,d,G", [(2, 256, 77, 160, 2), (1, 1024, 77, 80, 1), (2, 4096, 77, 40, 2), (1, 100, 154, 40, 3),
                                       (2, 200, 200, 40, 2)])
def test_attention_bwd_with_the_token_map_gradient_folded_in(B, N, M, d, G):
    """adap_attention_bwd_tok (the token maps' gradient added inside the dQ / dK epilogues, kw and gq from
    adap_attention_tokmap_prep) against adap_attention_bwd followed by the read-modify-write adap_attention_tokmap_bwd: the same
    dq / dk up to the one bf16 rounding the folded form saves; dv untouched.  (M = 200: the dK/dV kernel's direct epilogue, M = 77 /
    154: its query-split reduce.)"""
    H = 8
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(91)
    q, k, v = (torch.randn(B, n, H * d, generator=g).bfloat16().to(dev) for n in (N, M, M))
    go = torch.randn(B, N, H * d, generator=g).bfloat16().to(dev)
    w = ((torch.rand(B, M, G, generator=g) < 0.2).float() * torch.randint(1, 3, (B, M, G), generator=g).float()).to(dev)
    dt = (torch.randn(B, H, N, G, generator=g) * 0.3).to(dev)
    out, lse = ops.attention_fwd(q, k, v, H, None)
    dq0, dk0, dv0 = ops.attention_bwd(q, k, v, out, go, lse, H)
    dq_ref, dk_ref = dq0.clone(), dk0.clone()
    ops.attention_tokmap_bwd(dt, w, q, k, dq_ref, dk_ref, H)
    prep = ops.attention_tokmap_prep(dt, w, q, k, H)
    dq1, dk1, dv1 = ops.attention_bwd(q, k, v, out, go, lse, H, tok=(dt, w, prep))
    torch.cuda.synchronize()
    assert torch.equal(dv1, dv0)
    assert float((dq_ref.float() - dq0.float()).abs().max()) > 0           # the token maps do contribute
    assert rel(dq1.float(), dq_ref.float()) < 4e-3 and rel(dk1.float(), dk_ref.float()) < 4e-3
    # and against fp64: dq0/dk0 (bf16) + the dense formulation
    qh = q.double().view(B, N, H, d).permute(0, 2, 1, 3).cpu()
    kh = k.double().view(B, M, H, d).permute(0, 2, 1, 3).cpu()
    ds = dt.double().cpu() @ w.double().cpu()[:, None].transpose(2, 3)
    sc = d ** -0.5
    dq64 = dq0.double().cpu() + (sc * ds @ kh).permute(0, 2, 1, 3).reshape(B, N, H * d)
    dk64 = dk0.double().cpu() + (sc * ds.transpose(2, 3) @ qh).permute(0, 2, 1, 3).reshape(B, M, H * d)
    assert rel(dq1.float().cpu(), dq64.float()) < 4e-3 and rel(dk1.float().cpu(), dk64.float()) < 4e-3


@pytest.mark.parametrize("expo", [2, 3, 1])
@pytest.mark.parametrize("R,D,demean,align,rgs", [(8, 4096, True, True, 1.0), (5, 64, True, True, 0.05), (3, 1024, False, False, 1.0),
                                                  (700, 768, True, True, 0.05), (4, 256, True, True, 0.0)])
def test_cosine_rows_fwd_bwd(R, D, demean, align, rgs, expo):
    """adap_cosine_rows against the torch chain of ldm/util.py:499-517 (demean, ScaleGrad, sign-preserving square,
    F.cosine_embedding_loss) and its autograd, in fp64."""
    from adaprompt_amd import functional as HF
    g = torch.Generator().manual_seed(17)
    x = torch.randn(R, D, generator=g) + 0.3
    r = torch.randn(R, D, generator=g) * 0.7 - 0.1
    gl = torch.randn(R, generator=g)
    xd, rd = x.double().requires_grad_(True), r.double().requires_grad_(True)
    xt, rt = (xd - xd.mean(-1, keepdim=True), rd - rd.mean(-1, keepdim=True)) if demean else (xd, rd)
    tgt = rt * rt.abs().pow(expo - 1)
    ref = F.cosine_embedding_loss(xt, tgt, torch.full((R,), 1.0 if align else -1.0, dtype=torch.float64), reduction="none")
    (ref * gl.double()).sum().backward()
    dev = torch.device("cuda:0")
    xh, rh = x.to(dev).requires_grad_(True), r.to(dev).requires_grad_(True)
    out = HF.CosineRowsFn.apply(xh, rh, demean, align, rgs, expo)
    (out * gl.to(dev)).sum().backward()
    assert rel(out.cpu(), ref.detach().float()) < 1e-5
    assert rel(xh.grad.cpu(), xd.grad.float()) < 2e-5
    if rgs == 0:
        assert rh.grad is None
    else:
        assert rel(rh.grad.cpu(), (rd.grad * rgs).float()) < 2e-5


@pytest.mark.parametrize("L,B,H,N,G,have_bg,use_iw", [(3, 2, 8, 256, 2, True, False), (1, 4, 8, 64, 2, True, True),
                                                       (2, 2, 8, 1024, 1, False, False), (3, 4, 8, 4096, 2, True, False)])
def test_mask_hinges_fwd_bwd(L, B, H, N, G, have_bg, use_iw):
    """adap_mask_hinges_* against the torch chain of ddpm.py:4143-4238 (masked means with the 0.5 ScaleGrad, four hinge
    terms, masked_mean over the positives) and its autograd, in fp64."""
    from adaprompt_amd import functional as HF
    g = torch.Generator().manual_seed(23)
    maps = torch.randn(L, B, H, N, G, generator=g)
    f = (torch.rand(B, N, generator=g) < 0.35).float()
    iw = torch.tensor([1.0, 0.5, 0.0, 2.0][:B]) if use_iw else None
    gout = torch.randn(4, L, generator=g)
    m, m3 = 0.4, 0.4 * 16 / 4
    md = maps.double().requires_grad_(True)
    S = md[..., 0]
    Gm = md[..., 1] if have_bg else None
    fm = f.double().view(1, B, 1, N).expand(L, B, H, N)
    bm = 1 - fm

    class Half(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, gg):
            return gg * 0.5

    def mean_over(x, msk):
        return (x * msk).sum(dim=(2, 3), keepdim=True) / msk.sum(dim=(2, 3), keepdim=True).clamp(min=1e-6)

    def hinge(x):
        pos = (x > 0).double()
        xw = x if iw is None else x * iw.double().view(1, B, 1, 1)
        return (xw * pos).sum(dim=(1, 2, 3)) / pos.sum(dim=(1, 2, 3)).clamp(min=1e-6)

    aS = mean_over(Half.apply(S * fm), fm)
    terms = [hinge(S * bm + m - aS)]
    if have_bg:
        aG = mean_over(Gm * bm, bm)
        terms += [hinge(Gm * fm + m - aG), hinge(Gm * fm + m3 - aS), hinge(S * bm + m - aG)]
    ref = torch.stack(terms + [torch.zeros(L, dtype=torch.float64)] * (4 - len(terms)))
    (ref * gout.double()).sum().backward()
    dev = torch.device("cuda:0")
    mh = maps.to(dev).requires_grad_(True)
    out = HF.MaskHingesFn.apply(mh, f.to(dev), None if iw is None else iw.to(dev), m, m3, have_bg)
    (out * gout.to(dev)).sum().backward()
    assert rel(out.cpu(), ref.detach().float()) < 2e-5
    assert rel(mh.grad.cpu(), md.grad.float()) < 2e-5


def test_integration_md_ctypes_stub_runs_verbatim():
    """the binding INTEGRATION.md shows a reference maintainer (GroupNorm32 + SiLU through the C ABI with plain ctypes)
    is executed as printed there and compared with torch's group_norm + silu."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch\n.*?)```", src, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                                  # the stub opens the library by its path relative to the repo root
    try:
        exec(code, ns)
        gn = torch.nn.GroupNorm(32, 320, eps=1e-5).cuda()
        with torch.no_grad():
            gn.weight.normal_(1, 0.2)
            gn.bias.normal_(0, 0.2)
        x = torch.randn(2, 16, 16, 320, device="cuda")
        y, mean, rstd = ns["groupnorm32_silu"](x, gn)
    finally:
        os.chdir(cwd)
    ref = F.silu(gn(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    assert rel(y.float(), ref) < 5e-3


@pytest.mark.parametrize("B,H,N,d", [(3, 8, 4096, 40), (2, 8, 1024, 80), (2, 4, 200, 40)])
def test_attention_with_compacted_keys_equals_masked_attention(B, H, N, d):
    """key_count: the kept keys gathered to the front and the masked ones left out -- forward and backward against the same
    kernels run with the byte mask (the two differ only in the order of summation), and against torch."""
    C = H * d
    g = torch.Generator().manual_seed(3)
    q, k, v, do = (bf(torch.randn(B, N, C, generator=g)).to(dev()) for _ in range(4))
    mask = (torch.rand(B, N, generator=g) > 0.3)
    mask[0, :] = True                                      # one sample keeps everything
    mask[-1, N // 2:] = False                              # one loses a whole half (whole tiles of masked keys)
    mask = mask.to(dev())
    km = mask.to(torch.uint8).contiguous()
    qb, kb, vb, dob = (t.to(torch.bfloat16) for t in (q, k, v, do))
    perm = torch.argsort(km, dim=1, descending=True, stable=True)
    inv = torch.argsort(perm, dim=1).to(torch.int32).contiguous()
    perm = perm.to(torch.int32).contiguous()
    count = km.sum(dim=1, dtype=torch.int32).contiguous()
    kv = torch.cat([kb, vb], dim=-1)
    kvc = ops.gather_rows_bf16(kv, perm)
    assert torch.equal(kvc[0], kv[0]) and torch.equal(kvc[1, 0], kv[1, int(perm[1, 0])])
    o_c, lse_c = ops.attention_fwd(qb, kvc[..., :C], kvc[..., C:], H, None, key_count=count)
    o_m, lse_m = ops.attention_fwd(qb, kb, vb, H, km)
    assert rel(o_c.float(), o_m.float()) < 3e-3 and rel(lse_c, lse_m) < 2e-4          # (tile boundaries differ: other rounding)
    ref, sim, _ = ref_attention(q, k, v, H, mask)
    assert rel(o_c.float(), ref) < 6e-3
    dq_m, dk_m, dv_m = ops.attention_bwd(qb, kb, vb, o_m, dob, lse_m, H, km)
    dkvc = torch.empty(B, N, 2 * C, device=dev(), dtype=torch.bfloat16)
    dq_c, _, _ = ops.attention_bwd(qb, kvc[..., :C], kvc[..., C:], o_c, dob, lse_c, H, None, dk=dkvc[..., :C], dv=dkvc[..., C:],
                                   key_count=count)
    dkv = ops.gather_rows_bf16(dkvc, inv)
    assert rel(dq_c.float(), dq_m.float()) < 6e-3
    assert rel(dkv[..., :C].float(), dk_m.float()) < 6e-3 and rel(dkv[..., C:].float(), dv_m.float()) < 6e-3
    # masked keys receive exactly zero gradient
    assert float(dkv[~mask].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,N,M,d,masked", [(2, 8, 4096, 4096, 40, False), (2, 8, 1024, 1024, 80, True), (1, 8, 256, 256, 160, False),
                                              (2, 4, 300, 333, 40, True)])
def test_attention_with_pre_scaled_queries(B, H, N, M, d, masked):
    """scale = 0: q carries d^-1/2 * log2(e) (the self-attention projection's pack, functional.PRESCALE_Q) -- forward output /
    lse and the backward's dq (with respect to the pre-scaled q), dk, dv against fp64 autograd of the same function."""
    from adaprompt_amd import functional as HF
    C = H * d
    c = HF.q_prescale(d)
    g = torch.Generator().manual_seed(17)
    q0 = torch.randn(B, N, C, generator=g)
    qp = (q0 * c).bfloat16()                                   # what the scaled projection would emit
    k, v = torch.randn(B, M, C, generator=g).bfloat16(), torch.randn(B, M, C, generator=g).bfloat16()
    go = torch.randn(B, N, C, generator=g).bfloat16()
    mask = None
    if masked:
        mask = torch.rand(B, M, generator=g) > 0.3
        mask[:, 0] = True
    km = None if mask is None else mask.to(torch.uint8).contiguous().to(dev())
    out, lse = ops.attention_fwd(qp.to(dev()), k.to(dev()), v.to(dev()), H, km, scale=0.0)
    dq, dk, dv = ops.attention_bwd(qp.to(dev()), k.to(dev()), v.to(dev()), out, go.to(dev()), lse, H, km, scale=0.0)
    # fp64: scores = ln2 * (q' . k), i.e. the plain attention of q = q' / c with scale d^-1/2
    qd = qp.double().requires_grad_(True)
    kd, vd = k.double().requires_grad_(True), v.double().requires_grad_(True)
    qh = qd.view(B, N, H, d).permute(0, 2, 1, 3)
    kh, vh = kd.view(B, M, H, d).permute(0, 2, 1, 3), vd.view(B, M, H, d).permute(0, 2, 1, 3)
    sim = (qh @ kh.transpose(2, 3)) * 0.6931471805599453
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(torch.float32).max)
    ref = (sim.softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(B, N, C)
    (ref * go.double()).sum().backward()
    assert rel(out.float().cpu(), ref.detach().float()) < 6e-3
    assert rel(lse.cpu(), torch.logsumexp(sim.detach(), dim=-1).float()) < 1e-4
    assert rel(dq.float().cpu(), qd.grad.float()) < 1.2e-2
    assert rel(dk.float().cpu(), kd.grad.float()) < 1.2e-2
    assert rel(dv.float().cpu(), vd.grad.float()) < 1.2e-2
    # the plain form on the same values agrees to rounding (same kernels otherwise)
    out2, lse2 = ops.attention_fwd((qp.float() / c).bfloat16().to(dev()), k.to(dev()), v.to(dev()), H, km)
    assert rel(out.float(), out2.float()) < 1e-2


@pytest.mark.parametrize("rows,C", [(4096, 320), (1024, 640), (300, 1280), (64, 1280)])
def test_feed_forward_with_the_geglu_inside_its_contractions(rows, C):
    """adap_linear_geglu_fwd / _bwd (GEGLU in the epilogues of ff.net.0.proj and of ff.net.2's data gradient, the 8C
    pre-activation in a permuted channel order) against the separate kernels around the plain contractions: bit for bit."""
    from adaprompt_amd import functional as HF
    g = torch.Generator().manual_seed(23)
    x = (torch.randn(1, rows, C, generator=g) * 0.7).bfloat16().to(dev())
    w1 = (torch.randn(8 * C, C, generator=g) * C ** -0.5).to(dev())
    b1 = (torch.randn(8 * C, generator=g) * 0.1).to(dev())
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).to(dev())
    go = (torch.randn(1, rows, C, generator=g) * 0.5).bfloat16().to(dev())
    wc = HF.WeightCache()
    p1, p1g, p2 = wc.get("ff1", w1, b1), wc.get("ff1g", w1, b1, row_perm="geglu"), wc.get("ff2", w2)
    # separate kernels
    _, h_u = ops.linear(x, p1.fwd, 8 * C, bias=p1.bias, out_f32=False, out_bf16=True)
    gg_u = ops.geglu_fwd(h_u)
    _, dgg = ops.linear(go, p2.bwd, 4 * C, out_f32=False, out_bf16=True)
    dh_u = ops.geglu_bwd(dgg, h_u)
    gx_u, _ = ops.linear(dh_u, p1.bwd, C)
    # fused
    h_f, gg_f = ops.linear_geglu_fwd(x, p1g)
    dh_f = ops.linear_geglu_bwd(go, p2, h_f)
    gx_f, _ = ops.linear(dh_f, p1g.bwd, C)
    torch.cuda.synchronize()
    perm = ops.geglu_row_permutation(8 * C, dev())
    assert torch.equal(h_f, h_u[..., perm])
    assert torch.equal(gg_f, gg_u)
    assert torch.equal(dh_f, dh_u[..., perm])
    # the input gradient sums the same 8C products in a different order
    assert rel(gx_f, gx_u) < 1e-5
    # and the whole thing against torch in fp32
    hh = torch.nn.functional.linear(x.float(), w1.bfloat16().float(), b1)
    a, gate = hh.chunk(2, dim=-1)
    ref = a * torch.nn.functional.gelu(gate)
    assert rel(gg_f.float(), ref) < 6e-3


@pytest.mark.parametrize("rows,K,C,expect_fused", [(1024, 3840, 1280, True),       # 16 x 16 level, to_q|k|v data gradient: split K
                                                   (512, 5120, 640, True),         # a feed-forward data gradient at 32 x 32 rows / 8
                                                   (16384, 320, 320, False)])      # short K: never split -- the request is declined
def test_layernorm_backward_inside_the_splitk_reduce(rows, K, C, expect_fused):
    """``adap_conv2d_next_ln_bwd`` (round 5): the LayerNorm backward (attention.py:267-269) that consumes a data-gradient contraction's
    output, folded into that contraction's split-K reduce pass -- against the contraction followed by ``adap_layernorm_bwd``: the
    accumulated dx (f32) and its bf16 copy bit for bit; where the contraction does not go out split the request is declined
    (``adap_conv2d_last_ln_bwd`` = 0) and the outputs are written as usual."""
    from adaprompt_amd import _lib
    g16 = rnd(rows, K, seed=1, scale=0.3).to(torch.bfloat16)
    pk = ops.PackedConv(rnd(C, K, seed=2, scale=K ** -0.5))
    x = rnd(rows, C, seed=3)
    gamma = rnd(C, seed=4) * 0.2 + 1.0
    _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros_like(gamma))
    base = rnd(rows, C, seed=5)
    # reference: contraction, then the LayerNorm backward accumulating into a copy of the running gradient
    dy, _ = ops.linear(g16, pk.fwd, C, out_f32=True)
    want32 = base.clone()
    _, want16 = ops.layernorm_bwd(dy, x, gamma, mean, rstd, accumulate_into=want32, want_bf16=True)
    # fused: arm the request, run the same contraction
    got32 = base.clone()
    got16 = torch.zeros(rows, C, device=dev(), dtype=torch.bfloat16)
    _lib.call("adap_conv2d_next_ln_bwd", x.data_ptr(), C, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), got32.data_ptr(), C, 1,
              got16.data_ptr(), C)
    dy2, _ = ops.linear(g16, pk.fwd, C, out_f32=True)
    fused = _lib.call_long("adap_conv2d_last_ln_bwd")
    torch.cuda.synchronize()
    assert bool(fused) == expect_fused
    if fused:
        assert torch.equal(got32, want32) and torch.equal(got16, want16)
    else:
        assert torch.equal(dy2, dy) and torch.equal(got32, base)             # declined: nothing of the request was touched
    # the request is one-shot: the next contraction is an ordinary one
    dy3, _ = ops.linear(g16, pk.fwd, C, out_f32=True)
    assert _lib.call_long("adap_conv2d_last_ln_bwd") == 0 and torch.equal(dy3, dy)


# ---------------------------------------------------------------------------------------------
# key masks of the UNet's levels / pixel classes of the VAE's masked attention: one launch each, bit-exact against the
# reference's torch expressions (attention.py:223-232, :332; model.py:196-232)
# ---------------------------------------------------------------------------------------------
def _mask_images(B, h, w, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = (torch.rand(B, 1, h, w, generator=g) > 0.35).float()
    m[:, :, : h // 5] = 0                      # a border band, as the augmentation masks have
    m[:, :, :, -(w // 7):] = 0
    if B > 1:
        m[1] = 0                               # a sample that keeps no key: count = N, perm = identity
    if B > 2:
        m[2] = 1                               # ... and one that keeps all of them
    return m.to(dev())


@pytest.mark.parametrize("B,h,w,sizes", [
    (4, 512, 512, [(64, 64), (32, 32), (16, 16), (8, 8)]),      # the shipped levels (bs 4)
    (3, 500, 380, [(63, 48), (32, 24), (16, 12), (8, 6)]),       # non-integer resize ratios, ragged last chunk
    (2, 64, 64, [(96, 96), (48, 48)]),                           # more than 4096 keys per sample; an upscaled mask
    (1, 37, 53, [(40, 40)]),
])
def test_key_masks_one_launch_equals_the_torch_expressions(B, h, w, sizes):
    from adaprompt_amd import functional as HF
    from adaprompt_amd.ldm.modules.attention import KeyMasks
    img = _mask_images(B, h, w, seed=h + w)
    km = KeyMasks(img, sizes=sizes)
    for H, W in sizes:
        m2 = F.interpolate(img.float(), size=(H, W), mode="nearest")
        want = (m2.reshape(B, H * W) != 0).to(torch.uint8)
        assert torch.equal(km.at(H, W), want), (H, W)
        if H * W >= HF.COMPACT_KEYS_MIN_N:
            c = km.compaction(H, W)
            perm = torch.argsort(want, dim=1, descending=True, stable=True)
            inv = torch.argsort(perm, dim=1)
            count = want.sum(dim=1, dtype=torch.int32)
            count = torch.where(count == 0, torch.full_like(count, H * W), count)
            assert c.perm.dtype == torch.int32 and c.inv_perm.dtype == torch.int32 and c.count.dtype == torch.int32
            assert torch.equal(c.perm.long(), perm) and torch.equal(c.inv_perm.long(), inv) and torch.equal(c.count, count)
            assert c.mask.data_ptr() == km.at(H, W).data_ptr()
        else:
            with pytest.raises(ValueError):
                km.compaction(H, W)
    # a size that was not announced gets its own launch
    H, W = sizes[0][0] // 2 + 3, sizes[0][1] // 2 + 1
    want = (F.interpolate(img.float(), size=(H, W), mode="nearest").reshape(B, H * W) != 0).to(torch.uint8)
    assert torch.equal(km.at(H, W), want)
    with pytest.raises(ValueError):
        KeyMasks(img[:, 0])


@pytest.mark.parametrize("with_aug", [True, False])
def test_pixel_classes_one_launch_equals_the_torch_expression(with_aug):
    from adaprompt_amd.ldm.modules.diffusionmodules.model import AttnBlock
    B, h, w, hw = 3, 512, 384, (64, 48)
    fg = _mask_images(B, h, w, seed=5)
    fg[0, :, 100:200, 50:90] = 0.5                   # (a soft value: fg and 1 - fg both non-zero -> class 1 wins)
    aug = _mask_images(B, h // 2, w // 2, seed=6) if with_aug else None
    got = AttnBlock.pixel_classes({"fg_mask": fg, "aug_mask": aug}, hw, fg)
    f = F.interpolate(fg, size=hw, mode="nearest")
    a = torch.ones_like(f) if aug is None else F.interpolate(aug, size=hw, mode="nearest")
    want = torch.where(f * a != 0, 1, torch.where((1 - f) * a != 0, 2, 0)).reshape(B, -1).to(torch.uint8)
    assert got.dtype == torch.uint8 and torch.equal(got, want)
    assert AttnBlock.pixel_classes(None, hw, fg) is None and AttnBlock.pixel_classes({"fg_mask": None}, hw, fg) is None

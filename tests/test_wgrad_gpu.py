"""Weight-gradient kernels (include/adaprompt_hip.h: adap_conv2d_bwd_weight, adap_colsum, adap_norm_affine_bwd) against
torch autograd on the CPU in fp64.  The contraction runs on bf16 operands with f32 accumulation, so the tight check is
against the reference fed the SAME bf16-rounded operands (what is left is summation order); the loose one against
the unrounded fp32 operands bounds the operand rounding itself."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TIGHT = 2e-4        # same rounded operands: fp32 accumulation order only
LOOSE = 1.5e-2      # bf16 operand rounding, relative to the gradient's norm


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _ref_conv_dw(x, dy, Cout, K, stride, pad, up):
    """x [B,H,W,Cin], dy [B,Ho,Wo,Cout] (CPU, any dtype) -> dW [Cout,Cin,K,K], db in fp64 via autograd."""
    xd = x.double().permute(0, 3, 1, 2)
    if up:
        xd = F.interpolate(xd, scale_factor=2, mode="nearest")
    w = torch.zeros(Cout, x.shape[-1], K, K, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    if pad == "asym":                                   # VAE Downsample: pad (0,1,0,1), stride 2, no conv padding
        xd = F.pad(xd, (0, 1, 0, 1))
        y = F.conv2d(xd, w, b, stride=stride, padding=0)
    else:
        y = F.conv2d(xd, w, b, stride=stride, padding=pad)
    y.backward(dy.double().permute(0, 3, 1, 2))
    return w.grad, b.grad


CONV_CASES = [
    # B, H, W, Cin, Cout, K, stride, pad, up, x dtype, dy dtype
    (2, 16, 16, 64, 96, 3, 1, 1, 0, "bf16", "bf16"),
    (2, 16, 16, 64, 96, 3, 1, 1, 0, "f32", "f32"),
    (1, 8, 8, 320, 320, 3, 1, 1, 0, "bf16", "bf16"),
    (2, 16, 16, 32, 64, 3, 2, 1, 0, "f32", "bf16"),       # UNet Downsample
    (2, 8, 8, 32, 64, 3, 1, 1, 1, "f32", "f32"),          # Upsample: nearest x2 fused in front of the conv
    (2, 16, 16, 4, 64, 3, 1, 1, 0, "f32", "bf16"),        # input conv: Cin = 4 (scalar gather path)
    (2, 16, 16, 64, 4, 3, 1, 1, 0, "bf16", "f32"),        # out conv: Cout = 4
    (2, 12, 20, 40, 72, 1, 1, 0, 0, "bf16", "bf16"),      # 1x1, ragged tiles
    (3, 10, 6, 24, 40, 3, 1, 1, 0, "bf16", "bf16"),       # M = 180: not a multiple of 64
    (1, 8, 8, 512, 72, 3, 1, 1, 0, "bf16", "bf16"),       # taps*Cin = 4608 >= 4096: (tap, ci) becomes the row dimension
    (1, 16, 16, 4104, 40, 1, 1, 0, 0, "bf16", "f32"),     # the same for a 1x1 (ragged: 4104 = 128*32 + 8)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(map(str, c)))
def test_conv2d_bwd_weight(case):
    from adaprompt_amd import ops
    B, H, W, Cin, Cout, K, stride, pad, up, xdt, ydt = case
    g = torch.Generator().manual_seed(1234)
    Hs, Ws = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (Hs + 2 * pad - K) // stride + 1, (Ws + 2 * pad - K) // stride + 1
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, Ho, Wo, Cout, generator=g)
    xq = x.bfloat16() if xdt == "bf16" else x
    yq = dy.bfloat16() if ydt == "bf16" else dy
    dw_ref, db_ref = _ref_conv_dw(xq.bfloat16(), yq.bfloat16(), Cout, K, stride, pad, up)   # kernel rounds both to bf16
    dw_full, _ = _ref_conv_dw(x, dy, Cout, K, stride, pad, up)
    _, db_exact = _ref_conv_dw(xq, yq, Cout, K, stride, pad, up)                              # bias sum reads dy as given
    dev = torch.device("cuda:0")
    dw = torch.zeros(Cout, Cin, K, K, device=dev)
    db = torch.zeros(Cout, device=dev)
    ops.conv2d_bwd_weight(xq.to(dev), yq.to(dev), dw, db, K, stride, pad, up, accumulate=False)
    assert _rel(dw.cpu(), dw_ref) < TIGHT
    assert _rel(dw.cpu(), dw_full) < LOOSE
    assert _rel(db.cpu(), db_exact) < 1e-5
    # accumulate: a second call doubles both
    ops.conv2d_bwd_weight(xq.to(dev), yq.to(dev), dw, db, K, stride, pad, up, accumulate=True)
    assert _rel(dw.cpu(), 2 * dw_ref) < TIGHT
    assert _rel(db.cpu(), 2 * db_exact) < 1e-5
    # deterministic: bit-identical on a repeat
    dw2 = torch.zeros_like(dw)
    dw3 = torch.zeros_like(dw)
    ops.conv2d_bwd_weight(xq.to(dev), yq.to(dev), dw2, None, K, stride, pad, up, accumulate=False)
    ops.conv2d_bwd_weight(xq.to(dev), yq.to(dev), dw3, None, K, stride, pad, up, accumulate=False)
    assert torch.equal(dw2, dw3)


@pytest.mark.parametrize("rows,I,O", [(308, 768, 320), (2 * 77, 768, 1280), (4096, 320, 960), (1000, 40, 24),
                                      (512, 5120, 1280)])
def test_linear_bwd_weight(rows, I, O):
    from adaprompt_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(rows, I, generator=g).bfloat16()
    dy = torch.randn(rows, O, generator=g).bfloat16()
    ref = dy.double().t() @ x.double()
    dev = torch.device("cuda:0")
    dw = torch.full((O, I), 0.5, device=dev)
    db = torch.zeros(O, device=dev)
    ops.linear_bwd_weight(x.to(dev), dy.to(dev), dw, db, accumulate=True)
    assert _rel(dw.cpu() - 0.5, ref) < TIGHT
    assert _rel(db.cpu(), dy.double().sum(0)) < 1e-5
    # strided rows: a channel slice of a wider tensor (the fused q|k|v gradient)
    wide = torch.randn(rows, O + 64, generator=g).bfloat16().to(dev)
    dw.zero_()
    ops.linear_bwd_weight(x.to(dev), wide[:, 32:32 + O], dw, None, accumulate=False)
    assert _rel(dw.cpu(), wide[:, 32:32 + O].cpu().double().t() @ x.double()) < TIGHT


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_colsum_per_image(dt):
    from adaprompt_amd import ops
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(3, 24, 24, 200, generator=g)
    dyq = dy.bfloat16() if dt == "bf16" else dy
    dev = torch.device("cuda:0")
    out = torch.zeros(3, 200, device=dev)
    ops.colsum(dyq.to(dev), out, seg_rows=24 * 24, accumulate=False)
    assert _rel(out.cpu(), dyq.double().sum((1, 2))) < 1e-6
    tot = torch.ones(1, 200, device=dev)
    ops.colsum(dyq.to(dev), tot, accumulate=True)
    assert _rel(tot.cpu() - 1, dyq.double().sum((0, 1, 2))[None]) < 1e-5


@pytest.mark.parametrize("C,HW,act,xdt,ydt", [(64, 256, 1, "f32", "bf16"), (320, 64, 1, "bf16", "bf16"),
                                              (96, 100, 0, "f32", "f32")])
def test_groupnorm_affine_bwd(C, HW, act, xdt, ydt):
    from adaprompt_amd import ops
    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.randn(B, HW, C, generator=g) * 1.5 + 0.3
    dy = torch.randn(B, HW, C, generator=g)
    gamma = torch.randn(C, generator=g) * 0.5 + 1.0
    beta = torch.randn(C, generator=g) * 0.2
    xq = x.bfloat16() if xdt == "bf16" else x
    yq = dy.bfloat16() if ydt == "bf16" else dy
    dev = torch.device("cuda:0")
    _, _, mean, rstd = ops.groupnorm_fwd(xq.to(dev), gamma.to(dev), beta.to(dev), 1e-5, act)
    xd = xq.double().permute(0, 2, 1)                       # [B,C,HW]
    ga = gamma.double().clone().requires_grad_(True)
    be = beta.double().clone().requires_grad_(True)
    y = F.group_norm(xd, 32, ga, be, 1e-5)
    if act:
        y = F.silu(y)
    y.backward(yq.double().permute(0, 2, 1))
    dga = torch.zeros(C, device=dev)
    dbe = torch.zeros(C, device=dev)
    ops.norm_affine_bwd(yq.to(dev), xq.to(dev), gamma.to(dev), beta.to(dev), mean, rstd, 0, act, dga, dbe, accumulate=False)
    assert _rel(dga.cpu(), ga.grad) < 1e-4
    assert _rel(dbe.cpu(), be.grad) < 1e-4


def test_layernorm_affine_bwd():
    from adaprompt_amd import ops
    g = torch.Generator().manual_seed(12)
    rows, D = 300, 320
    x = torch.randn(2, rows // 2, D, generator=g) * 2 + 0.5
    dy = torch.randn(2, rows // 2, D, generator=g)
    gamma = torch.randn(D, generator=g) * 0.5 + 1.0
    beta = torch.randn(D, generator=g) * 0.2
    dev = torch.device("cuda:0")
    _, mean, rstd = ops.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev))
    ga = gamma.double().clone().requires_grad_(True)
    be = beta.double().clone().requires_grad_(True)
    F.layer_norm(x.double(), (D,), ga, be, 1e-5).backward(dy.double())
    dga = torch.zeros(D, device=dev)
    dbe = torch.zeros(D, device=dev)
    ops.norm_affine_bwd(dy.to(dev), x.to(dev), gamma.to(dev), None, mean, rstd, 1, 0, dga, dbe, accumulate=False)
    assert _rel(dga.cpu(), ga.grad) < 1e-4
    assert _rel(dbe.cpu(), be.grad) < 1e-4


@pytest.mark.parametrize("B,H,Cin,Cout,K", [(4, 16, 1280, 1280, 3), (4, 64, 320, 320, 3), (4, 32, 1920, 640, 3),
                                            (4, 64, 320, 2560, 1)])
def test_conv2d_bwd_weight_full_size_vs_torch_gpu(B, H, Cin, Cout, K):
    """BASELINE sizes (bs = 4): the UNet's own layer shapes against torch's conv2d weight gradient in fp32 on the same
    device (too large for the CPU fp64 reference), on bf16-representable operands so that only the accumulation order
    differs; plus the size-independent property that the gradient is linear in dy."""
    from adaprompt_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, H, H, Cin, device=dev, generator=g).bfloat16()
    dy1 = torch.randn(B, H, H, Cout, device=dev, generator=g).bfloat16()
    dy2 = torch.randn(B, H, H, Cout, device=dev, generator=g).bfloat16()

    def hip(dy):
        dw = torch.zeros(Cout, Cin, K, K, device=dev)
        ops.conv2d_bwd_weight(x, dy, dw, None, K, 1, K // 2, 0, accumulate=False)
        return dw

    w = torch.zeros(Cout, Cin, K, K, device=dev, requires_grad=True)
    prev = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    try:
        y = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, padding=K // 2)
        y.backward(dy1.float().permute(0, 3, 1, 2))
    finally:
        torch.backends.cudnn.allow_tf32 = prev
    d1 = hip(dy1)
    assert _rel(d1, w.grad) < 1e-3
    # linearity: dW(dy1 + dy2) = dW(dy1) + dW(dy2) up to the bf16 rounding of the summed operand
    d2, d12 = hip(dy2), hip((dy1.float() + dy2.float()).bfloat16())
    assert _rel(d12, d1 + d2) < 6e-3

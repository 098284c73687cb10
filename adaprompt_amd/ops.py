"""Raw (non-autograd) Python entry points of the HIP kernels.

PyTorch is plumbing here: it owns device memory and the stream; every function checks layout,
passes raw pointers to the C ABI (include/adaprompt_hip.h) on torch's current stream and returns
torch tensors.  Activations are pixel-major: [B, H, W, C] / [B, N, C] with the channel dim
contiguous and an arbitrary leading dimension (stride of dim -2)."""
import ctypes
import os

import torch

from . import _lib

F32 = torch.float32
BF16 = torch.bfloat16


_stream = _lib.current_stream


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py's roofline leg): brackets every launch of the selected
    kernel families with events on the stream the kernel runs on (torch's current stream) and keeps the
    algorithmic work of each launch.  Off (None) on the normal path."""

    def __init__(self):
        self.records = []          # (family, work, start_event, end_event)

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, family, work, e0, tag=None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((family, work, e0, e1, tag))

    def by_tag(self):
        """{(family, tag): {launches, work, ms}} -- per-shape view for tuning."""
        torch.cuda.synchronize()
        out = {}
        for fam, work, e0, e1, tag in self.records:
            d = out.setdefault((fam, tag), {"launches": 0, "work": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["work"] += work
            d["ms"] += e0.elapsed_time(e1)
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, work, e0, e1, _tag in self.records:
            d = out.setdefault(fam, {"launches": 0, "work": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["work"] += work
            d["ms"] += e0.elapsed_time(e1)
        return out


TIMER = None


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _rows_ld(t):
    """(rows, ld) of a pixel-major tensor whose leading dims are packed with respect to stride(-2)."""
    if t.is_contiguous() and t.is_cuda:              # the common case, checked in C
        c = t.shape[-1]
        return (t.numel() // c if c else 0), c
    assert t.is_cuda, "HIP kernels need device tensors (no CPU fallback)"
    assert t.stride(-1) == 1 or t.shape[-1] == 1, f"channel dim must be contiguous, got strides {t.stride()}"
    if t.dim() == 1:
        return 1, t.shape[0]
    ld = t.stride(-2)
    rows = t.shape[-2]
    exp = ld * t.shape[-2]
    for i in range(t.dim() - 3, -1, -1):
        assert t.shape[i] == 1 or t.stride(i) == exp, f"leading dims not packed: shape {tuple(t.shape)} strides {t.stride()}"
        exp *= t.shape[i]
        rows *= t.shape[i]
    return rows, ld


def _dt(t):
    if t.dtype == F32:
        return 0
    if t.dtype == BF16:
        return 1
    raise TypeError(f"unsupported dtype {t.dtype}")


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------

def _ceil(a, m):
    return (a + m - 1) // m * m


# Device data that is built once and then CACHED (weight packs, stacked packs, concatenated weights) is built by kernels on
# whatever stream is current at its first use -- and read afterwards from every stream without further ordering.  On one stream
# that is stream order; with several streams of whole passes in flight (MicroBatchLanes: micro-batch 1's forward starts while
# micro-batch 0's is still running) the second stream's first kernels could read a pack the first stream's pack kernel has not
# written yet (seen as a NaN loss in two processes sharing one card, where the lanes drift further apart).  So while such a
# mode is announced (multi_stream(+1): LatentDiffusion.training_window for the duration of a window on lanes), a cache fill drains
# the stream that built it before anybody gets to see the result; fills happen in the first window only (frozen weights).
CACHE_FILLS = 0
_MULTI_STREAM = 0


def multi_stream(delta):
    global _MULTI_STREAM
    _MULTI_STREAM = max(0, _MULTI_STREAM + int(delta))


def note_cache_fill():
    """call AFTER the kernels that build a cached device tensor have been issued."""
    global CACHE_FILLS
    CACHE_FILLS += 1
    if _MULTI_STREAM > 0 and torch.cuda.is_available() and os.environ.get("ADAP_DIAG_NO_FILL_DRAIN") != "1":
        torch.cuda.current_stream().synchronize()


class PackedConv:
    """bf16 weight packs of one nn.Conv2d / nn.Linear: forward [taps][O][I8] and (lazily) the
    data-gradient pack [taps][I4..][O8] (taps flipped, roles swapped).  I is zero-padded to a
    multiple of 8 (image 3->8, latent 4->8 channels)."""

    def __init__(self, weight, bias=None):
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        assert w.dim() == 4 and w.is_cuda
        self.w_f32 = w.contiguous().float()
        self.O, self.I, self.KH, self.KW = self.w_f32.shape
        self.I8 = _ceil(self.I, 8)
        self.O4 = _ceil(self.O, 4)
        self.bias = None if bias is None else bias.detach().float().contiguous()
        if self.bias is not None and self.O4 != self.O:
            b = torch.zeros(self.O4, device=w.device, dtype=F32)
            b[: self.O] = self.bias
            self.bias = b
        self.fwd = torch.empty(self.KH * self.KW, self.O4, self.I8, device=w.device, dtype=BF16)
        _lib.call("adap_pack_conv_weight", self.w_f32.data_ptr(), self.fwd.data_ptr(), self.O, self.I, self.KH,
                  self.KW, 0, self.O4, self.I8, _stream())
        self._bwd = None
        note_cache_fill()

    @property
    def bwd(self):
        if self._bwd is None:
            rows, cols = _ceil(self.I, 4), _ceil(self.O, 8)
            self._bwd = torch.empty(self.KH * self.KW, rows, cols, device=self.fwd.device, dtype=BF16)
            _lib.call("adap_pack_conv_weight", self.w_f32.data_ptr(), self._bwd.data_ptr(), self.O, self.I, self.KH,
                      self.KW, 1, rows, cols, _stream())
            note_cache_fill()
        return self._bwd

    def drop_f32(self):
        """free the f32 master copy once both packs exist (frozen weights)."""
        _ = self.bwd
        self.w_f32 = None


_CONV2D = None


def _conv2d_entry():
    global _CONV2D
    if _CONV2D is None:
        _CONV2D = _lib._fn("adap_conv2d_nhwc")
    return _CONV2D


GN_STATS_ATTR = "_adap_gn_stats"          # (partial records, records per image) a contraction's epilogue left on its output


def conv2d(x, w_packed, Cout, KH=1, stride=1, pad=0, up=0, out_hw=None, bias=None, chan_add=None, residual=None,
           out_f32=True, out_bf16=False, alpha=1.0, ksplit=0, y32=None, y16=None, gn_stats=False):
    """x [B,H,W,Cin] (f32 / bf16) * w_packed [KH*KH][Cout][Cin] -> (y32, y16), each [B,Ho,Wo,Cout] or None.
    Linear layers: pass x as [B, N, 1, Cin].
    ``gn_stats``: the output feeds a GroupNorm(32): ask the epilogue for its group statistics (``adap_conv2d_next_gn_partial``);
    if the kernel the call dispatches to has that epilogue, the outputs carry them (``GN_STATS_ATTR``) and ``groupnorm_fwd``
    skips its statistics pass."""
    # (host time matters here: ~400 calls per training step; what the C entry checks itself -- alignment, leading dims, null
    # pointers -- is not checked twice)
    B, H, W, Cin = x.shape
    ldx = Cin if x.is_contiguous() else _rows_ld(x)[1]
    ws_ = w_packed.shape
    assert w_packed.dtype == BF16 and ws_[0] == KH * KH and ws_[2] == Cin and ws_[1] == Cout, \
        f"weight pack {tuple(ws_)} vs Cin={Cin} Cout={Cout} (padded) taps={KH * KH}"
    if out_hw is None:
        He, We = (2 * H, 2 * W) if up else (H, W)
        Ho = (He + 2 * pad - KH) // stride + 1
        Wo = (We + 2 * pad - KH) // stride + 1
    else:
        Ho, Wo = out_hw
    dev = x.device
    if out_f32 and y32 is None:
        y32 = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=F32)
    ws = None
    if ksplit == 0:
        nws = _lib.size_query("adap_conv2d_workspace_floats", B, Ho, Wo, Cin, Cout, KH, KH)
    else:
        nws = ksplit * B * Ho * Wo * Cout if ksplit > 1 else 0
    if nws:
        ws = torch.empty(nws, device=dev, dtype=F32)
    if out_bf16 and y16 is None:
        y16 = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=BF16)
    ldy32 = 0 if y32 is None else (Cout if y32.is_contiguous() else _rows_ld(y32)[1])
    ldy16 = 0 if y16 is None else (Cout if y16.is_contiguous() else _rows_ld(y16)[1])
    ldr = 0
    if residual is not None:
        assert residual.dtype == F32 and residual.shape[-1] == Cout
        if residual.is_contiguous():
            rrows, ldr = residual.numel() // Cout, Cout
        else:
            rrows, ldr = _rows_ld(residual)
        assert rrows == B * Ho * Wo
    ld_ca = 0
    if chan_add is not None:
        assert chan_add.dtype == F32 and chan_add.dim() == 2 and chan_add.shape[0] == B and chan_add.stride(1) == 1
        ld_ca = chan_add.stride(0)
    if bias is not None:
        assert bias.dtype == F32 and bias.numel() >= Cout
    e0 = TIMER.start() if TIMER is not None else None
    part = None
    if gn_stats and Cout % 32 == 0 and (Ho * Wo) % 256 == 0:
        part = torch.empty(B, Ho * Wo // 64, 32, 2, device=dev, dtype=F32)            # one record per 64 pixels and group
        _lib.call("adap_conv2d_next_gn_partial", part.data_ptr(), Cout // 32)
    rc = _conv2d_entry()(x.data_ptr(), _dt(x), ldx, w_packed.data_ptr(),
                         0 if bias is None else bias.data_ptr(), 0 if chan_add is None else chan_add.data_ptr(), ld_ca,
                         0 if residual is None else residual.data_ptr(), ldr, 0 if y32 is None else y32.data_ptr(), ldy32,
                         0 if y16 is None else y16.data_ptr(), ldy16, B, H, W, Cin, Ho, Wo, Cout, KH, KH, stride, pad, up,
                         float(alpha), ksplit, 0 if ws is None else ws.data_ptr(), 1, 0, 0, 0, 0, _stream())
    if rc != 0:
        raise _lib.HipError(f"adap_conv2d_nhwc failed ({rc}): {_lib.load().adap_last_error().decode()}")
    if part is not None:
        chunks = _lib.call_long("adap_conv2d_last_gn_chunks")
        if chunks > 0:
            for y in (y32, y16):
                if y is not None:
                    # valid for exactly this tensor content: the version and shape travel with the records
                    setattr(y, GN_STATS_ATTR, (part, chunks, y._version, tuple(y.shape)))
    if e0 is not None:
        # algorithmic FLOPs = 2 * MACs of the convolution as the reference's nn.Conv2d / nn.Linear counts them
        v = _lib.call_long("adap_conv2d_last_variant")
        # the symbol names rocprofv3 --stats lists (ring: ..., ONE_TAP; halo: ..., ping-pong schedule, wide patch)
        one_tap = "true" if (KH == 1 and stride == 1 and pad == 0 and up == 0) else "false"
        kname = {0: "conv_gemm_kernel<{bn}, true>", 1: "conv_gemm_kernel<{bn}, false>",
                 2: "conv_gemm_ring_kernel<256, {bn}, 3, {ot}>", 3: "conv_gemm_ring_kernel<128, {bn}, 4, {ot}>",
                 4: "conv3x3_halo_kernel<{bn}, {pp}, {wide}>", 5: "conv3x3_win32_kernel<{bn}, {wide}>"}[v // 1000].format(
                     bn=v % 1000, ot=one_tap, pp="true" if v % 1000 == 128 else "false",
                     wide="true" if (H % 8 == 0 and W % 32 == 0) else "false")     # 8x32 patches, else 16x16
        TIMER.stop(kname, 2.0 * B * Ho * Wo * Cout * Cin * KH * KH, e0,
                   f"M={B * Ho * Wo} N={Cout} K={Cin}x{KH * KH} s{stride} up{up} {'f32' if x.dtype == F32 else 'bf16'}")
    return y32, y16


def conv3x3_rgb(x_hwc, w_packed, Cout, bias=None, gn_stats=False):
    """conv3x3 stride 1 pad 1 on an RGB image [B,H,W,3] f32 with ``PackedConv.fwd`` of its [Cout,3,3,3] weight ([9][Cout][8]);
    Cout in (32, 64, 128) -> f32 [B,H,W,Cout] (``adap_conv3x3_rgb``: the whole patch as one K step).  ``gn_stats`` as ``conv2d``."""
    B, H, W, C3 = x_hwc.shape
    assert x_hwc.dtype == F32 and x_hwc.is_contiguous() and C3 == 3 and tuple(w_packed.shape) == (9, Cout, 8)
    y = torch.empty(B, H, W, Cout, device=x_hwc.device, dtype=F32)
    e0 = TIMER.start() if TIMER is not None else None
    part = None
    if gn_stats and Cout % 32 == 0 and (H * W) % 256 == 0:
        part = torch.empty(B, H * W // 64, 32, 2, device=x_hwc.device, dtype=F32)
        _lib.call("adap_conv2d_next_gn_partial", part.data_ptr(), Cout // 32)
    _lib.call("adap_conv3x3_rgb", x_hwc.data_ptr(), 3, w_packed.data_ptr(), _ptr(bias), y.data_ptr(), 0, B, H, W, Cout, _stream())
    if part is not None:
        chunks = _lib.call_long("adap_conv2d_last_gn_chunks")
        if chunks > 0:
            setattr(y, GN_STATS_ATTR, (part, chunks, y._version, tuple(y.shape)))
    if e0 is not None:
        # an HBM-bound kernel (7 GFLOP against 537 MB of output at 512 x 512): its work is counted in algorithmic BYTES
        # (read the image, write the output), like the GroupNorms, not in FLOPs
        TIMER.stop("hbm:rgb_conv_in", float(B * H * W) * (12 + 4 * Cout), e0, f"M={B * H * W} N={Cout} K=3x9 rgb f32")
    return y


def linear(x, w_packed, Cout, bias=None, residual=None, out_f32=True, out_bf16=False, alpha=1.0):
    """x [..., Cin] -> [..., Cout] through the 1x1 path."""
    shp = x.shape
    rows, ld = _rows_ld(x)
    if (TIMER is None and w_packed.shape[0] == 1 and w_packed.shape[2] == shp[-1] and w_packed.shape[1] == Cout
            and w_packed.dtype == BF16 and w_packed.is_contiguous()):
        # straight to the C entry ([1, rows, 1, Cin] problem): ~250 calls per training step, and the generic conv2d wrapper
        # (strided views of x / residual / outputs, shape algebra) is several microseconds of host time each
        Cin = shp[-1]
        dev = x.device
        out_shape = tuple(shp[:-1]) + (Cout,)
        y32 = torch.empty(out_shape, device=dev, dtype=F32) if out_f32 else None
        y16 = torch.empty(out_shape, device=dev, dtype=BF16) if out_bf16 else None
        rl = 0
        if residual is not None:
            assert residual.dtype == F32 and residual.shape[-1] == Cout
            rr, rl = _rows_ld(residual)
            assert rr == rows
        nws = _lib.size_query("adap_conv2d_workspace_floats", 1, rows, 1, Cin, Cout, 1, 1)
        ws = torch.empty(nws, device=dev, dtype=F32) if nws else None
        rc = _conv2d_entry()(x.data_ptr(), _dt(x), ld, w_packed.data_ptr(), 0 if bias is None else bias.data_ptr(), 0, 0,
                             0 if residual is None else residual.data_ptr(), rl, 0 if y32 is None else y32.data_ptr(),
                             Cout if y32 is not None else 0, 0 if y16 is None else y16.data_ptr(), Cout if y16 is not None else 0,
                             1, rows, 1, Cin, rows, 1, Cout, 1, 1, 1, 0, 0, float(alpha), 0, 0 if ws is None else ws.data_ptr(),
                             1, 0, 0, 0, 0, _stream())
        if rc != 0:
            raise _lib.HipError(f"adap_conv2d_nhwc failed ({rc}): {_lib.load().adap_last_error().decode()}")
        return y32, y16
    x4 = x.as_strided((1, rows, 1, shp[-1]), (rows * ld, ld, ld, 1))
    r4 = None
    if residual is not None:
        rr, rl = _rows_ld(residual)
        r4 = residual.as_strided((1, rr, 1, Cout), (rr * rl, rl, rl, 1))
    y32, y16 = conv2d(x4, w_packed, Cout, 1, bias=bias, residual=r4, out_f32=out_f32, out_bf16=out_bf16, alpha=alpha)
    out_shape = tuple(shp[:-1]) + (Cout,)
    return (None if y32 is None else y32.view(out_shape)), (None if y16 is None else y16.view(out_shape))


def batched_matmul_nt(a, b, out_dtype=F32, alpha=1.0):
    """a [G, M, K] bf16, b [G, N, K] bf16 -> [G, M, N] (sum over K); the VAE mid attention's two bmm."""
    G, M, K = a.shape
    G2, N, K2 = b.shape
    assert G == G2 and K == K2 and a.is_contiguous() and b.is_contiguous() and a.dtype == BF16 and b.dtype == BF16
    y = torch.empty(G, M, N, device=a.device, dtype=out_dtype)
    y32, y16 = (y, None) if out_dtype == F32 else (None, y)
    _lib.call("adap_conv2d_nhwc", a.data_ptr(), 1, K, b.data_ptr(), 0, 0, 0, 0, 0, _ptr(y32), N, _ptr(y16), N,
              1, M, 1, K, M, 1, N, 1, 1, 1, 0, 0, float(alpha), 1, 0, G, M * K, N * K, M * N, M * N, _stream())
    return y


# --------------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------------

_GN_SYNC = {}
_GN_SYNC_STREAM = {}
_GN_TWO_PASS = 0


def gn_two_pass(on):
    """Hold the GroupNorms to the two-launch kernels while ``on`` (nestable: a count).  The single-launch kernel's workgroups
    wait for each other inside the launch, so its whole grid (up to one 512-thread workgroup per CU) must be resident; while a
    collective kernel occupies part of the chip for milliseconds -- ``GradReducer`` between ``reduce()`` and ``wait()`` -- the
    late workgroups would only start once it ends and the early ones would spin until then (measured:
    tests/test_parallel_gpu.py::test_groupnorm_beside_a_resident_collective_kernel).  Decided on the host, no sync."""
    global _GN_TWO_PASS
    _GN_TWO_PASS = max(0, _GN_TWO_PASS + (1 if on else -1))


def set_gn_single_launch_stream(device, raw_stream):
    """which stream's GroupNorms may take the single-launch path on ``device`` (default: the device's default stream)."""
    _GN_SYNC_STREAM[device.index] = raw_stream


def gn_single_launch_stream(device):
    """the raw stream that owns the single-launch GroupNorm path on ``device`` (the default stream unless set)."""
    owner = _GN_SYNC_STREAM.get(device.index)
    if owner is None:
        owner = _GN_SYNC_STREAM[device.index] = torch.cuda.default_stream(device).cuda_stream
    return owner


def gn_sync_buffer(device):
    """-> data pointer of the zero-initialised arrival-counter buffer of the single-launch GroupNorm kernels, or 0.

    The workgroups of such a kernel wait for each other inside the launch, so all of them must be resident together.  One
    kernel of <= 256 workgroups always is; two of them in flight on different streams could each hold part of the chip and
    wait for the rest for ever.  Hence exactly ONE stream per device takes that path (the default stream, where the UNet's
    forward / backward run); GroupNorms issued on any other stream (the prefetchers' side streams) run the two-launch
    kernels.  Kernels of one stream run in order and the counters reset themselves: allocated and zeroed once."""
    if _GN_TWO_PASS:
        return 0
    st = _stream()
    owner = _GN_SYNC_STREAM.get(device.index)
    if owner is None:
        owner = _GN_SYNC_STREAM[device.index] = torch.cuda.default_stream(device).cuda_stream
    if st != owner:
        return 0
    buf = _GN_SYNC.get(device.index)
    if buf is None:
        buf = _GN_SYNC[device.index] = torch.zeros(_lib.call_long("adap_groupnorm_sync_ints"), device=device, dtype=torch.int32)
    return buf.data_ptr()


def gn_sync_poisoned():
    """True if a single-launch GroupNorm ever gave up waiting for its sample's other workgroups (tests assert it is False)."""
    return any(int(b[1]) != 0 for b in _GN_SYNC.values())          # word 1 = poison (norms.hip GN_SYNC_POISON)


class GnSyncTimeout(RuntimeError):
    pass


_GN_POLL = {}      # device index -> (pinned int32[1], event of the copy in flight or None)


def gn_poison_poll():
    """Sync-free check of the single-launch GroupNorm's poison word, meant to be called once per optimiser step: looks at
    the copy issued by the PREVIOUS call (raises ``GnSyncTimeout`` if it is set), then issues the next 4-byte
    device -> pinned-host copy on the current stream.  A timed-out exchange also turns that GroupNorm's statistics into
    NaN (norms.hip gn_sweep), so the loss is NaN in the same step; this names the cause at most one step later.
    The workgroups of such a kernel wait for each other, so the whole grid must be resident: set ADAP_GN_TWO_PASS=1 when a
    device is shared by several processes / ranks, masked or partitioned."""
    for idx, buf in _GN_SYNC.items():
        host, ev = _GN_POLL.get(idx, (None, None))
        if ev is not None and ev.query() and int(host[0]) != 0:
            raise GnSyncTimeout(f"cuda:{idx}: a single-launch GroupNorm's in-launch exchange timed out (the grid was not "
                                "co-resident: device shared, masked or partitioned?); its outputs are NaN. Set "
                                "ADAP_GN_TWO_PASS=1 for such runs.")
        if host is None:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
        if ev is None or ev.query():
            host.copy_(buf[1:2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        _GN_POLL[idx] = (host, ev)


def groupnorm_fwd(x, gamma, beta, eps, act, out_f32=False, out_bf16=True):
    """x f32 / bf16 [B, ..., C] pixel-major -> (y32, y16, mean[B,32], rstd[B,32])"""
    B, C = x.shape[0], x.shape[-1]
    rows, ldx = _rows_ld(x)
    HW = rows // B
    ws = torch.empty(_lib.size_query("adap_groupnorm_workspace_floats", B, HW, C), device=x.device, dtype=F32)
    mean = torch.empty(B, 32, device=x.device, dtype=F32)
    rstd = torch.empty(B, 32, device=x.device, dtype=F32)
    y32 = torch.empty(x.shape, device=x.device, dtype=F32) if out_f32 else None
    y16 = torch.empty(x.shape, device=x.device, dtype=BF16) if out_bf16 else None
    e0 = TIMER.start() if TIMER is not None else None
    stats = getattr(x, GN_STATS_ATTR, None)          # left by the producing contraction's epilogue (conv2d(gn_stats=True))
    if stats is not None and (stats[2] != x._version or stats[3] != tuple(x.shape) or stats[1] != HW // 64
                              or stats[0].shape[0] != B):
        stats = None                                 # x was written since (or is another view): the statistics pass it is
    if stats is not None:
        part, chunks = stats[0], stats[1]
        _lib.call("adap_groupnorm_fwd_stats", x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), beta.data_ptr(), _ptr(y32), C,
                  _ptr(y16), C, mean.data_ptr(), rstd.data_ptr(), part.data_ptr(), chunks, B, HW, C, float(eps), int(act),
                  _stream())
    else:
        _lib.call("adap_groupnorm_fwd", x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), beta.data_ptr(), _ptr(y32), C, _ptr(y16), C,
                  mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), gn_sync_buffer(x.device), B, HW, C, float(eps),
                  int(act), _stream())
    if e0 is not None:
        # algorithmic bytes: read x once (4 B) + write y (2 B bf16 / 4 B f32) per element
        TIMER.stop("groupnorm_fwd", float(x.numel()) * (x.element_size() + (4 if out_f32 else 0) + (2 if out_bf16 else 0)), e0)
    return y32, y16, mean, rstd


def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, act, out_f32=True, out_bf16=False, accumulate_into=None, add_from=None):
    """``accumulate_into``: dx is added into that f32 tensor in place.  ``add_from``: dx + add_from goes to a NEW f32
    tensor (the addend, e.g. a block's incoming gradient, is left untouched -- no clone needed)."""
    B, C = x.shape[0], x.shape[-1]
    rows, ldx = _rows_ld(x)
    _, lddy = _rows_ld(dy)
    HW = rows // B
    ws = torch.empty(_lib.size_query("adap_groupnorm_workspace_floats", B, HW, C), device=x.device, dtype=F32)
    acc = 0
    dx32 = None
    add_ptr, ldadd = 0, 0
    if accumulate_into is not None:
        dx32, acc = accumulate_into, 1
    elif add_from is not None:
        assert add_from.dtype == F32 and add_from.numel() == x.numel()
        dx32, acc = torch.empty(x.shape, device=x.device, dtype=F32), 1
        add_ptr, ldadd = add_from.data_ptr(), _rows_ld(add_from)[1]
    elif out_f32:
        dx32 = torch.empty(x.shape, device=x.device, dtype=F32)
    lddx32 = _rows_ld(dx32)[1] if dx32 is not None else 0
    dx16 = torch.empty(x.shape, device=x.device, dtype=BF16) if out_bf16 else None
    _lib.call("adap_groupnorm_bwd", dy.data_ptr(), _dt(dy), lddy, x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), beta.data_ptr(),
              mean.data_ptr(), rstd.data_ptr(), _ptr(dx32), lddx32, acc, _ptr(dx16), C, add_ptr, ldadd, ws.data_ptr(),
              gn_sync_buffer(x.device), B, HW, C, int(act), _stream())
    return dx32, dx16


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    assert x.dtype == F32
    rows, ldx = _rows_ld(x)
    D = x.shape[-1]
    y = torch.empty(x.shape, device=x.device, dtype=BF16)
    mean = torch.empty(rows, device=x.device, dtype=F32)
    rstd = torch.empty(rows, device=x.device, dtype=F32)
    _lib.call("adap_layernorm_fwd", x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), D, mean.data_ptr(),
              rstd.data_ptr(), rows, D, float(eps), _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, accumulate_into=None, want_bf16=False):
    """-> dx f32 (accumulated into ``accumulate_into`` when given) [, bf16 copy of the final dx]"""
    assert dy.dtype == F32 and x.dtype == F32
    rows, ldx = _rows_ld(x)
    _, lddy = _rows_ld(dy)
    D = x.shape[-1]
    if accumulate_into is None:
        dx, acc = torch.empty(x.shape, device=x.device, dtype=F32), 0
    else:
        dx, acc = accumulate_into, 1
    dx16 = torch.empty(x.shape, device=x.device, dtype=BF16) if want_bf16 else None
    _lib.call("adap_layernorm_bwd", dy.data_ptr(), lddy, x.data_ptr(), ldx, gamma.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), dx.data_ptr(), _rows_ld(dx)[1], acc, _ptr(dx16), D, rows, D, _stream())
    return (dx, dx16) if want_bf16 else dx


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------

def gather_rows_bf16(src, idx, out=None):
    """src [B, R, C] bf16 (rows may be strided), idx [B, n] int32 -> [B, n, C]: out[b, i] = src[b, idx[b, i]]."""
    B, R, C = src.shape
    n = idx.shape[1]
    assert src.dtype == BF16 and idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape[0] == B
    if out is None:
        out = torch.empty(B, n, C, device=src.device, dtype=BF16)
    assert out.dtype == BF16 and tuple(out.shape) == (B, n, C)
    _lib.call("adap_gather_rows_bf16", src.data_ptr(), _rows_ld(src)[1], idx.data_ptr(), out.data_ptr(), _rows_ld(out)[1], B, R, n, C,
              _stream())
    return out


def attention_fwd(q, k, v, heads, key_mask=None, scale=None, key_count=None):
    """q [B,N,C], k/v [B,M,C] bf16 -> out [B,N,C] bf16, lse [B,H,N] f32.  ``scale``: the score scale (default d^-1/2).
    ``key_count`` [B] int32: sample b attends to its first key_count[b] keys only (compacted keys)."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    assert q.dtype == BF16 and k.dtype == BF16 and v.dtype == BF16
    out = torch.empty(B, N, C, device=q.device, dtype=BF16)
    lse = torch.empty(B, heads, N, device=q.device, dtype=F32)
    if key_mask is not None:
        assert key_mask.dtype == torch.uint8 and key_mask.shape == (B, M) and key_mask.is_contiguous()
    e0 = TIMER.start() if TIMER is not None else None
    if key_count is not None:
        assert key_count.dtype == torch.int32 and key_count.shape == (B,) and key_count.is_contiguous()
    _lib.call("adap_attention_fwd", q.data_ptr(), _rows_ld(q)[1], k.data_ptr(), _rows_ld(k)[1], v.data_ptr(),
              _rows_ld(v)[1], _ptr(key_mask), _ptr(key_count), out.data_ptr(), C, lse.data_ptr(), B, heads, N, M, d,
              float(d) ** -0.5 if scale is None else float(scale), _stream())
    if e0 is not None:
        TIMER.stop("attention_fwd", 4.0 * B * heads * N * M * d, e0, f"N={N} M={M} d={d}")      # QK^T + PV, SURVEY.md 8d
    return out, lse


def attention_tokmap_prep(d_tokmap, tok_w, q, k, heads):
    """the dq / dk-independent half of the token maps' backward (kw = w^T K, gq = d_tokmap^T Q) -> the workspace tensor that
    ``attention_bwd(..., tok=(d_tokmap, tok_w, ws))`` folds into its epilogues.  May run on another stream."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    G = tok_w.shape[2]
    assert d_tokmap.dtype == F32 and d_tokmap.is_contiguous() and tuple(d_tokmap.shape) == (B, heads, N, G)
    assert tok_w.dtype == F32 and tok_w.is_contiguous() and tuple(tok_w.shape) == (B, M, G)
    ws = torch.empty(_lib.size_query("adap_attention_tokmap_prep_workspace_floats", B, heads, N, d, G), device=q.device, dtype=F32)
    _lib.call("adap_attention_tokmap_prep", d_tokmap.data_ptr(), tok_w.data_ptr(), q.data_ptr(), _rows_ld(q)[1], k.data_ptr(),
              _rows_ld(k)[1], ws.data_ptr(), B, heads, N, M, d, G, _stream())
    return ws


def attention_bwd(q, k, v, out, dout, lse, heads, key_mask=None, dq=None, dk=None, dv=None, out_dtype=BF16, key_count=None,
                  scale=None, tok=None):
    """dq/dk/dv may be caller-provided pixel-major views (e.g. slices of one fused [B,N,3C] buffer);
    otherwise fresh tensors of ``out_dtype`` are allocated.  ``scale``: the score scale the forward used (default d^-1/2).
    ``tok`` = (d_tokmap, tok_w, attention_tokmap_prep's workspace): the token maps' gradient is added into dq / dk inside the
    kernels' epilogues."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    assert dout.dtype == BF16 and out.dtype == BF16
    dev = q.device
    delta = torch.empty(_lib.size_query("adap_attention_bwd_workspace_floats", B, heads, N, M, d), device=dev, dtype=F32)
    dq = torch.empty(B, N, C, device=dev, dtype=out_dtype) if dq is None else dq
    dk = torch.empty(B, M, C, device=dev, dtype=out_dtype) if dk is None else dk
    dv = torch.empty(B, M, C, device=dev, dtype=out_dtype) if dv is None else dv

    def pp(t):
        return (0, t.data_ptr()) if t.dtype == BF16 else (t.data_ptr(), 0)
    (dq32, dq16), (dk32, dk16), (dv32, dv16) = pp(dq), pp(dk), pp(dv)
    args = (q.data_ptr(), _rows_ld(q)[1], k.data_ptr(), _rows_ld(k)[1], v.data_ptr(),
            _rows_ld(v)[1], _ptr(key_mask), _ptr(key_count), out.data_ptr(), _rows_ld(out)[1], dout.data_ptr(),
            _rows_ld(dout)[1], lse.data_ptr(), delta.data_ptr(),
            dq32, dq16, _rows_ld(dq)[1], dk32, dk16, _rows_ld(dk)[1], dv32, dv16, _rows_ld(dv)[1],
            B, heads, N, M, d, float(d) ** -0.5 if scale is None else float(scale))
    if tok is None:
        _lib.call("adap_attention_bwd", *args, _stream())
    else:
        d_tokmap, tok_w, prep = tok
        _lib.call("adap_attention_bwd_tok", *args, d_tokmap.data_ptr(), tok_w.data_ptr(), prep.data_ptr(), tok_w.shape[2], _stream())
    return dq, dk, dv


def attention_capture(q, k, heads, want_q=True, tok_w=None, dense=True):
    """side outputs of attention.py:245-255 -> (attnscore, attn, q_scaled[, tokmap]).  ``tok_w`` f32 [B, M, G]: also
    the per-head token maps [B, heads, N, G] = attnscore . tok_w (what the cross-layer consistency loss reads).
    ``dense=False`` (needs ``tok_w``): only the token maps are formed -- the three dense tensors (86 MB per 64 x 64 layer
    at bs 4) are neither computed to memory nor returned (None)."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    dev = q.device
    assert dense or tok_w is not None
    score = torch.empty(B, heads, N, M, device=dev, dtype=F32) if dense else None
    prob = torch.empty(B, heads, N, M, device=dev, dtype=F32) if dense else None
    qs = torch.empty(B, heads, N, d, device=dev, dtype=F32) if (want_q and dense) else None
    tokmap, G = None, 0
    if tok_w is not None:
        assert tok_w.dtype == F32 and tok_w.is_contiguous() and tok_w.shape[:2] == (B, M) and tok_w.shape[2] <= 4, tok_w.shape
        G = tok_w.shape[2]
        tokmap = torch.empty(B, heads, N, G, device=dev, dtype=F32)
    _lib.call("adap_attention_capture", q.data_ptr(), _rows_ld(q)[1], k.data_ptr(), _rows_ld(k)[1], _ptr(score),
              _ptr(prob), _ptr(qs), _ptr(tok_w), _ptr(tokmap), G, B, heads, N, M, d, float(d) ** -0.5, _stream())
    return (score, prob, qs) if tok_w is None else (score, prob, qs, tokmap)


def attention_tokmap_bwd(d_tokmap, tok_w, q, k, dq, dk, heads):
    """add the gradient of the token maps (``attention_capture(..., tok_w)``) into the layer's bf16 dq / dk."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    G = tok_w.shape[2]
    assert d_tokmap.dtype == F32 and d_tokmap.is_contiguous() and tuple(d_tokmap.shape) == (B, heads, N, G)
    assert dq.dtype == BF16 and dk.dtype == BF16 and dq.shape == q.shape and dk.shape == k.shape
    ws = torch.empty(_lib.size_query("adap_attention_tokmap_bwd_workspace_floats", B, heads, N, d, G), device=q.device, dtype=F32)
    _lib.call("adap_attention_tokmap_bwd", d_tokmap.data_ptr(), tok_w.data_ptr(), q.data_ptr(), _rows_ld(q)[1], k.data_ptr(),
              _rows_ld(k)[1], dq.data_ptr(), _rows_ld(dq)[1], dk.data_ptr(), _rows_ld(dk)[1], ws.data_ptr(), B, heads, N, M, d,
              G, float(d) ** -0.5, _stream())


def attention_capture_bwd(d_score, d_qs, q, k, dq, dk, heads):
    """add the gradients of the captured side outputs into the bf16 ``dq`` [B,N,C] / ``dk`` [B,M,C] of the same layer:
    d_score f32 [B,h,N,M] (gradient of attnscore) and / or d_qs f32 [B,h,N,d] (gradient of q * d^-1/4)."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    assert dq.dtype == BF16 and dq.shape == q.shape and (d_score is None or (dk is not None and dk.dtype == BF16))
    for t, shp in ((d_score, (B, heads, N, M)), (d_qs, (B, heads, N, d))):
        assert t is None or (t.dtype == F32 and tuple(t.shape) == shp and t.is_contiguous()), (None if t is None else t.shape, shp)
    ws = None
    if d_score is not None:
        ws = torch.empty(_lib.size_query("adap_attention_capture_bwd_workspace_floats", B, heads, N, M, d), device=q.device,
                         dtype=F32)
    _lib.call("adap_attention_capture_bwd", _ptr(d_score), _ptr(d_qs), q.data_ptr(), _rows_ld(q)[1], k.data_ptr(),
              _rows_ld(k)[1], dq.data_ptr(), _rows_ld(dq)[1], _ptr(dk), 0 if dk is None else _rows_ld(dk)[1], _ptr(ws), B, heads,
              N, M, d, float(d) ** -0.5, _stream())


# --------------------------------------------------------------------------------------------
# misc
# --------------------------------------------------------------------------------------------

def geglu_fwd(h):
    rows, ldh = _rows_ld(h)
    inner = h.shape[-1] // 2
    out = torch.empty(tuple(h.shape[:-1]) + (inner,), device=h.device, dtype=BF16)
    _lib.call("adap_geglu_fwd", h.data_ptr(), ldh, out.data_ptr(), inner, rows, inner, _stream())
    return out


def act_fwd(x, kind):
    """x f32 [..., C] -> bf16: ``kind`` "quick_gelu" | "gelu" (the CLIP vision MLP's activation)."""
    assert x.dtype == F32
    rows, ldx = _rows_ld(x)
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
    _lib.call("adap_act_fwd", x.data_ptr(), ldx, out.data_ptr(), x.shape[-1], rows, x.shape[-1],
              {"quick_gelu": 0, "gelu": 1}[kind], _stream())
    return out


def geglu_bwd(dout, h):
    rows, ldh = _rows_ld(h)
    inner = h.shape[-1] // 2
    dh = torch.empty(h.shape, device=h.device, dtype=BF16)
    _lib.call("adap_geglu_bwd", dout.data_ptr(), _rows_ld(dout)[1], h.data_ptr(), ldh, dh.data_ptr(), 2 * inner, rows,
              inner, _stream())
    return dh


def linear_small(x, w, bias=None, pre_silu=False, post_silu=False):
    """exact f32, x [R, K] -> [R, N]; the kernel keeps up to 8 rows in registers per pass over the weights, larger
    batches (the sampler's 16) go through in slices of 8"""
    R, K = x.shape
    N = w.shape[0]
    assert x.dtype == F32 and w.dtype == F32 and w.is_contiguous() and x.stride(1) == 1
    y = torch.empty(R, N, device=x.device, dtype=F32)
    for r0 in range(0, R, 8):
        r = min(8, R - r0)
        _lib.call("adap_linear_small", x[r0:].data_ptr(), x.stride(0), w.data_ptr(), _ptr(bias), y[r0:].data_ptr(), N, r, K,
                  N, int(pre_silu), int(post_silu), _stream())
    return y


def timestep_embedding(t, dim):
    t = t.to(torch.int64).contiguous()
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=F32)
    _lib.call("adap_timestep_embedding", t.data_ptr(), out.data_ptr(), t.shape[0], dim, _stream())
    return out


def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac):
    x0, noise = x0.contiguous(), noise.contiguous()
    t = t.to(torch.int64).contiguous()
    out = torch.empty_like(x0)
    B = x0.shape[0]
    assert t.numel() == B and sqrt_ac.numel() == sqrt_1mac.numel()
    _lib.call("adap_q_sample", x0.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(),
              out.data_ptr(), B, x0.numel() // B, sqrt_ac.numel(), _stream())
    return out


def posterior_sample(moments, noise, scale):
    """moments [..., 2Z] pixel-major f32, noise [..., Z] -> z [..., Z]"""
    rows, ldm = _rows_ld(moments)
    Z = moments.shape[-1] // 2
    noise = noise.contiguous()
    z = torch.empty_like(noise)
    _lib.call("adap_posterior_sample", moments.data_ptr(), ldm, noise.data_ptr(), z.data_ptr(), rows, Z, float(scale),
              _stream())
    return z


def masked_mse(out, tgt, img_mask, fg_mask, w_fg, w_bg, want_grad=True):
    """out/tgt [B,H,W,C] f32 contiguous, masks [B,H,W] f32 or None -> (loss[1], grad or None)"""
    out, tgt = out.contiguous(), tgt.contiguous()
    C = out.shape[-1]
    P = out.numel() // C
    loss = torch.empty(1, device=out.device, dtype=F32)
    grad = torch.empty_like(out) if want_grad else None
    if img_mask is not None:
        img_mask = img_mask.contiguous().float()
        assert img_mask.numel() == P
    if fg_mask is not None:
        fg_mask = fg_mask.contiguous().float()
        assert fg_mask.numel() == P
    _lib.call("adap_masked_mse", out.data_ptr(), tgt.data_ptr(), _ptr(img_mask), _ptr(fg_mask), float(w_fg), float(w_bg),
              P, C, loss.data_ptr(), _ptr(grad), _stream())
    return loss, grad


def concat2(a, b):
    ra, lda = _rows_ld(a)
    rb, ldb = _rows_ld(b)
    assert ra == rb and a.dtype == F32 and b.dtype == F32
    Ca, Cb = a.shape[-1], b.shape[-1]
    out = torch.empty(tuple(a.shape[:-1]) + (Ca + Cb,), device=a.device, dtype=F32)
    _lib.call("adap_concat2", a.data_ptr(), lda, Ca, b.data_ptr(), ldb, Cb, out.data_ptr(), Ca + Cb, ra, _stream())
    return out


def sumpool2x2(x):
    B, H2, W2, C = x.shape
    assert x.is_contiguous() and x.dtype == F32
    out = torch.empty(B, H2 // 2, W2 // 2, C, device=x.device, dtype=F32)
    _lib.call("adap_sumpool2x2", x.data_ptr(), out.data_ptr(), B, H2 // 2, W2 // 2, C, _stream())
    return out


def pad_cast_bf16(x, Cout=None):
    rows, ldi = _rows_ld(x)
    Cin = x.shape[-1]
    Cout = Cin if Cout is None else Cout
    out = torch.empty(tuple(x.shape[:-1]) + (Cout,), device=x.device, dtype=BF16)
    _lib.call("adap_pad_cast_bf16", x.data_ptr(), ldi, Cin, out.data_ptr(), Cout, Cout, rows, _stream())
    return out


def upsample2x_bf16(x):
    """x f32 [B,H,W,C] -> bf16 [B,2H,2W,C], nearest neighbour (``adap_upsample2x_bf16``)."""
    B, H, W, C = x.shape
    assert x.dtype == F32 and x.stride(-1) == 1
    _, ldi = _rows_ld(x)
    out = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=BF16)
    _lib.call("adap_upsample2x_bf16", x.data_ptr(), ldi, out.data_ptr(), B, H, W, C, _stream())
    return out


def transpose_bf16(x):
    G, R, C = x.shape
    assert x.is_contiguous() and x.dtype == BF16
    out = torch.empty(G, C, R, device=x.device, dtype=BF16)
    _lib.call("adap_transpose_bf16", x.data_ptr(), out.data_ptr(), G, R, C, _stream())
    return out


def vae_softmax(S, scale, pixel_class=None):
    """S f32 [G, N, N] -> P bf16 [G, N, N] with the post-softmax hetero-pair zero fill."""
    G, N, N2 = S.shape
    assert S.is_contiguous() and S.dtype == F32
    P = torch.empty(G, N, N2, device=S.device, dtype=BF16)
    if pixel_class is not None:
        assert pixel_class.dtype == torch.uint8 and pixel_class.shape == (G, N2) and pixel_class.is_contiguous()
    _lib.call("adap_vae_softmax", S.data_ptr(), N2, P.data_ptr(), N2, _ptr(pixel_class), G * N, N2, N, float(scale),
              _stream())
    return P


def axpy_(y, x, a=1.0):
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == F32 and y.dtype == F32
    _lib.call("adap_axpy", x.data_ptr(), y.data_ptr(), float(a), x.numel(), _stream())
    return y


def add2(a, b, want_bf16=True):
    """a + b for two f32 pixel-major tensors of one shape (arbitrary leading dimensions) -> (f32, bf16 | None), packed."""
    assert a.dtype == F32 and b.dtype == F32 and a.shape == b.shape
    C = a.shape[-1]
    rows, lda = _rows_ld(a)
    _, ldb = _rows_ld(b)
    y32 = torch.empty(a.shape, device=a.device, dtype=F32)
    y16 = torch.empty(a.shape, device=a.device, dtype=BF16) if want_bf16 else None
    _lib.call("adap_add2", a.data_ptr(), lda, b.data_ptr(), ldb, y32.data_ptr(), _ptr(y16), rows, C, _stream())
    return y32, y16


# --------------------------------------------------------------------------------------------
# weight gradients (unfreeze_model: True)
# --------------------------------------------------------------------------------------------

def _byte_ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), device=device, dtype=torch.uint8)


def conv2d_bwd_weight(x, dy, dw, dbias=None, KH=1, stride=1, pad=0, up=0, accumulate=True):
    """dw f32 [Cout, Cin, KH, KH] (+)= the weight gradient of conv2d(x) -> dy; dbias f32 [Cout] (+)= sum of dy.
    x [B,H,W,Cin], dy [B,Ho,Wo,Cout] pixel-major, f32 or bf16.  ``dw`` None: bias gradient only."""
    assert x.dim() == 4 and dy.dim() == 4 and x.shape[0] == dy.shape[0]
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    xr, ldx = _rows_ld(x)
    yr, ldy = _rows_ld(dy)
    assert xr == B * H * W and yr == B * Ho * Wo
    if dw is not None:
        assert dw.dtype == F32 and dw.is_contiguous() and dw.numel() == Cout * Cin * KH * KH, (dw.shape, Cout, Cin, KH)
    if dbias is not None:
        assert dbias.dtype == F32 and dbias.is_contiguous() and dbias.numel() == Cout
    nb = _lib.size_query("adap_conv2d_bwd_weight_workspace_bytes", B, Ho, Wo, Cin, Cout, KH, KH)
    assert nb >= 0, "conv2d_bwd_weight: problem too large"
    ws = _byte_ws(nb, x.device)
    _lib.call("adap_conv2d_bwd_weight", x.data_ptr(), _dt(x), ldx, dy.data_ptr(), _dt(dy), ldy, _ptr(dw), _ptr(dbias),
              B, H, W, Cin, Ho, Wo, Cout, KH, KH, stride, pad, up, int(bool(accumulate)), ws.data_ptr(), ws.numel(), _stream())


def linear_bwd_weight(x, dy, dw, dbias=None, accumulate=True):
    """dw f32 [O, I] (+)= dy^T x over all rows; x [..., I], dy [..., O] (f32 or bf16, channel dim contiguous)."""
    xr, ldx = _rows_ld(x)
    yr, ldy = _rows_ld(dy)
    assert xr == yr, (x.shape, dy.shape)
    I, O = x.shape[-1], dy.shape[-1]
    x4 = x.as_strided((1, xr, 1, I), (xr * ldx, ldx, ldx, 1))
    y4 = dy.as_strided((1, yr, 1, O), (yr * ldy, ldy, ldy, 1))
    conv2d_bwd_weight(x4, y4, dw, dbias, 1, 1, 0, 0, accumulate)


def colsum(dy, out, seg_rows=None, accumulate=True):
    """out f32 [nseg, C] (+)= per-segment column sums of dy [..., C] (segments of ``seg_rows`` consecutive rows)."""
    rows, ld = _rows_ld(dy)
    C = dy.shape[-1]
    seg_rows = rows if seg_rows is None else seg_rows
    assert rows % seg_rows == 0 and out.dtype == F32 and out.is_contiguous() and out.numel() == rows // seg_rows * C
    ws = torch.empty(_lib.size_query("adap_colsum_workspace_floats", rows, seg_rows, C), device=dy.device, dtype=F32)
    _lib.call("adap_colsum", dy.data_ptr(), _dt(dy), ld, rows, seg_rows, C, out.data_ptr(), int(bool(accumulate)),
              ws.data_ptr(), _stream())


def norm_affine_bwd(dy, x, gamma, beta, mean, rstd, kind, act, dgamma, dbeta, accumulate=True):
    """dgamma / dbeta f32 [C] (+)= the affine gradients of GroupNorm32 (kind 0; x [B,...,C], mean/rstd [B,32]) or
    LayerNorm (kind 1; mean/rstd [rows]); ``dy`` is the gradient of the norm's output, after SiLU when act == 1."""
    rows, ldx = _rows_ld(x)
    yr, ldy = _rows_ld(dy)
    C = x.shape[-1]
    assert rows == yr and dy.shape[-1] == C
    HW = rows // x.shape[0] if kind == 0 else 1
    for t in (dgamma, dbeta):
        assert t is None or (t.dtype == F32 and t.is_contiguous() and t.numel() == C)
    ws = torch.empty(_lib.size_query("adap_colsum_workspace_floats", rows, rows, C), device=x.device, dtype=F32)
    _lib.call("adap_norm_affine_bwd", dy.data_ptr(), _dt(dy), ldy, x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), _ptr(beta),
              mean.data_ptr(), rstd.data_ptr(), kind, act, _ptr(dgamma), _ptr(dbeta), int(bool(accumulate)), ws.data_ptr(),
              rows, HW, C, _stream())


def cosine_rows(x, r, demean, align, ref_grad_scale=1.0, gl=None, want_dx=True, want_dr=True, exponent=2):
    """x, r f32 [R, D] (rows contiguous).  gl None: -> loss [R] (ldm/util.py:437-535 per-row term, exponent 1 / 2 / 3).
    gl f32 [R]: -> (dx, dr), the gradients of sum(gl * loss)."""
    assert x.dtype == F32 and r.dtype == F32 and x.dim() == 2 and x.shape == r.shape and x.stride(1) == 1 and r.stride(1) == 1
    R, D = x.shape
    if gl is None:
        loss = torch.empty(R, device=x.device, dtype=F32)
        _lib.call("adap_cosine_rows", x.data_ptr(), x.stride(0), r.data_ptr(), r.stride(0), 0, loss.data_ptr(), 0, 0, 0, 0, R, D,
                  int(bool(demean)), int(bool(align)), float(ref_grad_scale), int(exponent), _stream())
        return loss
    assert gl.dtype == F32 and gl.numel() == R and gl.is_contiguous()
    dx = torch.empty(R, D, device=x.device, dtype=F32) if want_dx else None
    dr = torch.empty(R, D, device=x.device, dtype=F32) if want_dr else None
    _lib.call("adap_cosine_rows", x.data_ptr(), x.stride(0), r.data_ptr(), r.stride(0), gl.data_ptr(), 0, _ptr(dx), D, _ptr(dr), D,
              R, D, int(bool(demean)), int(bool(align)), float(ref_grad_scale), int(exponent), _stream())
    return dx, dr


EM_TOK_ROWS, EM_OUT_FLOATS, EM_COEF_ROWS = 11, 8, 10          # include/adaprompt_hip.h ADAP_EM_*
EM_SC_BELOW, EM_MC_BELOW = 8, 9


def elastic_match_fwd(q, f, fg, cutoff):
    """q f32 [4, Cq, N], f f32 [4, Cf, N] (subject single / subject comp / mix single / mix comp of one instance), fg f32 [N]
    -> (P2 [2, N, N], RT [Cf, N], tok [EM_TOK_ROWS, N], out [EM_OUT_FLOATS]): ldm/util.py:2241-2368 in fixed-order f32 sums
    (csrc/stage2loss.hip).  out[:3] = (map_align, sc_ss_fg, sc_mc_bg); tok[EM_SC_BELOW], tok[EM_MC_BELOW] the weight vectors."""
    assert q.dtype == F32 and f.dtype == F32 and fg.dtype == F32 and q.is_contiguous() and f.is_contiguous() and fg.is_contiguous()
    assert q.dim() == 3 and f.dim() == 3 and q.shape[0] == 4 and f.shape[0] == 4 and q.shape[2] == f.shape[2] == fg.numel()
    Cq, N, Cf = q.shape[1], q.shape[2], f.shape[1]
    P2 = torch.empty(2, N, N, device=q.device, dtype=F32)
    RT = torch.empty(Cf, N, device=q.device, dtype=F32)
    tok = torch.empty(EM_TOK_ROWS, N, device=q.device, dtype=F32)
    out = torch.empty(EM_OUT_FLOATS, device=q.device, dtype=F32)
    _lib.call("adap_elastic_match_fwd", q.data_ptr(), Cq, f.data_ptr(), Cf, fg.data_ptr(), N, float(cutoff), P2.data_ptr(),
              RT.data_ptr(), tok.data_ptr(), out.data_ptr(), _stream())
    return P2, RT, tok, out


def elastic_match_bwd(q, f, fg, cutoff, gs_q, gs_feat, gs_mix, P2, RT, tok, out, g_map, g_fg, g_bg, g_scb, g_mcb):
    """gradients of sum(g * outputs) of ``elastic_match_fwd`` (any g may be None = zero) -> (dq [4, Cq, N], df [4, Cf, N])."""
    Cq, N, Cf = q.shape[1], q.shape[2], f.shape[1]
    for g in (g_map, g_fg, g_bg):
        assert g is None or (g.dtype == F32 and g.numel() == 1)
    for g in (g_scb, g_mcb):
        assert g is None or (g.dtype == F32 and g.numel() == N and g.is_contiguous())
    dS2 = torch.empty(2, N, N, device=q.device, dtype=F32)
    dRT = torch.empty(Cf, N, device=q.device, dtype=F32)
    coef = torch.empty(EM_COEF_ROWS, N, device=q.device, dtype=F32)
    dq, df = torch.empty_like(q), torch.empty_like(f)
    _lib.call("adap_elastic_match_bwd", q.data_ptr(), Cq, f.data_ptr(), Cf, fg.data_ptr(), N, float(cutoff), float(gs_q),
              float(gs_feat), float(gs_mix), P2.data_ptr(), RT.data_ptr(), tok.data_ptr(), out.data_ptr(), _ptr(g_map), _ptr(g_fg),
              _ptr(g_bg), _ptr(g_scb), _ptr(g_mcb), dS2.data_ptr(), dRT.data_ptr(), coef.data_ptr(), dq.data_ptr(), df.data_ptr(),
              _stream())
    return dq, df


PM_REC = 16           # include/adaprompt_hip.h ADAP_PM_REC


def promptmix_attn_terms(a, gs_mix, rec=None, g_delta=None, g_norm=None):
    """a f32 [4, H, N] (subject single / subject comp / mix single / mix comp subject score maps of one instance).
    rec None: forward -> (out [2] = (subj_attn_delta_align, subj_attn_norm_distill) of the layer, rec).  rec given: backward for
    the two incoming gradients (f32 scalars or None) -> da [4, H, N]  (ddpm.py:3714-3930, csrc/stage2loss.hip)."""
    assert a.dtype == F32 and a.dim() == 3 and a.shape[0] == 4 and a.is_contiguous()
    H, N = a.shape[1], a.shape[2]
    if rec is None:
        rec = torch.empty(H, PM_REC, device=a.device, dtype=F32)
        out = torch.empty(2, device=a.device, dtype=F32)
        _lib.call("adap_promptmix_attn_terms", a.data_ptr(), H, N, float(gs_mix), rec.data_ptr(), out.data_ptr(), 0, 0, 0, _stream())
        return out, rec
    for g in (g_delta, g_norm):
        assert g is None or (g.dtype == F32 and g.numel() == 1)
    da = torch.empty_like(a)
    _lib.call("adap_promptmix_attn_terms", a.data_ptr(), H, N, float(gs_mix), rec.data_ptr(), 0, _ptr(g_delta), _ptr(g_norm),
              da.data_ptr(), _stream())
    return da


def attn_spatial_weight(a0, a1=None, reversed=True):
    """a0 (and a1) f32 [H, N] subject score maps of one instance at the feature map's resolution -> f32 [N]: the mean-1 spatial
    weight of ldm/util.py:1718 convert_attn_to_spatial_weight, averaged over the sources."""
    assert a0.dtype == F32 and a0.dim() == 2 and a0.is_contiguous()
    assert a1 is None or (a1.dtype == F32 and a1.shape == a0.shape and a1.is_contiguous())
    H, N = a0.shape
    sw = torch.empty(N, device=a0.device, dtype=F32)
    _lib.call("adap_attn_spatial_weight", a0.data_ptr(), _ptr(a1), H, N, int(bool(reversed)), sw.data_ptr(), _stream())
    return sw


def bg_suppress(a, scb, mcb, gs_mix, saved=None, g_s=None, g_m=None):
    """a f32 [4, H, N] pooled subject score maps, scb / mcb f32 [N].  saved None: forward -> (out [4], col [2, N]); out[:2] =
    (comp_subj_bg_attn_suppress, comp_mix_bg_attn_suppress) of the layer.  saved = (out, col): backward -> (da, dscb, dmcb)."""
    assert a.dtype == F32 and a.dim() == 3 and a.shape[0] == 4 and a.is_contiguous()
    H, N = a.shape[1], a.shape[2]
    for v in (scb, mcb):
        assert v.dtype == F32 and v.numel() == N and v.is_contiguous()
    if saved is None:
        out = torch.empty(4, device=a.device, dtype=F32)
        col = torch.empty(2, N, device=a.device, dtype=F32)
        _lib.call("adap_bg_suppress", a.data_ptr(), scb.data_ptr(), mcb.data_ptr(), H, N, float(gs_mix), col.data_ptr(), out.data_ptr(),
                  0, 0, 0, 0, 0, _stream())
        return out, col
    out, col = saved
    for g in (g_s, g_m):
        assert g is None or (g.dtype == F32 and g.numel() == 1)
    da = torch.empty_like(a)
    dscb, dmcb = torch.empty(N, device=a.device, dtype=F32), torch.empty(N, device=a.device, dtype=F32)
    _lib.call("adap_bg_suppress", a.data_ptr(), scb.data_ptr(), mcb.data_ptr(), H, N, float(gs_mix), col.data_ptr(), out.data_ptr(),
              _ptr(g_s), _ptr(g_m), da.data_ptr(), dscb.data_ptr(), dmcb.data_ptr(), _stream())
    return da, dscb, dmcb


def ortho_rows(a, b, g=None, want_da=True, want_db=True):
    """a, b f32 [R, D] (rows contiguous).  g None: -> a - <a,b>/(<b,b>+1e-6) b per row (ldm/util.py:280 ortho_subtract).
    g f32 [R, D] = the gradient of the result: -> (da, db)."""
    assert a.dtype == F32 and b.dtype == F32 and a.dim() == 2 and a.shape == b.shape and a.stride(1) == 1 and b.stride(1) == 1
    R, D = a.shape
    nws = _lib.size_query("adap_ortho_rows_workspace_floats", R, D)          # > 0: few very long rows, sliced over two launches
    ws = torch.empty(nws, device=a.device, dtype=F32) if nws else None
    if g is None:
        out = torch.empty(R, D, device=a.device, dtype=F32)
        _lib.call("adap_ortho_rows_ws", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), 0, 0, out.data_ptr(), D, 0, 0, 0, 0,
                  R, D, _ptr(ws), _stream())
        return out
    assert g.dtype == F32 and g.shape == a.shape and g.stride(1) == 1
    da = torch.empty(R, D, device=a.device, dtype=F32) if want_da else None
    db = torch.empty(R, D, device=a.device, dtype=F32) if want_db else None
    _lib.call("adap_ortho_rows_ws", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), g.data_ptr(), g.stride(0), 0, 0, _ptr(da),
              D, _ptr(db), D, R, D, _ptr(ws), _stream())
    return da, db


def mask_hinges(maps, fmask, iw, margin, margin_bg_at_mf, have_bg, gout=None, ws=None):
    """maps f32 [L, B, H, N, G] contiguous (column 0 subject, column 1 background); fmask f32 [B, N] in {0,1}.
    gout None -> (out [4, L], workspace);  gout [4, L] + the forward's workspace -> d maps [L, B, H, N, G]."""
    assert maps.dtype == F32 and maps.is_contiguous() and maps.dim() == 5
    L, B, H, N, G = maps.shape
    assert fmask.dtype == F32 and fmask.is_contiguous() and tuple(fmask.shape) == (B, N) and (not have_bg or G >= 2)
    S = maps.data_ptr()
    Gp = S + 4 if have_bg else 0
    if gout is None:
        out = torch.empty(4, L, device=maps.device, dtype=F32)
        ws = torch.empty(_lib.size_query("adap_mask_hinges_workspace_floats", L, B), device=maps.device, dtype=F32)
        _lib.call("adap_mask_hinges_fwd", S, Gp, G, fmask.data_ptr(), _ptr(iw), out.data_ptr(), ws.data_ptr(), L, B, H, N,
                  float(margin), float(margin_bg_at_mf), _stream())
        return out, ws
    assert gout.dtype == F32 and gout.is_contiguous() and tuple(gout.shape) == (4, L)
    d = torch.zeros_like(maps) if G > (2 if have_bg else 1) else torch.empty_like(maps)
    _lib.call("adap_mask_hinges_bwd", S, Gp, G, fmask.data_ptr(), _ptr(iw), gout.data_ptr(), ws.data_ptr(), d.data_ptr(),
              d.data_ptr() + 4 if have_bg else 0, G, L, B, H, N, float(margin), float(margin_bg_at_mf), _stream())
    if not have_bg and G > 1:
        d[..., 1:].zero_()
    return d


# --------------------------------------------------------------------------------------------
# the recon iteration's attention regularisers on the captured token maps (csrc/regloss.hip)
# --------------------------------------------------------------------------------------------

_REG_TABLES = {}


def reg_losses(token_maps, complem_w, pairs, fg_mask, inst_w, Bk, have_bg, margin, margin_bg_at_mf, fg_grad_scale, coefs):
    """token_maps: list of L f32 [Bt, H, N_l, G] tensors (adap_attention_capture's token maps); complem_w: L floats (the layer's
    normalised complementary-loss weight, 0 = no part in it); pairs: tuple of (x layer idx, reference layer idx, weight) of the
    cross-layer consistency loss; fg_mask f32 [Bt, Hm, Hm] or None; inst_w f32 [Bk] or None; coefs = d total / d (L_fg, L_bg,
    L_complem, L_subj_mb_suppress, L_bg_mf_suppress, L_mask_contrast).
    -> (parts f32 [8] = the six losses, the total, 0;  [d total / d token_maps[l]])  -- values and gradients in one call."""
    L = len(token_maps)
    Bt, H, _, G = token_maps[0].shape
    Ns = tuple(int(t.shape[2]) for t in token_maps)
    for t in token_maps:
        assert t.dtype == F32 and t.is_cuda and t.is_contiguous() and t.shape[0] == Bt and t.shape[1] == H and t.shape[3] == G
    key = (Ns, tuple(float(w) for w in complem_w), tuple(pairs), int(Bk), int(H), int(G))
    tab = _REG_TABLES.get(key)
    if tab is None:
        if len(_REG_TABLES) > 16:
            _REG_TABLES.clear()
        ln = (ctypes.c_int * L)(*Ns)
        cw = (ctypes.c_float * L)(*[float(w) for w in complem_w])
        n = len(pairs)
        px = (ctypes.c_int * max(n, 1))(*[int(a) for a, _, _ in pairs])
        pr = (ctypes.c_int * max(n, 1))(*[int(b) for _, b, _ in pairs])
        pw = (ctypes.c_float * max(n, 1))(*[float(w) for _, _, w in pairs])
        nws = _lib.call_long("adap_reg_losses_workspace_floats", ln, L, px, pr, n, int(Bk), int(H), int(G))
        assert nws > 0, "adap_reg_losses_workspace_floats: unsupported configuration"
        tab = _REG_TABLES[key] = (ln, cw, px, pr, pw, n, int(nws))
    ln, cw, px, pr, pw, n, nws = tab
    offs, total = [], 0                 # (not part of the cached tables: the maps' batch size Bt is not in their key)
    for t in token_maps:
        offs.append(total)
        total += t.numel()
    dev = token_maps[0].device
    dflat = torch.empty(total, device=dev, dtype=F32)
    dtm = [dflat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, token_maps)]
    ws = torch.empty(nws, device=dev, dtype=F32)
    parts = torch.empty(8, device=dev, dtype=F32)
    Hm = 0
    if fg_mask is not None:
        assert fg_mask.dtype == F32 and fg_mask.is_contiguous() and fg_mask.dim() == 3 and fg_mask.shape[0] == Bt \
            and fg_mask.shape[1] == fg_mask.shape[2], fg_mask.shape
        Hm = fg_mask.shape[1]
    if inst_w is not None:
        assert inst_w.dtype == F32 and inst_w.is_contiguous() and inst_w.numel() >= Bk
    tp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in token_maps])
    dp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in dtm])
    _lib.call("adap_reg_losses", tp, dp, ln, cw, L, px, pr, pw, n, _ptr(fg_mask), Hm, _ptr(inst_w), Bt, int(Bk), H, G,
              int(bool(have_bg)), float(margin), float(margin_bg_at_mf), float(fg_grad_scale), *[float(c) for c in coefs],
              parts.data_ptr(), ws.data_ptr(), nws, _stream())
    return parts, dtm


def prompt_delta_loss(emb4, mask4, coef, cls_grad_scale=0.05):
    """emb4 f32 [4*Bs, L, T, D] (subject-single / subject-comp / class-single / class-comp static embeddings), mask4 f32
    [4*Bs, T, 1] or [4*Bs, T] (its start-token column is zeroed in place, as the reference does)
    -> (out f32 [2] = {loss, coef * loss}, d (coef * loss) / d emb4)."""
    assert emb4.dtype == F32 and emb4.is_cuda and emb4.is_contiguous() and emb4.dim() == 4 and emb4.shape[0] % 4 == 0
    Bs, L, T, D = emb4.shape[0] // 4, emb4.shape[1], emb4.shape[2], emb4.shape[3]
    assert mask4.dtype == F32 and mask4.is_contiguous() and mask4.numel() == 4 * Bs * T, mask4.shape
    demb = torch.empty_like(emb4)
    out = torch.empty(2, device=emb4.device, dtype=F32)
    ws = torch.empty(_lib.size_query("adap_prompt_delta_loss_workspace_floats", Bs, L, T), device=emb4.device, dtype=F32)
    _lib.call("adap_prompt_delta_loss", emb4.data_ptr(), demb.data_ptr(), mask4.data_ptr(), Bs, L, T, D, float(coef),
              float(cls_grad_scale), out.data_ptr(), ws.data_ptr(), _stream())
    return out, demb


# --------------------------------------------------------------------------------------------
# FeedForward with its GEGLU inside the two contractions (csrc/conv_gemm.hip, ConvParams::epi)
# --------------------------------------------------------------------------------------------

def geglu_row_permutation(C8, device):
    """perm [C8] int64: row r of the permuted ff.net.0.proj pack is row perm[r] of the checkpoint's weight (blocks of 16 value
    channels alternate with the blocks of their 16 gate channels)."""
    r = torch.arange(C8, device=device)
    blk, half, within = r // 32, (r % 32) // 16, r % 16
    return blk * 16 + within + half * (C8 // 2)


def linear_geglu_fwd(x16, pk_perm):
    """x16 bf16 [..., Cin], pk_perm: PackedConv of ff.net.0.proj with rows in ``geglu_row_permutation`` order
    -> (h bf16 [..., 8C] permuted order (for the backward), a * gelu(gate) bf16 [..., 4C])."""
    assert x16.dtype == BF16
    rows, ldx = _rows_ld(x16)
    C8 = pk_perm.O4
    h = torch.empty(tuple(x16.shape[:-1]) + (C8,), device=x16.device, dtype=BF16)
    out = torch.empty(tuple(x16.shape[:-1]) + (C8 // 2,), device=x16.device, dtype=BF16)
    _lib.call("adap_linear_geglu_fwd", x16.data_ptr(), ldx, pk_perm.fwd.data_ptr(), _ptr(pk_perm.bias), h.data_ptr(), C8,
              out.data_ptr(), C8 // 2, rows, x16.shape[-1], C8, _stream())
    return h, out


def linear_geglu_bwd(g16, pk_ff2, h):
    """g16 bf16 [..., C] = d(ff.net.2 output); pk_ff2: PackedConv of ff.net.2; h: the permuted pre-activation of the forward
    -> dh bf16 [..., 8C] (permuted order: the operand of the permuted ff.net.0.proj data-gradient pack)."""
    assert g16.dtype == BF16 and h.dtype == BF16 and h.is_contiguous()
    rows, ldg = _rows_ld(g16)
    C8 = h.shape[-1]
    bw = pk_ff2.bwd                     # [1][4C][C]
    assert bw.shape[1] == C8 // 2 and bw.shape[2] == g16.shape[-1]
    dh = torch.empty(h.shape, device=h.device, dtype=BF16)
    _lib.call("adap_linear_geglu_bwd", g16.data_ptr(), ldg, bw.data_ptr(), h.data_ptr(), C8, dh.data_ptr(), C8, rows,
              g16.shape[-1], C8 // 2, _stream())
    return dh

"""Block-level autograd Functions of the UNet: each one sequences HIP kernels for the forward
and, explicitly, for the backward.  In the shipped config the UNet is frozen (``unfreeze_model:
False`` v1-finetune-ada.yaml:26 / ddpm.py:775-786): gradients flow only to the inputs --
ultimately to the layerwise text context that the subject-basis generators produced -- and, through
the captured cross-attention scores, from the recon iteration's consistency loss (ddpm.py:3246-3270).
With ``unfreeze_model: True`` the same backward functions also add the weight gradients into
``param.grad`` (``train_of`` / ``_dw_*``).  torch.autograd is used as the tape between blocks;
inside a block nothing is left to it: residual adds, time-embedding adds, bias adds and gradient
accumulation are fused into kernel epilogues.

All activations are pixel-major f32 (the residual stream) or bf16 (matrix-core operands)."""
import os

import torch

from . import ops

F32 = torch.float32
BF16 = torch.bfloat16

# Validation mode (``ADAP_F32_STORAGE=1`` or ``set_f32_storage(True)``): bf16 is used ONLY as the matrix-core operand
# format.  Every tensor the shipped path keeps in HBM as bf16 although its consumer is not a contraction -- the ResBlock's
# h1 (read by the second GroupNorm), the GEGLU pre-activation, the data gradients that feed a GroupNorm / GEGLU backward,
# the read-modify-write of the captured-activation gradients into dq/dk -- is f32 instead, and rounded to bf16 exactly
# once, where a contraction reads it.  It exists to put a number on what bf16 *storage* (as opposed to bf16 MFMA operands)
# costs in accuracy (tests/test_precision_gpu.py); it runs a few torch element-wise kernels and is not a speed path.
F32_STORAGE = os.environ.get("ADAP_F32_STORAGE", "0") == "1"


def set_f32_storage(on):
    global F32_STORAGE
    F32_STORAGE = bool(on)


def _geglu_fwd_f32(hh):
    a, gate = hh.chunk(2, dim=-1)
    return (a * torch.nn.functional.gelu(gate)).to(BF16)


def _geglu_bwd_f32(dout, hh):
    a, gate = hh.chunk(2, dim=-1)
    cdf = 0.5 * (1.0 + torch.erf(gate * 0.7071067811865476))
    pdf = torch.exp(-0.5 * gate * gate) * 0.3989422804014327
    return torch.cat([dout * (gate * cdf), dout * a * (cdf + gate * pdf)], dim=-1).to(BF16)


def _op16(t):
    """the bf16 operand of a contraction (validation mode hands f32 tensors around)."""
    return t if t is None or t.dtype == BF16 else t.to(BF16)


# ---------------------------------------------------------------------------------------------
# The side lane: a second HIP stream per device for the pieces of a transformer block that do not sit on its dependency
# chain.  At bs = 4 every kernel of the block is latency-bound (a dependent launch costs ~4 us of boundary on top of a
# 10-50 us kernel that leaves most CUs idle), so work that only depends on the block's INPUTS -- the cross-attention K/V
# projection of the 77 context tokens, the capture of the distillation side outputs, the context gradient -- runs beside the
# chain instead of in it.  fork(): the lane waits for everything issued on the main stream so far; join(ev): the main stream
# waits for the lane's event.  Tensors allocated inside ``with lane:`` belong to the lane's pool: they are only reused by later
# lane work, which always starts with a fork issued after the free.  ADAP_SIDE_LANE=0 switches it off.
# ---------------------------------------------------------------------------------------------
# ADAP_PRESCALE_Q=1: the self-attention query projection's weight pack carries d^-1/2 * log2(e) (folded in before the pack's one
# bf16 rounding), so the attention kernels get their scores in the exp2 domain straight from the matrix core
# (adap_attention_fwd / _bwd with scale = 0): one VALU instruction per score less.  Parity-green (tests/test_kernels_gpu.py,
# the model tests: the same error figures as the unscaled form).  Round 3 measured nothing for it on one stream (124.27 vs
# 124.40 img/s); round 5, on two lanes with the attention workgroups' XCD map: 24.09 vs 24.27 ms per micro-batch in both of two
# interleaved pairs (profiles/r05_ab_misc.log) -- ON by default since; ADAP_PRESCALE_Q=0 switches it off.
PRESCALE_Q = os.environ.get("ADAP_PRESCALE_Q", "1") == "1"
LOG2E = 1.4426950408889634


def q_prescale(d_head):
    return float(d_head) ** -0.5 * LOG2E


SIDE_LANE = os.environ.get("ADAP_SIDE_LANE", "1") != "0"
# the token maps' gradient inside the cross-attention backward's epilogues (0: the separate read-modify-write kernels; A/B aid)
TOKMAP_FOLD = os.environ.get("ADAP_TOKMAP_FOLD", "1") != "0"
# FeedForward's GEGLU inside its two contractions (ops.linear_geglu_fwd / _bwd): the frozen-UNet path; 0 = separate kernels
GEGLU_FUSED = os.environ.get("ADAP_GEGLU_FUSED", "1") != "0"
_LANES = {}


class _Lane:
    def __init__(self, device):
        self.side = torch.cuda.Stream(device=device)
        self.pending = None             # event of lane work whose join was deferred (the captures of a UNet forward)

    def fork(self):
        self.side.wait_stream(torch.cuda.current_stream())

    def __enter__(self):
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        self._ctx.__exit__(*a)

    def mark(self):
        """an event at the lane's current position (call inside ``with lane``)"""
        ev = torch.cuda.Event()
        ev.record(self.side)
        return ev

    mark_now = mark                     # (the event is recorded on the lane's stream whichever stream is current)


_LANE_STREAMS = set()


def side_lane(t):
    """the side lane of the CURRENT stream on t's device, or None (CPU tensors, lane switched off, or inside a lane already).
    One lane per issuing stream: the prefetchers run whole UNet passes on streams of their own (the distillation teacher's
    rollout), and a lane shared with the main stream would chain the two pipelines to each other through its fork / join
    events (measured: config 2's mix 75 -> 63 img/s)."""
    if not SIDE_LANE or not t.is_cuda:
        return None
    cur = ops._stream()                                  # (raw handle: torch.cuda.current_stream() costs ~9 us)
    if cur in _LANE_STREAMS:
        return None
    key = (t.device.index, cur)
    lane = _LANES.get(key)
    if lane is None:
        if len(_LANES) >= 8:                             # streams come and go (tests): do not collect lanes for ever
            return None
        lane = _LANES[key] = _Lane(t.device)
        _LANE_STREAMS.add(lane.side.cuda_stream)
    return lane


def join_side_lane(device=None):
    """the current stream waits for its lane's work whose join was deferred (UNetModel.forward calls this before it hands out
    the captured activations)."""
    if not _LANES:
        return
    idx = torch.cuda.current_device() if device is None else device.index
    lane = _LANES.get((idx, ops._stream()))
    if lane is not None and lane.pending is not None:
        torch.cuda.current_stream(lane.side.device).wait_event(lane.pending)
        lane.pending = None


class KeyCompaction:
    """a self-attention key mask [B,N] uint8 with its compaction: ``perm`` / ``inv_perm`` [B,N] int32, ``count`` [B] int32
    (attention.KeyMasks.compaction)."""

    def __init__(self, mask, perm, inv_perm, count):
        self.mask, self.perm, self.inv_perm, self.count = mask, perm, inv_perm, count


# self-attention with a key mask at N >= this many tokens gathers the kept keys to the front and runs on them alone (two row
# gathers per layer against ~1/4 of the key tiles at the training masks' border widths); ADAP_COMPACT_KEYS=0 switches it off
COMPACT_KEYS_MIN_N = int(os.environ.get("ADAP_COMPACT_KEYS_MIN_N", "1024")) if os.environ.get("ADAP_COMPACT_KEYS", "1") != "0" else 1 << 30


# Set by UNetModel.forward for the duration of a pass: a stamp of ALL the model's parameters (sum of their version counters --
# every in-place change increases it -- and the xor of their addresses), taken once per pass.  While it is set, a block may
# reuse the dict of packs / parameter tuples it built for the same stamp instead of re-stamping each of its ~15 weights on
# every call (256 WeightCache.get + ~3700 nn.Module attribute lookups per training step: ~1.5 ms of host time).
MODEL_STAMP = None


def model_stamp(params):
    v, a, g = 0, 0, 0
    for p_ in params:
        v += p_._version
        a ^= p_.data_ptr()
        g += p_.requires_grad           # (freezing / unfreezing changes what a block's "train" entry holds)
    return v, a, g


class WeightCache:
    """bf16 packs of a module's parameters, rebuilt when a parameter changes (``_version``)."""

    def __init__(self):
        self._packs = {}

    def get(self, key, weights, bias=None, cat_dim0=False, row_scales=None, row_perm=None):
        """``row_scales``: one factor per weight in ``weights``, applied in f32 before the pack's single bf16 rounding (the
        self-attention query projection carries d^-1/2 * log2(e): PRESCALE_Q)."""
        ws = weights if isinstance(weights, (list, tuple)) else [weights]
        bs = bias if isinstance(bias, (list, tuple)) else [bias]
        stamp = tuple((w.data_ptr(), w._version) for w in ws) + tuple((b.data_ptr(), b._version) for b in bs if b is not None) \
            + (tuple(row_scales) if row_scales is not None else ()) + ((row_perm,) if row_perm is not None else ())
        hit = self._packs.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            if row_scales is not None:
                ws = [x.detach().float() * float(f) if f != 1.0 else x for x, f in zip(ws, row_scales)]
            w = torch.cat([x.detach() for x in ws], dim=0) if len(ws) > 1 else ws[0].detach()
            b = None
            if bs[0] is not None:
                b = torch.cat([x.detach() for x in bs], dim=0) if len(bs) > 1 else bs[0].detach()
            if row_perm == "geglu":          # the fused-GEGLU row order of ff.net.0.proj (ops.geglu_row_permutation)
                perm = ops.geglu_row_permutation(w.shape[0], w.device)
                w = w[perm]
                b = None if b is None else b[perm]
            pk = ops.PackedConv(w, b)
        self._packs[key] = (stamp, pk)
        return pk

    def clear(self):
        self._packs.clear()


# ---------------------------------------------------------------------------------------------
# bf16 side copies of residual-stream gradients.  Autograd hands ONE tensor (f32, the reference's precision for the
# residual stream) from a block's backward to the previous block's; the first thing that block does with it is a
# data-gradient contraction, whose matrix-core operand is bf16 anyway.  The kernel that produces the f32 gradient
# (GroupNorm backward with accumulation, conv epilogue) writes the bf16 operand copy in the same pass, and the copy
# travels beside autograd in this table, keyed by the f32 tensor's address.  The entry keeps the f32 tensor alive (so
# its address cannot be reused, and autograd's input buffer cannot add into it in place: use_count > 1) and records
# its version; a gradient that was summed with another one by autograd is a new tensor and simply misses here.
# ---------------------------------------------------------------------------------------------
_GRAD16 = {}
STATS = [0, 0]          # hits, misses of the table (diagnostics)


def _stash16(g32, g16):
    if g16 is not None:
        _GRAD16[g32.data_ptr()] = (g32, g32._version, g16)
    return g32


def _operand(g):
    """the bf16 copy of gradient ``g`` if its producer left one (same storage, unmodified), else ``g`` itself."""
    e = _GRAD16.pop(g.data_ptr(), None)
    if e is not None and e[0].numel() == g.numel() and e[0]._version == e[1] == g._version:
        STATS[0] += 1
        return e[2].view(g.shape)
    STATS[1] += 1
    return g


def clear_grad_copies():
    _GRAD16.clear()


# ---------------------------------------------------------------------------------------------
# weight gradients (`unfreeze_model: True`): a block's parameter tensors travel in P["train"] = {layer: (weight, bias)};
# the block's backward adds their gradients straight into ``param.grad`` (a view of the optimiser's flat gradient
# buffer when Prodigy.grad_buffer() set one up) -- autograd never sees the parameters, only the activation tape.
# ---------------------------------------------------------------------------------------------
def train_of(**layers):
    """{name: (weight, bias)} of the given modules if any of their parameters trains (`unfreeze_model: True`,
    ddpm.py:775-786), else None: the block Functions then also produce weight gradients (functional._dw_*)."""
    out, any_grad = {}, False
    for name, m in layers.items():
        if m is None:
            continue
        w, b = m.weight, getattr(m, "bias", None)
        out[name] = (w, b)
        any_grad = any_grad or w.requires_grad or (b is not None and b.requires_grad)
    return out if any_grad else None


# GradReducer's bucketed exchange (parallel.GradReducer.begin_backward): called with every parameter whose gradient a block
# Function has finished writing through raw pointers -- autograd's post-accumulate hooks never fire for those
GRAD_DONE = None


# How often a block's backward will run before its parameters' gradients are final: a block Function invoked several times in
# one graph (shared_step(batched_student=False, num_denoising_steps > 1): ND UNet passes under one autograd.backward) adds into
# the same flat-buffer range once per invocation.  Every block forward that records a backward node counts here, per trainable
# parameter; GradReducer.begin_backward() takes the counts and lets a chunk go out only after the LAST expected completion.
FWD_PASSES = {}


def _note_forward(ctx, T):
    if T is None or not any(ctx.needs_input_grad):        # frozen block, or no backward node (no-grad pass)
        return
    for w, b in T.values():
        for p_ in (w, b):
            if p_ is not None and p_.requires_grad:
                FWD_PASSES[id(p_)] = FWD_PASSES.get(id(p_), 0) + 1


def take_forward_passes():
    """{id(param): block invocations recorded since the last call}; resets the counters."""
    out = dict(FWD_PASSES)
    FWD_PASSES.clear()
    return out


def _grads_done(T):
    if GRAD_DONE is not None and T is not None:
        for w, b in T.values():
            if w is not None and w.requires_grad:
                GRAD_DONE(w)
            if b is not None and b.requires_grad:
                GRAD_DONE(b)


def _acc(param):
    if param is None or not param.requires_grad:
        return None
    if param.grad is None:
        param.grad = torch.zeros_like(param, memory_format=torch.contiguous_format)
    return param.grad


def _dw_conv(T, key, x, dy, K=1, stride=1, pad=0, up=0, with_bias=True):
    w, b = T[key]
    dw, db = _acc(w), (_acc(b) if with_bias else None)
    if dw is not None or db is not None:
        ops.conv2d_bwd_weight(x, dy, dw, db, K, stride, pad, up, accumulate=True)


def _dw_lin(T, key, x, dy):
    w, b = T[key]
    dw, db = _acc(w), _acc(b)
    if dw is not None or db is not None:
        ops.linear_bwd_weight(x, dy, dw, db, accumulate=True)


def _dw_norm(T, key, dy, x, gamma, beta, mean, rstd, kind, act):
    w, b = T[key]
    dg, db = _acc(w), _acc(b)
    if dg is not None or db is not None:
        ops.norm_affine_bwd(dy, x, gamma, beta, mean, rstd, kind, act, dg, db, accumulate=True)


def _conv_bwd_data(g, pk, K, pad, bf16=False):
    """dX of a stride-1 conv: the same implicit GEMM with the flipped/transposed pack.  ``bf16``: the result only
    feeds a GroupNorm backward, so it is written as bf16 (half the traffic of both kernels)."""
    gx32, gx16 = ops.conv2d(g, pk.bwd, pk.bwd.shape[1], K, 1, K - 1 - pad, out_f32=not bf16, out_bf16=bf16)
    return gx16 if bf16 else gx32


def _lin_bwd(g, pk, out_f32=True, out_bf16=False):
    """dX of a Linear / 1x1 conv: g [..., O] -> [..., I]."""
    return ops.linear(g, pk.bwd[:, :, :], pk.bwd.shape[1], out_f32=out_f32, out_bf16=out_bf16)


# ---------------------------------------------------------------------------------------------
# ResBlock (openaimodel.py:259-279)
# ---------------------------------------------------------------------------------------------
RESBLOCK_C = os.environ.get("ADAP_RESBLOCK_C", "1") != "0"      # the frozen ResBlock issued from one C call (csrc/blocks.hip)
_RB_WS = {}


def _resblock_ws(B, H, W, Cin, Cout, has_skip):
    """(GroupNorm workspace floats, split-K workspace floats) of a ResBlock's forward and backward calls: the largest each."""
    key = (B, H, W, Cin, Cout, has_skip)
    v = _RB_WS.get(key)
    if v is None:
        q = ops._lib.size_query
        gn = max(q("adap_groupnorm_workspace_floats", B, H * W, Cin), q("adap_groupnorm_workspace_floats", B, H * W, Cout))
        convs = [(Cin, Cout, 3), (Cout, Cout, 3), (Cout, Cin, 3)] + ([(Cin, Cout, 1), (Cout, Cin, 1)] if has_skip else [])
        sk = max(q("adap_conv2d_workspace_floats", B, H, W, ci, co, k, k) for ci, co, k in convs)
        v = _RB_WS[key] = (gn, sk)
    return v


class ResBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, emb_out, P):
        """x [B,H,W,Cin] f32; emb_out [B,Cout] f32 (= emb_layers(emb), added per (batch, channel));
        P: dict with gn1/gn2 (gamma, beta), conv1/conv2/skip PackedConv."""
        _note_forward(ctx, P.get("train"))
        g1w, g1b = P["gn1"]
        g2w, g2b = P["gn2"]
        c1, c2, sk = P["conv1"], P["conv2"], P["skip"]
        if (RESBLOCK_C and not F32_STORAGE and P.get("train") is None and ops.TIMER is None and x.dtype == torch.float32
                and x.is_contiguous() and emb_out.is_contiguous() and c1.O4 == c1.O and c1.I8 == c1.I):
            # one C call issues the block's five launches (host time: ~100 us of wrappers and allocations -> ~30)
            B, H, W, Cin = x.shape
            Cout = c1.O
            dev = x.device
            gn_n, sk_n = _resblock_ws(B, H, W, Cin, Cout, sk is not None)
            a1 = torch.empty(B, H, W, Cin, device=dev, dtype=BF16)
            hh = torch.empty(2, B, H, W, Cout, device=dev, dtype=BF16)          # h1 (kept for the backward) | a2
            out = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float32)
            skip = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float32) if sk is not None else None
            stats = torch.empty(4, B, 32, device=dev, dtype=torch.float32)
            ws = torch.empty(gn_n + sk_n, device=dev, dtype=torch.float32)
            ops._lib.call("adap_resblock_fwd", x.data_ptr(), emb_out.data_ptr(), g1w.data_ptr(), g1b.data_ptr(), g2w.data_ptr(),
                          g2b.data_ptr(), c1.fwd.data_ptr(), ops._ptr(c1.bias), c2.fwd.data_ptr(), ops._ptr(c2.bias),
                          0 if sk is None else sk.fwd.data_ptr(), 0 if sk is None else ops._ptr(sk.bias), a1.data_ptr(),
                          hh[0].data_ptr(), hh[1].data_ptr(), ops._ptr(skip), out.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                          (ws.data_ptr() + 4 * gn_n) if sk_n else 0, ops.gn_sync_buffer(dev), B, H, W, Cin, Cout, ops._stream())
            ctx.P = P
            ctx.fast = True
            ctx.save_for_backward(x, hh[0], stats)
            return out
        ctx.fast = False
        _, a1, m1, r1 = ops.groupnorm_fwd(x, g1w, g1b, 1e-5, 1)
        # h1 is block-internal (read only by the second norm and its backward): kept in bf16
        h1_32, h1 = ops.conv2d(a1, c1.fwd, c1.O4, 3, 1, 1, bias=c1.bias, chan_add=emb_out, out_f32=F32_STORAGE,
                               out_bf16=not F32_STORAGE)
        if F32_STORAGE:
            h1 = h1_32
        _, a2, m2, r2 = ops.groupnorm_fwd(h1, g2w, g2b, 1e-5, 1)
        if sk is None:
            skip = x
        else:
            skip, _ = ops.conv2d(x, sk.fwd, sk.O4, 1, bias=sk.bias)
        out, _ = ops.conv2d(a2, c2.fwd, c2.O4, 3, 1, 1, bias=c2.bias, residual=skip)
        ctx.P = P
        train = P.get("train") is not None          # weight gradients also need the two normalised activations
        ctx.save_for_backward(x, h1, m1, r1, m2, r2, a1 if train else None, a2 if train else None)
        return out

    @staticmethod
    def backward(ctx, g):
        P = ctx.P
        if ctx.fast and not ctx.needs_input_grad[1] and ops.TIMER is None:
            x, h1, stats = ctx.saved_tensors
            g1w, g1b = P["gn1"]
            g2w, g2b = P["gn2"]
            c1, c2, sk = P["conv1"], P["conv2"], P["skip"]
            B, H, W, Cin = x.shape
            Cout = c1.O
            dev = x.device
            g = g.contiguous()
            gop = _operand(g)
            gn_n, sk_n = _resblock_ws(B, H, W, Cin, Cout, sk is not None)
            sc = torch.empty(2, B, H, W, Cout, device=dev, dtype=BF16)            # ga2 | gh1
            ga1 = torch.empty(2, B, H, W, Cin, device=dev, dtype=BF16)           # ga1 | gx16
            gx = torch.empty(B, H, W, Cin, device=dev, dtype=torch.float32)
            ws = torch.empty(gn_n + sk_n, device=dev, dtype=torch.float32)
            ops._lib.call("adap_resblock_bwd", gop.data_ptr(), 0 if gop.dtype == torch.float32 else 1, g.data_ptr(), x.data_ptr(),
                          h1.data_ptr(), stats.data_ptr(), g1w.data_ptr(), g1b.data_ptr(), g2w.data_ptr(), g2b.data_ptr(),
                          c1.bwd.data_ptr(), c2.bwd.data_ptr(), 0 if sk is None else sk.bwd.data_ptr(), sc[0].data_ptr(),
                          sc[1].data_ptr(), ga1[0].data_ptr(), gx.data_ptr(), ga1[1].data_ptr(), ws.data_ptr(),
                          (ws.data_ptr() + 4 * gn_n) if sk_n else 0, ops.gn_sync_buffer(dev), B, H, W, Cin, Cout, ops._stream())
            return _stash16(gx, ga1[1]), None, None
        if ctx.fast:          # (saved in the compact form; the general path below wants the per-tensor one)
            x, h1, stats = ctx.saved_tensors
            m1, r1, m2, r2 = stats[0], stats[1], stats[2], stats[3]
            a1 = a2 = None
        else:
            x, h1, m1, r1, m2, r2, a1, a2 = ctx.saved_tensors
        T = P.get("train")
        g1w, g1b = P["gn1"]
        g2w, g2b = P["gn2"]
        c1, c2, sk = P["conv1"], P["conv2"], P["skip"]
        gop = _operand(g)
        ga2 = _conv_bwd_data(gop, c2, 3, 1, bf16=not F32_STORAGE)
        _, gh1 = ops.groupnorm_bwd(ga2, h1, g2w, g2b, m2, r2, 1, out_f32=False, out_bf16=True)
        ga1 = _conv_bwd_data(gh1, c1, 3, 1, bf16=not F32_STORAGE)
        g_emb = None
        db1 = _acc(T["conv1"][1]) if T is not None else None
        if ctx.needs_input_grad[1] or db1 is not None:
            # d emb_out[b][c] = sum over the image's pixels of d h1 (openaimodel.py:264-268); conv1's bias gradient is its
            # sum over the batch
            g_emb = torch.empty(x.shape[0], gh1.shape[-1], device=x.device, dtype=F32)
            ops.colsum(gh1, g_emb, seg_rows=gh1.shape[1] * gh1.shape[2], accumulate=False)
            if db1 is not None:
                db1.add_(g_emb.sum(0))
        if T is not None:
            _dw_conv(T, "conv2", a2, gop, 3, 1, 1, with_bias=False)
            _dw_norm(T, "gn2", _op16(ga2), _op16(h1), g2w, g2b, m2, r2, 0, 1)
            _dw_conv(T, "conv1", a1, gh1, 3, 1, 1, with_bias=False)
            _dw_norm(T, "gn1", _op16(ga1), x, g1w, g1b, m1, r1, 0, 1)
            if sk is not None:
                _dw_conv(T, "skip", x, gop, 1, with_bias=False)
            # conv2 and the 1x1 skip see the same output gradient: one column sum serves both bias gradients
            dbs = [d for d in (_acc(T["conv2"][1]), _acc(T["skip"][1]) if sk is not None else None) if d is not None]
            if len(dbs) == 1:
                ops.colsum(gop, dbs[0], accumulate=True)
            elif len(dbs) == 2:
                tmp = torch.empty(1, gop.shape[-1], device=x.device, dtype=F32)
                ops.colsum(gop, tmp, accumulate=False)
                dbs[0].add_(tmp[0])
                dbs[1].add_(tmp[0])
            if not ctx.needs_input_grad[1]:
                g_emb = None
        _grads_done(T)
        if sk is None:          # identity skip: dx + g straight into a new tensor (no clone of g)
            gx, gx16 = ops.groupnorm_bwd(ga1, x, g1w, g1b, m1, r1, 1, out_bf16=True, add_from=g)
        else:
            gx, _ = ops.conv2d(gop, sk.bwd, sk.bwd.shape[1], 1)
            _, gx16 = ops.groupnorm_bwd(ga1, x, g1w, g1b, m1, r1, 1, out_bf16=True, accumulate_into=gx)
        return _stash16(gx, gx16), g_emb, None


# ---------------------------------------------------------------------------------------------
# SpatialTransformer with one BasicTransformerBlock (attention.py:260-341)
# ---------------------------------------------------------------------------------------------
# the frozen SpatialTransformer block issued from one C call each way (csrc/blocks.hip adap_stblock_fwd / _bwd)
STBLOCK_C = os.environ.get("ADAP_STBLOCK_C", "1") != "0"
STB_CALLS = [0, 0, 0]                  # forward / backward calls that took the C path, backward calls that skipped the input gradient
_STB_WS = {}
# include/adaprompt_hip.h: ADAP_STB_* flags
_STB_SAME_CTX, _STB_COMPACT, _STB_CAPTURE, _STB_Q1_PRESCALED, _STB_TOKGRAD, _STB_WANT_GK, _STB_WANT_GV, _STB_G_BF16, _STB_NO_GX, \
    _STB_KV_GIVEN = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
_STB_CAPTURE_DEFERRED, _STB_TOKPREP_GIVEN = 1024, 2048
STB_PRUNE_GX = os.environ.get("ADAP_STB_PRUNE_GX", "1") != "0"     # A/B switch: skip the input-gradient half where it is not needed


def _stb_sizes(B, N, C, Cctx, M, heads, G):
    key = (B, N, C, Cctx, M, heads, G)
    v = _STB_WS.get(key)
    if v is None:
        q = ops._lib.size_query
        d = C // heads
        v = (q("adap_groupnorm_workspace_floats", B, N, C), q("adap_stblock_workspace_floats", B, N, C, Cctx, M),
             max(q("adap_attention_bwd_workspace_floats", B, heads, N, N, d), q("adap_attention_bwd_workspace_floats", B, heads, N, M, d)),
             q("adap_attention_tokmap_prep_workspace_floats", B, heads, N, d, G) if G else 0)
        v = _STB_WS[key] = tuple((n + 63) // 64 * 64 for n in v)          # the carved pieces stay 256-byte aligned
    return v


def _stb_weights(P, same_ctx):
    """the block's device pointers in ADAP_STW_* / ADAP_STWB_* order (ctypes arrays), cached on the pack dict (which lives as
    long as the weights it was built from are unchanged: functional.MODEL_STAMP)."""
    hit = P.get("_stb")
    if hit is not None:
        return hit
    import ctypes
    gn, n1, n2, n3 = P["norm"], P["norm1"], P["norm2"], P["norm3"]
    pin, qkv, o1, q2, o2, ff1g, ff2, pout = (P["proj_in"], P["qkv1"], P["to_out1"], P["q2"], P["to_out2"], P["ff1g"](), P["ff2"],
                                             P["proj_out"])
    kv, v2 = (P["kv2"], None) if same_ctx else (P["k2"], P["v2"])
    dp = lambda t: 0 if t is None else t.data_ptr()          # noqa: E731
    fw = [dp(gn[0]), dp(gn[1]), dp(pin.fwd), dp(pin.bias), dp(n1[0]), dp(n1[1]), dp(qkv.fwd), dp(o1.fwd), dp(o1.bias), dp(n2[0]),
          dp(n2[1]), dp(q2.fwd), dp(kv.fwd), 0 if v2 is None else dp(v2.fwd), dp(o2.fwd), dp(o2.bias), dp(n3[0]), dp(n3[1]),
          dp(ff1g.fwd), dp(ff1g.bias), dp(ff2.fwd), dp(ff2.bias), dp(pout.fwd), dp(pout.bias)]
    bw = [dp(gn[0]), dp(gn[1]), dp(pin.bwd), dp(n1[0]), dp(qkv.bwd), dp(o1.bwd), dp(n2[0]), dp(q2.bwd), dp(kv.bwd),
          0 if v2 is None else dp(v2.bwd), dp(o2.bwd), dp(n3[0]), dp(ff1g.bwd), dp(ff2.bwd), dp(pout.bwd)]
    ok = (pin.bias is not None and o1.bias is not None and o2.bias is not None and ff1g.bias is not None and ff2.bias is not None
          and pout.bias is not None and all(pk.O4 == pk.O and pk.I8 == pk.I for pk in (pin, qkv, o1, q2, o2, ff1g, ff2, pout, kv)))
    hit = P["_stb"] = ((ctypes.c_void_p * len(fw))(*fw), (ctypes.c_void_p * len(bw))(*bw), ok, kv.I)
    return hit


# ---------------------------------------------------------------------------------------------
# The distillation layers' token maps off the blocks' chains (csrc/attention.hip TokBatch): a block's capture launch, and the
# three launches of its gradient prologue (kw = w^T K, gq = dT^T Q), are dependent links of the block's chain -- ~20 us of a lane's
# time each -- although nothing in the block reads the maps, and the prologue only needs what the forward saved plus the losses'
# gradients.  With BATCH_TOKMAPS the blocks of one UNet forward DEFER their captures (``DEFERRED_CAPTURES``, flushed by
# UNetModel.forward as ONE launch over all layers) and ``prepare_tokmap_backward`` makes all layers' prologues in THREE launches
# in front of ``autograd.backward``: 12 + 36 launches per micro-batch become 1 + 3.
# ---------------------------------------------------------------------------------------------
BATCH_TOKMAPS = os.environ.get("ADAP_BATCH_TOKMAPS", "1") != "0"
DEFERRED_CAPTURES = None          # a list while a UNetModel.forward collects its blocks' token-map captures


def _tok_batched(entry, items, fwd):
    """items: [(q [B,N,C] bf16, kv [B,M,2C] bf16, tok_w, tokmap | None, d_tokmap | None, workspace | None, heads, scale)]"""
    import ctypes
    for i0 in range(0, len(items), 16):
        part = items[i0:i0 + 16]
        n = len(part)
        ptrs, lds, dims, scales = (ctypes.c_void_p * (6 * n))(), (ctypes.c_long * (2 * n))(), (ctypes.c_int * (6 * n))(), \
            (ctypes.c_float * n)()
        for i, (q, kv, tok_w, tokmap, dt, ws, heads, scale) in enumerate(part):
            B, N, C = q.shape
            M = kv.shape[1]
            for j, t_ in enumerate((q, kv, tok_w, tokmap, dt, ws)):
                ptrs[6 * i + j] = None if t_ is None else t_.data_ptr()
            lds[2 * i], lds[2 * i + 1] = ops._rows_ld(q)[1], ops._rows_ld(kv)[1]
            for j, v in enumerate((B, heads, N, M, C // heads, tok_w.shape[2])):
                dims[6 * i + j] = v
            scales[i] = scale
        if fwd:
            ops._lib.call(entry, n, ptrs, lds, dims, scales, ops._stream())
        else:
            ops._lib.call(entry, n, ptrs, lds, dims, ops._stream())


def flush_deferred_captures():
    """the token maps the blocks of the current UNet forward deferred, as one launch (UNetModel.forward, after its last block)."""
    global DEFERRED_CAPTURES
    items, DEFERRED_CAPTURES = DEFERRED_CAPTURES, None
    if items:
        _tok_batched("adap_attention_tokmap_fwd_batched", items, True)


def prepare_tokmap_backward(roots, grads):
    """call in front of ``torch.autograd.backward(roots, grads)``: for every root that is a transformer block's token-map output
    (the fused regularisers' gradients enter there), the block's gradient prologue is made NOW, all layers in three launches; the
    block's backward finds it on its ctx (``tok_prep``) and skips its own three."""
    if not BATCH_TOKMAPS:
        return
    items, owners = [], []
    for r, g in zip(roots, grads):
        fn = getattr(r, "grad_fn", None)
        if fn is None or type(fn).__name__ != "SpatialTransformerFnBackward" or not getattr(fn, "fast", False):
            continue
        tok_w = getattr(fn, "tok_w", None)
        if tok_w is None or getattr(r, "output_nr", -1) != 4 or g is None or g.dtype != torch.float32 or not g.is_contiguous():
            continue
        if getattr(fn, "tok_prep", None) is not None:
            continue
        x, _gn, _tres, _ln, _qkv1, obuf, _lse, _hh, _kv1c, kv2, ctx_k, _cv = fn.saved_tensors
        B, H, W, C = x.shape
        heads, G = fn.heads, tok_w.shape[2]
        if tuple(g.shape) != (B, heads, H * W, G):
            continue
        ws = torch.empty(_stb_sizes(B, H * W, C, ctx_k.shape[-1], kv2.shape[1], heads, G)[3], device=x.device, dtype=torch.float32)
        items.append((obuf[1], kv2, tok_w, None, g, ws, heads, 1.0))
        owners.append((fn, ws, g))
    if items:
        _tok_batched("adap_attention_tokmap_prep_batched", items, False)
        for fn, ws, g in owners:
            fn.tok_prep = (ws, g.data_ptr(), g._version)


HOIST_KV = os.environ.get("ADAP_HOIST_KV", "1") != "0"      # A/B switch: the 16 layers' context K | V projections as grouped launches


class HoistedKV:
    """a layer's cross-attention context together with its K | V projection, made in front of the UNet by ``ContextKVFn``:
    ``kv`` bf16 [B, M, 2C] (the block's ``kv2``), ``dkv`` the slot of the same shape its backward writes the gradient into."""
    __slots__ = ("ctx", "kv", "dkv")

    def __init__(self, ctx, kv, dkv):
        self.ctx, self.kv, self.dkv = ctx, kv, dkv


class ContextKVFn(torch.autograd.Function):
    """The cross-attention K | V projections of ALL conditioned layers (attention.py:195-213 ``to_k`` / ``to_v`` of ``attn2``, 16
    of them in SD-1.5, each on its own layer's 77 context tokens) hoisted in front of the UNet: a layer's projection is a
    308-row GEMM (B = 4) that costs a whole dependent launch (25-29 us) at the head of its block's chain; layers that are
    neighbours in the layerwise context and have the same width (SD-1.5: layers 0-1, 2-3, 4-9, 10-12, 13-15) go out as ONE
    batched launch (``adap_conv2d_nhwc`` nbatch: per-layer weights, per-layer rows), 5 launches instead of 16, and the context
    gradients -- dK Wk + dV Wv per layer -- as 5 behind the UNet's backward instead of 16 inside it.  Frozen weights only.

    forward(ctx_l f32 [L, B, M, Cctx] layer-major, runs) -> L tensors bf16 [B, M, 2C_l];  ``runs``: [(first layer, layers, C,
    stacked forward packs [n][2C][Cctx], stacked data-gradient packs [n][Cctx][2C])].  The blocks write their dK | dV straight
    into the run's gradient buffer (``slots``, handed out with the outputs through ``LAST_SLOTS``), so the backward here is
    the batched contraction alone."""
    LAST_SLOTS = None

    @staticmethod
    def forward(ctx, ctx_l, runs):
        L, B, M, Cctx = ctx_l.shape
        assert ctx_l.dtype == torch.float32 and ctx_l.is_contiguous()
        c16 = ops.pad_cast_bf16(ctx_l)
        rows = B * M
        outs, slots, dbufs = [], [], []
        for l0, n, C, wf, _wb in runs:
            kv = torch.empty(n, B, M, 2 * C, device=ctx_l.device, dtype=BF16)
            ops._lib.call("adap_conv2d_nhwc", c16[l0].data_ptr(), 1, Cctx, wf.data_ptr(), 0, 0, 0, 0, 0, 0, 0, kv.data_ptr(), 2 * C,
                          1, rows, 1, Cctx, rows, 1, 2 * C, 1, 1, 1, 0, 0, 1.0, 1, 0, n, rows * Cctx, 2 * C * Cctx, 0, rows * 2 * C,
                          ops._stream())
            outs += list(kv.unbind(0))
            if ctx.needs_input_grad[0]:            # (a no-grad pass -- the teacher's, a sampler's -- has no gradient slots)
                dkv = torch.empty(n, B, M, 2 * C, device=ctx_l.device, dtype=BF16)
                slots += list(dkv.unbind(0))
                dbufs.append(dkv)
            else:
                slots += [None] * n
        ctx.runs, ctx.dbufs, ctx.slots, ctx.shape = runs, dbufs, slots, (L, B, M, Cctx)
        ctx.set_materialize_grads(False)
        ContextKVFn.LAST_SLOTS = slots
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        L, B, M, Cctx = ctx.shape
        rows = B * M
        for g, slot in zip(gs, ctx.slots):
            if g is None:
                slot.zero_()                       # (a layer whose K | V nothing downstream used)
            elif g.data_ptr() != slot.data_ptr():
                slot.copy_(g)                      # (a gradient that did not come from the block's own write into the slot)
        g_ctx = torch.empty(L, B, M, Cctx, device=ctx.dbufs[0].device, dtype=torch.float32)
        for (l0, n, C, _wf, wb), dkv in zip(ctx.runs, ctx.dbufs):
            ops._lib.call("adap_conv2d_nhwc", dkv.data_ptr(), 1, 2 * C, wb.data_ptr(), 0, 0, 0, 0, 0, g_ctx[l0].data_ptr(), Cctx, 0, 0,
                          1, rows, 1, 2 * C, rows, 1, Cctx, 1, 1, 1, 0, 0, 1.0, 1, 0, n, rows * 2 * C, Cctx * 2 * C, rows * Cctx, 0,
                          ops._stream())
        return g_ctx, None


class SpatialTransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ctx_k, ctx_v, P, heads, key_mask, capture, tok_w=None, tokmap_only=False, kv=None, dkv_slot=None):
        """``kv`` / ``dkv_slot``: the cross-attention K | V projection of this layer's context already made (``ContextKVFn``:
        bf16 [B, M, 2C]) and where the backward is to leave dK | dV; ``ctx_k`` / ``ctx_v`` are then only shapes."""
        _note_forward(ctx, P.get("train"))
        B, H, W, C = x.shape
        N = H * W
        same_ctx = ctx_k is ctx_v or (ctx_k.data_ptr() == ctx_v.data_ptr() and ctx_k.shape == ctx_v.shape)
        ctx.fast = False
        if (STBLOCK_C and GEGLU_FUSED and TOKMAP_FOLD and not F32_STORAGE and P.get("train") is None and ops.TIMER is None
                and x.dtype == torch.float32 and x.is_contiguous() and ctx_k.dtype == torch.float32 and ctx_k.is_contiguous()
                and ctx_v.dtype == torch.float32 and ctx_v.is_contiguous() and ctx_k.shape == ctx_v.shape
                and (not capture or (tokmap_only and tok_w is not None))
                and (key_mask is None or isinstance(key_mask, KeyCompaction) or torch.is_tensor(key_mask))):
            fw, _bw, ok, Cctx = _stb_weights(P, same_ctx)
            if ok and ctx_k.shape[-1] == Cctx:
                return SpatialTransformerFn._forward_c(ctx, x, ctx_k, ctx_v, P, heads, key_mask, capture, tok_w, same_ctx, fw, Cctx,
                                                       kv, dkv_slot)
        # the cross-attention K/V projection reads only the 77 context tokens: on the side lane, under the block's first half
        lane = side_lane(x)
        M = ctx_k.shape[1]

        def project_kv():
            if same_ctx:
                return ops.linear(ctx_k, P["kv2"].fwd, 2 * C, out_f32=False, out_bf16=True)[1]
            kv = torch.empty(B, M, 2 * C, device=x.device, dtype=BF16)
            _, kk = ops.linear(ctx_k, P["k2"].fwd, C, out_f32=False, out_bf16=True)
            _, vv = ops.linear(ctx_v, P["v2"].fwd, C, out_f32=False, out_bf16=True)
            kv[..., :C].copy_(kk)
            kv[..., C:].copy_(vv)
            return kv
        kv2_ready = None
        ctx.dkv_slot = dkv_slot if kv is not None else None
        if kv is not None:
            kv2 = kv
        elif lane is not None:
            lane.fork()
            with lane:
                kv2 = project_kv()
                kv2_ready = lane.mark()
        else:
            kv2 = project_kv()
        gnw, gnb = P["norm"]
        _, xn, gm, gr = ops.groupnorm_fwd(x, gnw, gnb, 1e-6, 0)
        pin = P["proj_in"]
        t0, _ = ops.linear(xn.view(B, N, C), pin.fwd, C, bias=pin.bias)
        # --- attn1 (self) : fused q|k|v projection -------------------------------------------
        n1, l1m, l1r = ops.layernorm_fwd(t0, *P["norm1"])
        qkv = P["qkv1"]
        sc1 = 0.0 if P.get("q1_prescaled") else None       # 0 = "q carries d^-1/2 log2(e)" (include/adaprompt_hip.h)
        _, qkv1 = ops.linear(n1, qkv.fwd, 3 * C, out_f32=False, out_bf16=True)
        q1, k1, v1 = qkv1[..., :C], qkv1[..., C:2 * C], qkv1[..., 2 * C:]
        kc, kv1c = None, None
        if isinstance(key_mask, KeyCompaction):
            kc, key_mask = key_mask, key_mask.mask
            kv1c = ops.gather_rows_bf16(qkv1[..., C:], kc.perm)                   # kept keys first: [B,N,2C]
            o1, lse1 = ops.attention_fwd(q1, kv1c[..., :C], kv1c[..., C:], heads, None, key_count=kc.count, scale=sc1)
        else:
            o1, lse1 = ops.attention_fwd(q1, k1, v1, heads, key_mask, scale=sc1)
        to1 = P["to_out1"]
        t1, _ = ops.linear(o1, to1.fwd, C, bias=to1.bias, residual=t0)
        # --- attn2 (cross) ----------------------------------------------------------------------
        n2, l2m, l2r = ops.layernorm_fwd(t1, *P["norm2"])
        _, q2 = ops.linear(n2, P["q2"].fwd, C, out_f32=False, out_bf16=True)
        if kv2_ready is not None:
            torch.cuda.current_stream().wait_event(kv2_ready)
        k2, v2 = kv2[..., :C], kv2[..., C:]
        o2, lse2 = ops.attention_fwd(q2, k2, v2, heads, None)
        cap = (None, None, None)
        dense = not (tokmap_only and tok_w is not None)
        if capture:
            if lane is not None:
                # the side outputs are read by the losses after the UNet's forward: the capture runs beside the rest of the
                # block, its join is deferred to UNetModel.forward (join_side_lane)
                lane.fork()
                with lane:
                    cap = ops.attention_capture(q2, k2, heads, tok_w=tok_w, dense=dense)
                    lane.pending = lane.mark()
            else:
                cap = ops.attention_capture(q2, k2, heads, tok_w=tok_w, dense=dense)   # (+ token maps when tok_w is given)
            if not dense:
                # shape-only stand-ins for the dense side outputs (the fused regularisers read the token maps; the host
                # expressions, given token maps, read only the shapes): one element each, expanded
                M2 = k2.shape[1]
                stub = x.new_empty(1)
                cap = (stub.expand(B, heads, N, M2), stub.expand(B, heads, N, M2), stub.expand(B, heads, N, C // heads), cap[3])
        to2 = P["to_out2"]
        t2, _ = ops.linear(o2, to2.fwd, C, bias=to2.bias, residual=t1)
        # --- GEGLU feed-forward -----------------------------------------------------------------
        n3, l3m, l3r = ops.layernorm_fwd(t2, *P["norm3"])
        ff1, ff2 = P["ff1"], P["ff2"]
        ff1g = P.get("ff1g") if (GEGLU_FUSED and not F32_STORAGE and P.get("train") is None) else None
        if ff1g is not None:
            # h stays in the permuted channel order of the fused epilogue; the backward uses the same pack's data gradient
            hh, gg = ops.linear_geglu_fwd(n3, ff1g())
        else:
            hh32, hh = ops.linear(n3, ff1.fwd, 8 * C, bias=ff1.bias, out_f32=F32_STORAGE, out_bf16=not F32_STORAGE)
            if F32_STORAGE:
                hh = hh32
                gg = _geglu_fwd_f32(hh)
            else:
                gg = ops.geglu_fwd(hh)
        ctx.geglu_fused = ff1g is not None
        # t3 only feeds proj_out, whose matrix-core operand is bf16 anyway: write it as bf16 only
        _, t3 = ops.linear(gg, ff2.fwd, C, bias=ff2.bias, residual=t2, out_f32=False, out_bf16=True)
        pout = P["proj_out"]
        out, _ = ops.linear(t3, pout.fwd, C, bias=pout.bias, residual=x.view(B, N, C))
        ctx.P, ctx.heads, ctx.same_ctx = P, heads, same_ctx
        ctx.key_mask, ctx.key_compaction = key_mask, kc
        ctx.tok_w = tok_w if capture else None
        tr = P.get("train") is not None      # weight gradients also need each contraction's input operand
        ctx.save_for_backward(x, gm, gr, t0, l1m, l1r, qkv1, o1, lse1, t1, l2m, l2r, q2, kv2, o2, lse2, t2, l3m, l3r, hh,
                              ctx_k, ctx_v, *((xn, n1, n2, n3, gg, t3) if tr else (None,) * 6), kv1c)
        out = out.view(B, H, W, C)
        if capture:
            # attnscore and q*d^-1/4 stay in the graph (the cross-layer consistency loss of the recon iteration reads
            # attnscore with gradient, ddpm.py:3246-3270); the probabilities are a monitoring output only.
            # Unused side outputs must cost nothing in backward: their gradients arrive as None, not as zeros.
            ctx.set_materialize_grads(False)
            if dense:
                ctx.mark_non_differentiable(cap[1])
            else:
                ctx.mark_non_differentiable(cap[0], cap[1], cap[2])
            return (out,) + tuple(cap)
        return out

    @staticmethod
    def _forward_c(ctx, x, ctx_k, ctx_v, P, heads, key_mask, capture, tok_w, same_ctx, fw, Cctx, kv=None, dkv_slot=None):
        import ctypes
        B, H, W, C = x.shape
        N, M = H * W, ctx_k.shape[1]
        rows = B * N
        dev = x.device
        G = tok_w.shape[2] if capture else 0
        gn_n, sk_n, _at_n, _prep_n = _stb_sizes(B, N, C, Cctx, M, heads, G)
        lane = side_lane(x)
        flags = (_STB_SAME_CTX if same_ctx else 0) | (_STB_CAPTURE if capture else 0) | (_STB_Q1_PRESCALED if P.get("q1_prescaled") else 0)
        kc = mask_ptr = None
        if isinstance(key_mask, KeyCompaction):
            kc, key_mask = key_mask, key_mask.mask
            flags |= _STB_COMPACT
        elif key_mask is not None:
            assert key_mask.dtype == torch.uint8 and key_mask.shape == (B, N) and key_mask.is_contiguous()
            mask_ptr = key_mask.data_ptr()
        e = torch.empty
        if kv is not None:
            assert kv.dtype == BF16 and kv.is_contiguous() and tuple(kv.shape) == (B, M, 2 * C)
            kv2 = kv
            flags |= _STB_KV_GIVEN
        else:
            kv2 = e(B, M, 2 * C, device=dev, dtype=BF16)
        ctx.dkv_slot = dkv_slot if kv is not None else None
        gn_stats = e(2, B, 32, device=dev, dtype=torch.float32)
        tres = e(3, B, N, C, device=dev, dtype=torch.float32)
        ln_stats = e(6, rows, device=dev, dtype=torch.float32)
        qkv1 = e(B, N, 3 * C, device=dev, dtype=BF16)
        obuf = e(3, B, N, C, device=dev, dtype=BF16)
        lse = e(2, B, heads, N, device=dev, dtype=torch.float32)
        hh = e(B, N, 8 * C, device=dev, dtype=BF16)
        kv1c = e(B, N, 2 * C, device=dev, dtype=BF16) if kc is not None else None
        tokmap = e(B, heads, N, G, device=dev, dtype=torch.float32) if capture else None
        out = e(B, H, W, C, device=dev, dtype=torch.float32)
        scr = e(7 * rows * C, device=dev, dtype=BF16)
        ws = e(gn_n + 2 * sk_n, device=dev, dtype=torch.float32)
        wp = ws.data_ptr()
        if capture:
            assert tok_w.dtype == torch.float32 and tok_w.is_contiguous() and tuple(tok_w.shape[:2]) == (B, M) and G <= 4
        dp = lambda t: 0 if t is None else t.data_ptr()          # noqa: E731
        tens = (ctypes.c_void_p * 23)(x.data_ptr(), ctx_k.data_ptr(), ctx_v.data_ptr(), mask_ptr or 0,
                                      0 if kc is None else kc.perm.data_ptr(), 0 if kc is None else kc.count.data_ptr(),
                                      dp(tok_w) if capture else 0, kv2.data_ptr(), gn_stats.data_ptr(), tres.data_ptr(),
                                      ln_stats.data_ptr(), qkv1.data_ptr(), obuf.data_ptr(), lse.data_ptr(), hh.data_ptr(), dp(kv1c),
                                      dp(tokmap), out.data_ptr(), scr.data_ptr(), wp, (wp + 4 * gn_n) if sk_n else 0,
                                      (wp + 4 * (gn_n + sk_n)) if sk_n else 0, ops.gn_sync_buffer(dev))
        deferred = capture and DEFERRED_CAPTURES is not None
        if deferred:
            flags |= _STB_CAPTURE_DEFERRED
        cfg = (ctypes.c_int * 11)(B, H, W, C, heads, M, Cctx, flags, G, 0, 0)
        ops._lib.call("adap_stblock_fwd", cfg, fw, tens, 0 if lane is None else lane.side.cuda_stream, ops._stream())
        STB_CALLS[0] += 1
        if deferred:         # (the same arithmetic, one launch for all layers behind the UNet's forward: flush_deferred_captures)
            DEFERRED_CAPTURES.append((obuf[1], kv2, tok_w, tokmap, None, None, heads, float(C // heads) ** -0.5))
        if capture and lane is not None and not deferred:
            lane.pending = lane.mark_now()              # the capture's join is deferred to UNetModel.forward (join_side_lane)
        ctx.fast = True
        ctx.tok_prep = None
        ctx.P, ctx.heads, ctx.same_ctx = P, heads, same_ctx
        ctx.key_mask, ctx.key_compaction = key_mask, kc
        ctx.tok_w = tok_w if capture else None
        ctx.save_for_backward(x, gn_stats, tres, ln_stats, qkv1, obuf, lse, hh, kv1c, kv2, ctx_k, ctx_v)
        if capture:
            ctx.set_materialize_grads(False)
            stub = x.new_empty(1)
            cap = (stub.expand(B, heads, N, M), stub.expand(B, heads, N, M), stub.expand(B, heads, N, C // heads), tokmap)
            ctx.mark_non_differentiable(cap[0], cap[1], cap[2])
            return (out,) + cap
        return out

    @staticmethod
    def _backward_c(ctx, g, g_tokmap):
        import ctypes
        x, gn_stats, tres, ln_stats, qkv1, obuf, lse, hh, kv1c, kv2, ctx_k, ctx_v = ctx.saved_tensors
        P, heads, same_ctx = ctx.P, ctx.heads, ctx.same_ctx
        B, H, W, C = x.shape
        N, M, Cctx = H * W, kv2.shape[1], ctx_k.shape[-1]
        rows = B * N
        dev = x.device
        if g is None:                       # only a side output was used downstream
            g = torch.zeros_like(x)
        lane = side_lane(x)
        tok = g_tokmap is not None
        G = ctx.tok_w.shape[2] if tok else 0
        gn_n, sk_n, at_n, prep_n = _stb_sizes(B, N, C, Cctx, M, heads, G)
        gop = _operand(g)            # bf16 copy if the producer left one; g itself (f32) is only the final addend
        if g.dim() == 4 and not g.is_contiguous():          # a channel slice out of a concat gradient: rows ld apart
            ld = g.stride(-2)
            gop = gop.as_strided((B, N, C), (N * ld, ld, 1)) if gop.stride(-2) == ld and gop.shape == g.shape \
                else g.contiguous().view(B, N, C)
        else:
            gop = gop.reshape(B, N, C)
        ldg, ldg32 = ops._rows_ld(gop)[1], ops._rows_ld(g)[1]
        kc = ctx.key_compaction
        want_gk = bool(ctx.needs_input_grad[1] or (same_ctx and ctx.needs_input_grad[2]))
        want_gv = bool(not same_ctx and ctx.needs_input_grad[2])
        # nothing before this block needs a gradient (the UNet's first transformer block when the UNet is frozen): only the
        # context gradient is wanted, and the call stops after the cross attention
        no_gx = STB_PRUNE_GX and not ctx.needs_input_grad[0]
        flags = ((_STB_SAME_CTX if same_ctx else 0) | (_STB_COMPACT if kc is not None else 0) | (_STB_TOKGRAD if tok else 0)
                 | (_STB_Q1_PRESCALED if P.get("q1_prescaled") else 0) | (_STB_WANT_GK if want_gk else 0)
                 | (_STB_WANT_GV if want_gv else 0) | (_STB_G_BF16 if gop.dtype == BF16 else 0) | (_STB_NO_GX if no_gx else 0))
        e = torch.empty
        gx = None if no_gx else e(B, H, W, C, device=dev, dtype=torch.float32)
        gx16 = None if no_gx else e(B, H, W, C, device=dev, dtype=BF16)
        slot = ctx.dkv_slot
        given = ctx.needs_input_grad[9] if len(ctx.needs_input_grad) > 9 else False
        dkv2 = slot if (given and slot is not None) else e(B, M, 2 * C, device=dev, dtype=BF16)
        g_ck = e(B, M, Cctx, device=dev, dtype=torch.float32) if want_gk else None
        g_cv = e(B, M, Cctx, device=dev, dtype=torch.float32) if want_gv else None
        s32 = e(2 * rows * C + at_n + prep_n + gn_n + 2 * sk_n, device=dev, dtype=torch.float32)
        s16 = e(17 * rows * C, device=dev, dtype=BF16)
        p32 = s32.data_ptr()
        at_p = p32 + 8 * rows * C
        prep_p = at_p + 4 * at_n
        gn_p = prep_p + 4 * prep_n
        sk_p = gn_p + 4 * gn_n
        if tok:
            g_tokmap = g_tokmap.contiguous()
            assert g_tokmap.dtype == torch.float32 and tuple(g_tokmap.shape) == (B, heads, N, G)
            prep = ctx.tok_prep
            # (made by prepare_tokmap_backward from exactly this gradient tensor: otherwise the block makes its own)
            if prep is not None and prep[1] == g_tokmap.data_ptr() and prep[2] == g_tokmap._version:
                prep_p = prep[0].data_ptr()
                flags |= _STB_TOKPREP_GIVEN
        dp = lambda t: 0 if t is None else t.data_ptr()          # noqa: E731
        _fw, bw, _ok, _ = _stb_weights(P, same_ctx)
        km = ctx.key_mask
        tens = (ctypes.c_void_p * 30)(gop.data_ptr(), g.data_ptr(), x.data_ptr(), gn_stats.data_ptr(), tres.data_ptr(), ln_stats.data_ptr(),
                                      qkv1.data_ptr(), obuf.data_ptr(), lse.data_ptr(), hh.data_ptr(), dp(kv1c), kv2.data_ptr(),
                                      0 if (km is None or kc is not None) else km.data_ptr(), 0 if kc is None else kc.inv_perm.data_ptr(),
                                      0 if kc is None else kc.count.data_ptr(), dp(g_tokmap) if tok else 0,
                                      ctx.tok_w.data_ptr() if tok else 0, prep_p if tok else 0, dp(gx), dp(gx16),
                                      dkv2.data_ptr(), dp(g_ck), dp(g_cv), p32, s16.data_ptr(), at_p, gn_p, sk_p if sk_n else 0,
                                      (sk_p + 4 * sk_n) if sk_n else 0, ops.gn_sync_buffer(dev))
        cfg = (ctypes.c_int * 11)(B, H, W, C, heads, M, Cctx, flags, G, ldg, ldg32)
        ops._lib.call("adap_stblock_bwd", cfg, bw, tens, 0 if lane is None else lane.side.cuda_stream, ops._stream())
        STB_CALLS[1] += 1
        g_kv = dkv2 if given else None
        if no_gx:
            STB_CALLS[2] += 1
            return None, g_ck, g_cv, None, None, None, None, None, None, g_kv, None
        return _stash16(gx, gx16), g_ck, g_cv, None, None, None, None, None, None, g_kv, None

    @staticmethod
    def backward(ctx, g, g_score=None, _g_prob=None, g_qs=None, g_tokmap=None):
        if ctx.fast:
            assert g_score is None and g_qs is None
            return SpatialTransformerFn._backward_c(ctx, g, g_tokmap)
        (x, gm, gr, t0, l1m, l1r, qkv1, o1, lse1, t1, l2m, l2r, q2, kv2, o2, lse2, t2, l3m, l3r, hh, ctx_k,
         ctx_v, xn, n1, n2, n3, gg, t3, kv1c) = ctx.saved_tensors
        P, heads = ctx.P, ctx.heads
        T = P.get("train")
        B, H, W, C = x.shape
        N = H * W
        if g is None:                       # only a side output was used downstream
            g = torch.zeros_like(x)
        # the token maps' gradient: its dq / dk-independent half (kw = w^T K, gq = dT^T Q) runs on the side lane under the
        # feed-forward's backward; the cross-attention backward then adds the rest inside its epilogues
        lane = side_lane(x)
        tok = tok_ready = None
        if g_tokmap is not None and not F32_STORAGE and TOKMAP_FOLD:
            C_ = x.shape[-1]
            g_tokmap = g_tokmap.contiguous()
            if lane is not None:
                lane.fork()
                with lane:
                    tok = (g_tokmap, ctx.tok_w, ops.attention_tokmap_prep(g_tokmap, ctx.tok_w, q2, kv2[..., :C_], heads))
                    tok_ready = lane.mark()
            else:
                tok = (g_tokmap, ctx.tok_w, ops.attention_tokmap_prep(g_tokmap, ctx.tok_w, q2, kv2[..., :C_], heads))
        gop = _operand(g)            # bf16 copy if the producer left one; g itself (f32) is only the final addend
        if g.dim() == 4 and not g.is_contiguous():          # a channel slice out of a concat gradient: rows ld apart
            ld = g.stride(-2)
            gop = gop.as_strided((B, N, C), (N * ld, ld, 1)) if gop.stride(-2) == ld and gop.shape == g.shape \
                else g.contiguous().view(B, N, C)
        else:
            gop = gop.reshape(B, N, C)
        # every f32 residual-stream gradient also gets a bf16 copy from the kernel that produces it, so the next
        # data-gradient contraction reads bf16 (LDS-DMA path) instead of converting f32 on the fly
        gt3, gt3h = _lin_bwd(gop, P["proj_out"], out_f32=True, out_bf16=True)     # [B,N,C]
        # feed-forward
        if ctx.geglu_fused:
            ghh = ops.linear_geglu_bwd(gt3h, P["ff2"], hh)                         # bf16 [B,N,8C], permuted channel order
            gn3, _ = _lin_bwd(ghh, P["ff1g"]())
        else:
            if F32_STORAGE:
                ggg, _ = _lin_bwd(gt3h, P["ff2"], out_f32=True, out_bf16=False)
                ghh = _geglu_bwd_f32(ggg, hh)
            else:
                _, ggg = _lin_bwd(gt3h, P["ff2"], out_f32=False, out_bf16=True)       # bf16 [B,N,4C]
                ghh = ops.geglu_bwd(ggg, hh)                                           # bf16 [B,N,8C]
            gn3, _ = _lin_bwd(ghh, P["ff1"])
        if T is not None:
            _dw_lin(T, "proj_out", t3, gop)
            _dw_lin(T, "ff2", gg, gt3h)
            _dw_lin(T, "ff1", n3, ghh)
            _dw_norm(T, "norm3", gn3, t2, P["norm3"][0], None, l3m, l3r, 1, 0)
        gt2, gt2h = ops.layernorm_bwd(gn3, t2, P["norm3"][0], l3m, l3r, accumulate_into=gt3, want_bf16=True)
        # cross attention
        _, go2 = _lin_bwd(gt2h, P["to_out2"], out_f32=False, out_bf16=True)
        M = kv2.shape[1]
        dq2 = torch.empty(B, N, C, device=x.device, dtype=BF16)
        given = ctx.needs_input_grad[9] if len(ctx.needs_input_grad) > 9 else False
        dkv2 = ctx.dkv_slot if (given and ctx.dkv_slot is not None) else torch.empty(B, M, 2 * C, device=x.device, dtype=BF16)
        if tok_ready is not None:
            torch.cuda.current_stream().wait_event(tok_ready)
        ops.attention_bwd(q2, kv2[..., :C], kv2[..., C:], o2, go2, lse2, heads, None, dq=dq2, dk=dkv2[..., :C],
                          dv=dkv2[..., C:], tok=tok)
        if tok is not None:
            g_tokmap = None                     # already in dq2 / dk2
        dq_acc, dk_acc = dq2, dkv2[..., :C]
        side = g_score is not None or g_qs is not None or g_tokmap is not None
        if F32_STORAGE and side:          # the side outputs' gradients go to their own (zeroed) tensors, summed in f32 below
            dq_acc, dk_acc = torch.zeros_like(dq2), torch.zeros(B, M, C, device=x.device, dtype=BF16)
        if g_score is not None or g_qs is not None:      # gradients of the captured attnscore / q side outputs
            ops.attention_capture_bwd(None if g_score is None else g_score.contiguous(),
                                      None if g_qs is None else g_qs.contiguous(), q2, kv2[..., :C], dq_acc,
                                      dk_acc, heads)
        if g_tokmap is not None:                         # ... and of the token maps (compact: no dense d attnscore)
            ops.attention_tokmap_bwd(g_tokmap.contiguous(), ctx.tok_w, q2, kv2[..., :C], dq_acc, dk_acc, heads)
        if F32_STORAGE and side:
            dq2 = (dq2.float() + dq_acc.float()).to(BF16)
            dkv2[..., :C].copy_(dkv2[..., :C].float() + dk_acc.float())
        gn2, _ = _lin_bwd(dq2, P["q2"])
        if T is not None:
            _dw_lin(T, "to_out2", o2, gt2h)
            _dw_lin(T, "q2", n2, dq2)
            _dw_lin(T, "k2", ctx_k, dkv2[..., :C])
            _dw_lin(T, "v2", ctx_v, dkv2[..., C:])
            _dw_norm(T, "norm2", gn2, t1, P["norm2"][0], None, l2m, l2r, 1, 0)
        gt1, gt1h = ops.layernorm_bwd(gn2, t1, P["norm2"][0], l2m, l2r, accumulate_into=gt2, want_bf16=True)
        g_ck = g_cv = None
        ctx_grad_ready = None

        def context_grads():
            gk = gv = None
            if ctx.same_ctx:
                if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                    gk, _ = _lin_bwd(dkv2, P["kv2"])                              # dK Wk + dV Wv in one contraction
            else:
                if ctx.needs_input_grad[1]:
                    gk, _ = _lin_bwd(dkv2[..., :C], P["k2"])
                if ctx.needs_input_grad[2]:
                    gv, _ = _lin_bwd(dkv2[..., C:], P["v2"])
            return gk, gv
        if lane is not None:
            # the context gradient is an output of the block, not an input of anything in it: beside the self-attention backward
            lane.fork()
            with lane:
                g_ck, g_cv = context_grads()
                ctx_grad_ready = lane.mark()
        else:
            g_ck, g_cv = context_grads()
        # self attention
        _, go1 = _lin_bwd(gt1h, P["to_out1"], out_f32=False, out_bf16=True)
        dqkv1 = torch.empty(B, N, 3 * C, device=x.device, dtype=BF16)
        sc1 = 0.0 if P.get("q1_prescaled") else None
        kc = ctx.key_compaction
        if kc is not None:
            dkvc = torch.empty(B, N, 2 * C, device=x.device, dtype=BF16)
            ops.attention_bwd(qkv1[..., :C], kv1c[..., :C], kv1c[..., C:], o1, go1, lse1, heads, None,
                              dq=dqkv1[..., :C], dk=dkvc[..., :C], dv=dkvc[..., C:], key_count=kc.count, scale=sc1)
            ops.gather_rows_bf16(dkvc, kc.inv_perm, out=dqkv1[..., C:])          # back to pixel order (masked keys: zeros)
        else:
            ops.attention_bwd(qkv1[..., :C], qkv1[..., C:2 * C], qkv1[..., 2 * C:], o1, go1, lse1, heads, ctx.key_mask,
                              dq=dqkv1[..., :C], dk=dqkv1[..., C:2 * C], dv=dqkv1[..., 2 * C:], scale=sc1)
        gn1, _ = _lin_bwd(dqkv1, P["qkv1"])
        if T is not None:
            _dw_lin(T, "to_out1", o1, gt1h)
            # (with the pre-scaled pack dq is the gradient with respect to c * q: the parameter's own gradient is c times it)
            _dw_lin(T, "q1", n1, dqkv1[..., :C] if sc1 is None else (dqkv1[..., :C].float() * q_prescale(C // heads)).to(BF16))
            _dw_lin(T, "k1", n1, dqkv1[..., C:2 * C])
            _dw_lin(T, "v1", n1, dqkv1[..., 2 * C:])
            _dw_norm(T, "norm1", gn1, t0, P["norm1"][0], None, l1m, l1r, 1, 0)
        gt0, gt0h = ops.layernorm_bwd(gn1, t0, P["norm1"][0], l1m, l1r, accumulate_into=gt1, want_bf16=True)
        # proj_in, GroupNorm
        gxn32, gxn = _lin_bwd(gt0h, P["proj_in"], out_f32=F32_STORAGE, out_bf16=not F32_STORAGE)
        if F32_STORAGE:
            gxn = gxn32
        gnw, gnb = P["norm"]
        if T is not None:
            _dw_lin(T, "proj_in", xn.view(B, N, C), gt0h)
            _dw_norm(T, "norm", _op16(gxn).view(B, H, W, C), x, gnw, gnb, gm, gr, 0, 0)
            _grads_done(T)
        gx, gx16 = ops.groupnorm_bwd(gxn.view(B, H, W, C), x, gnw, gnb, gm, gr, 0, out_bf16=True,
                                     add_from=g if g.dim() == 4 else g.view(B, H, W, C))   # dx + g, no clone of g
        if ctx_grad_ready is not None:
            torch.cuda.current_stream().wait_event(ctx_grad_ready)
        return _stash16(gx, gx16), g_ck, g_cv, None, None, None, None, None, None, (dkv2 if given else None), None


# ---------------------------------------------------------------------------------------------
# Downsample / Upsample / head / concat
# ---------------------------------------------------------------------------------------------
UPSAMPLE_BF16 = os.environ.get("ADAP_UPSAMPLE_BF16", "1") != "0"


class ConvFn(torch.autograd.Function):
    """conv3x3 on the f32 residual stream: stride 1 (mode 'same'), UNet Downsample (stride 2, pad 1;
    openaimodel.py:138-164) or Upsample (nearest x2 then conv; openaimodel.py:95-123)."""

    @staticmethod
    def forward(ctx, x, pk, mode, train=None):
        _note_forward(ctx, train)
        ctx.pk, ctx.mode, ctx.in_hw, ctx.train = pk, mode, (x.shape[1], x.shape[2]), train
        ctx.save_for_backward(x if train is not None else None)
        if mode == "down":
            y, _ = ops.conv2d(x, pk.fwd, pk.O4, 3, 2, 1, bias=pk.bias)
        elif mode == "up":
            if UPSAMPLE_BF16 and train is None and not F32_STORAGE and x.dtype == torch.float32 and x.shape[-1] % 64 == 0:
                # the interpolate as its own small kernel (f32 -> bf16, x2 nearest), the conv on a plain bf16 image: the
                # stencil-window kernel instead of the register-staged gather variant (same bf16 operand values either way)
                y, _ = ops.conv2d(ops.upsample2x_bf16(x), pk.fwd, pk.O4, 3, 1, 1, bias=pk.bias)
            else:
                y, _ = ops.conv2d(x, pk.fwd, pk.O4, 3, 1, 1, up=1, bias=pk.bias)
        else:
            y, _ = ops.conv2d(x, pk.fwd, pk.O4, 3, 1, 1, bias=pk.bias)
        return y

    @staticmethod
    def backward(ctx, g):
        pk, mode = ctx.pk, ctx.mode
        if mode == "same" or (mode in ("up", "down") and ctx.train is None and not F32_STORAGE):
            # the bf16 side copy the gradient's producer left (the same values, rounded the way the f32 path rounds them in
            # registers): the data-gradient conv then takes the stencil-window kernel instead of the register-staged one
            g = _operand(g)
        if ctx.train is not None:
            (x,) = ctx.saved_tensors
            _dw_conv(ctx.train, "conv", x, g, 3, 2 if mode == "down" else 1, 1, 1 if mode == "up" else 0)
            _grads_done(ctx.train)
        if mode == "down":
            gx, _ = ops.conv2d(g, pk.bwd, pk.bwd.shape[1], 3, 1, 1, up=2, out_hw=ctx.in_hw)
        elif mode == "up":
            gx = ops.sumpool2x2(_conv_bwd_data(g, pk, 3, 1))
        else:
            gx, gx16 = ops.conv2d(g, pk.bwd, pk.bwd.shape[1], 3, 1, 1, out_f32=True, out_bf16=True)
            _stash16(gx, gx16)
        return gx, None, None, None


class OutHeadFn(torch.autograd.Function):
    """``out``: GroupNorm32(1e-5) -> SiLU -> conv3x3 320->4 (openaimodel.py:693-697)."""

    @staticmethod
    def forward(ctx, h, gn, pk, train=None):
        _note_forward(ctx, train)
        _, a, m, r = ops.groupnorm_fwd(h, gn[0], gn[1], 1e-5, 1)
        y, _ = ops.conv2d(a, pk.fwd, pk.O4, 3, 1, 1, bias=pk.bias)
        ctx.gn, ctx.pk, ctx.train = gn, pk, train
        ctx.save_for_backward(h, m, r, a if train is not None else None)
        return y

    @staticmethod
    def backward(ctx, g):
        h, m, r, a = ctx.saved_tensors
        pk = ctx.pk
        g = g.contiguous()
        g16 = ops.pad_cast_bf16(g, pk.bwd.shape[2])          # 4 -> 8 channels for the K dim
        ga = _conv_bwd_data(g16, pk, 3, 1, bf16=not F32_STORAGE)
        if ctx.train is not None:
            _dw_conv(ctx.train, "conv", a, g, 3, 1, 1)
            _dw_norm(ctx.train, "gn", _op16(ga), h, ctx.gn[0], ctx.gn[1], m, r, 0, 1)
            _grads_done(ctx.train)
        gh, _ = ops.groupnorm_bwd(ga, h, ctx.gn[0], ctx.gn[1], m, r, 1)
        return gh, None, None, None


class InConvFn(torch.autograd.Function):
    """``input_blocks.0.0`` when the UNet trains: conv3x3 4 -> 320 on the noisy latent.  The latent itself gets no
    gradient (it is data); ``anchor`` is any tensor that requires grad, so that autograd calls this backward -- and with
    it, because the output now requires grad, every later block's -- even when no context gradient is wanted."""

    @staticmethod
    def forward(ctx, x, pk, train, anchor):
        _note_forward(ctx, train)
        x16 = ops.pad_cast_bf16(x, pk.I8)
        y, _ = ops.conv2d(x16, pk.fwd, pk.O4, 3, 1, 1, bias=pk.bias)
        ctx.train = train
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        _dw_conv(ctx.train, "conv", x, _operand(g), 3, 1, 1)
        _grads_done(ctx.train)
        return None, None, None, None


class ConcatFn(torch.autograd.Function):
    """torch.cat([h, skip], dim=channels) (openaimodel.py:1018) for pixel-major tensors."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[-1]
        return ops.concat2(a, b)

    @staticmethod
    def backward(ctx, g):
        g16 = _operand(g)
        ga, gb = g[..., :ctx.ca], g[..., ctx.ca:]
        if g16.dtype == BF16:                 # the consumer left a bf16 copy of the concat's gradient: slice it too
            _stash16(ga, g16[..., :ctx.ca])
            _stash16(gb, g16[..., ctx.ca:])
        return ga, gb


class SkipFn(torch.autograd.Function):
    """An encoder activation that is used twice: by the next block and, later, by the decoder's concat
    (openaimodel.py:985-1018: ``hs.append(h)`` / ``torch.cat([h, hs.pop()])``).  Forward is two views; backward adds
    the two gradients in one kernel that also writes the bf16 operand copy of the sum (instead of autograd's add)."""

    @staticmethod
    def forward(ctx, h):
        return h.view_as(h), h.view_as(h)

    @staticmethod
    def backward(ctx, g_main, g_skip):
        _GRAD16.pop(g_main.data_ptr(), None)          # copies of the addends are not needed
        _GRAD16.pop(g_skip.data_ptr(), None)
        y32, y16 = ops.add2(g_main, g_skip)
        return _stash16(y32, y16)


class CosineRowsFn(torch.autograd.Function):
    """per-row demeaned cosine loss against a sign-preserving squared reference (ldm/util.py:437-535, exponent 1 / 2 / 3), one
    HIP launch forward and one backward instead of ~25 element-wise torch launches each way."""

    @staticmethod
    def forward(ctx, x, r, demean, align, ref_grad_scale, exponent=2):
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        r2 = r.reshape(-1, r.shape[-1]).float().contiguous()
        ctx.save_for_backward(x2, r2)
        ctx.cfg = (bool(demean), bool(align), float(ref_grad_scale), x.shape, r.shape, int(exponent))
        return ops.cosine_rows(x2, r2, demean, align, exponent=exponent).reshape(x.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        x2, r2 = ctx.saved_tensors
        demean, align, rgs, xs, rs, expo = ctx.cfg
        want_dr = ctx.needs_input_grad[1] and rgs != 0
        dx, dr = ops.cosine_rows(x2, r2, demean, align, rgs, gl=g.reshape(-1).float().contiguous(),
                                 want_dx=ctx.needs_input_grad[0], want_dr=want_dr, exponent=expo)
        return (None if dx is None else dx.reshape(xs)), (None if dr is None else dr.reshape(rs)), None, None, None, None


class ElasticMatchFn(torch.autograd.Function):
    """Stage 2's elastic matching loss of one layer (ldm/util.py:2241-2368) as one C call each way (csrc/stage2loss.hip):
    (q [4, Cq, N], f [4, Cf, N], fg f32 [N]) -> (map_align, sc_ss_fg, sc_mc_bg, sc_below [1, 1, N], mc_below [1, 1, N]).
    Fixed-order sums: two runs are bit-equal (the torch expressions went through a vendor GEMM and were not)."""

    @staticmethod
    def forward(ctx, q, f, fg, cutoff, gs_q, gs_feat, gs_mix):
        q4, f4 = q.float().contiguous(), f.float().contiguous()
        fgf = fg.reshape(-1).float().contiguous()
        P2, RT, tok, out = ops.elastic_match_fwd(q4, f4, fgf, cutoff)
        ctx.save_for_backward(q4, f4, fgf, P2, RT, tok, out)
        ctx.cfg = (float(cutoff), float(gs_q), float(gs_feat), float(gs_mix), q.shape, f.shape)
        ctx.set_materialize_grads(False)
        N = fgf.numel()
        return (out[0], out[1], out[2], tok[ops.EM_SC_BELOW].view(1, 1, N), tok[ops.EM_MC_BELOW].view(1, 1, N))

    @staticmethod
    def backward(ctx, g_map, g_fg, g_bg, g_scb, g_mcb):
        q4, f4, fgf, P2, RT, tok, out = ctx.saved_tensors
        cutoff, gs_q, gs_feat, gs_mix, qs, fs = ctx.cfg
        sc = lambda g: None if g is None else g.reshape(1).float()
        vec = lambda g: None if g is None else g.reshape(-1).float().contiguous()
        dq, df = ops.elastic_match_bwd(q4, f4, fgf, cutoff, gs_q, gs_feat, gs_mix, P2, RT, tok, out, sc(g_map), sc(g_fg),
                                       sc(g_bg), vec(g_scb), vec(g_mcb))
        return (dq.reshape(qs) if ctx.needs_input_grad[0] else None, df.reshape(fs) if ctx.needs_input_grad[1] else None,
                None, None, None, None, None)


class SubjAttnTermsFn(torch.autograd.Function):
    """calc_prompt_mix_loss's two per-layer terms on the subject tokens' score maps (ddpm.py:3714-3930: the delta alignment of
    ldm/util.py:543-594 with the cosine of exponent 3, and the L1 between mean scores), one launch each way:
    a [4, H, N] -> (subj_attn_delta_align, subj_attn_norm_distill) of the layer."""

    @staticmethod
    def forward(ctx, a, gs_mix):
        a4 = a.float().contiguous()
        out, rec = ops.promptmix_attn_terms(a4, gs_mix)
        ctx.save_for_backward(a4, rec)
        ctx.cfg = (float(gs_mix), a.shape)
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_delta, g_norm):
        a4, rec = ctx.saved_tensors
        gs_mix, shape = ctx.cfg
        sc = lambda g: None if g is None else g.reshape(1).float()
        return ops.promptmix_attn_terms(a4, gs_mix, rec, sc(g_delta), sc(g_norm)).reshape(shape), None


class BgSuppressFn(torch.autograd.Function):
    """the two background-suppression terms of calc_comp_fg_bg_preserve_loss (ddpm.py:4520-4545), one launch each way:
    (a [4, H, N], sc_below [.., N], mc_below [.., N]) -> (comp_subj_bg_attn_suppress, comp_mix_bg_attn_suppress) of the layer."""

    @staticmethod
    def forward(ctx, a, scb, mcb, gs_mix):
        a4 = a.float().contiguous()
        s1, m1 = scb.reshape(-1).float().contiguous(), mcb.reshape(-1).float().contiguous()
        out, col = ops.bg_suppress(a4, s1, m1, gs_mix)
        ctx.save_for_backward(a4, s1, m1, out, col)
        ctx.cfg = (float(gs_mix), a.shape, scb.shape, mcb.shape)
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_s, g_m):
        a4, s1, m1, out, col = ctx.saved_tensors
        gs_mix, sa, ss, sm = ctx.cfg
        sc = lambda g: None if g is None else g.reshape(1).float()
        da, dscb, dmcb = ops.bg_suppress(a4, s1, m1, gs_mix, (out, col), sc(g_s), sc(g_m))
        return da.reshape(sa), dscb.reshape(ss), dmcb.reshape(sm), None


class OrthoRowsFn(torch.autograd.Function):
    """``ortho_subtract`` over the last dim (ldm/util.py:280): one HIP launch forward and one backward instead of ~8 + ~15
    element-wise / reduction torch launches (Stage 2 calls it ~60 times per micro-batch)."""

    @staticmethod
    def forward(ctx, a, b):
        a2 = a.reshape(-1, a.shape[-1]).float().contiguous()
        b2 = b.reshape(-1, b.shape[-1]).float().contiguous()
        ctx.save_for_backward(a2, b2)
        ctx.shapes = (a.shape, b.shape)
        return ops.ortho_rows(a2, b2).reshape(a.shape)

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        sa, sb = ctx.shapes
        da, db = ops.ortho_rows(a2, b2, g.reshape(a2.shape).float().contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (None if da is None else da.reshape(sa)), (None if db is None else db.reshape(sb))


class MaskHingesFn(torch.autograd.Function):
    """the four mask hinge terms of the fg/bg complementary loss (ddpm.py:4143-4238) for a stack of same-resolution
    token maps [L, B, heads, N, groups]: three launches forward, one backward (ops.mask_hinges)."""

    @staticmethod
    def forward(ctx, maps, fmask, iw, margin, margin_bg_at_mf, have_bg):
        maps = maps.contiguous()
        out, ws = ops.mask_hinges(maps, fmask, iw, margin, margin_bg_at_mf, have_bg)
        ctx.save_for_backward(maps, fmask, ws)
        ctx.cfg = (iw, float(margin), float(margin_bg_at_mf), bool(have_bg))
        return out

    @staticmethod
    def backward(ctx, gout):
        maps, fmask, ws = ctx.saved_tensors
        iw, margin, m3, have_bg = ctx.cfg
        d = ops.mask_hinges(maps, fmask, iw, margin, m3, have_bg, gout=gout.contiguous().float(), ws=ws)
        return d, None, None, None, None, None

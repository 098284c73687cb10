"""SpatialTransformer / BasicTransformerBlock / CrossAttention / GEGLU FeedForward
(reference ldm/modules/attention.py:32-59, 147-341) as parameter containers with the
reference's attribute names (so checkpoints load by key), executed by one fused block Function
(adaprompt_amd.functional.SpatialTransformerFn): pixel-major activations, fused q|k|v
projection, flash attention on the matrix cores, residual adds in GEMM epilogues."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as HF
from ... import ops
from ..util import default
from .diffusionmodules.util import zero_module


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        assert glu, "SD-1.5 uses the gated feed-forward"
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class CrossAttention(nn.Module):
    """to_q / to_k / to_v / to_out parameters + the flags the reference toggles from UNetModel.forward
    (save_attn_vars, use_conv_attn_kernel_size, is_training; attention.py:147-170)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        assert inner_dim == query_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self.save_attn_vars = False
        self.cached_activations = None
        self.use_conv_attn_kernel_size = -1
        self.infeat_size = None
        self.is_training = True


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint


class KeyMasks:
    """img_mask [B,1,h,w] -> the self-attention key mask [B, H*W] uint8 of a level (attention.py:223-232, :332: nearest
    resize, != 0) and, from ``functional.COMPACT_KEYS_MIN_N`` keys on, its compaction -- for all of ``sizes`` (the UNet's
    levels) in ONE launch (``adap_key_masks``) instead of once per transformer; a size outside ``sizes`` gets its own launch
    when it is first asked for."""

    def __init__(self, img_mask, sizes=()):
        m = img_mask
        if m.dim() != 4 or m.shape[1] != 1:
            raise ValueError(f"img_mask must be [B,1,h,w], got {tuple(m.shape)}")
        if m.device.type != "cuda":
            raise RuntimeError("KeyMasks (MI355X): the mask must live on the GPU -- there is no CPU path")
        self.img_mask = m.detach().float().contiguous()
        self._by_res = {}
        if sizes:
            self._build(list(dict.fromkeys((int(H), int(W)) for H, W in sizes)))

    def _build(self, sizes):
        import ctypes
        m = self.img_mask
        B, _, h, w = m.shape
        n = len(sizes)
        dims, offs = (ctypes.c_int * (2 * n))(), (ctypes.c_long * (4 * n))()
        nb, ni, compact = 0, 0, []
        for l, (H, W) in enumerate(sizes):
            N = H * W
            dims[2 * l], dims[2 * l + 1] = H, W
            offs[4 * l] = nb
            nb += (B * N + 15) // 16 * 16
            compact.append(N >= HF.COMPACT_KEYS_MIN_N)
            if compact[-1]:
                offs[4 * l + 1], offs[4 * l + 2], offs[4 * l + 3] = ni, ni + B * N, ni + 2 * B * N
                ni += 2 * B * N + (B + 3) // 4 * 4
            else:
                offs[4 * l + 1] = offs[4 * l + 2] = offs[4 * l + 3] = -1
        u8 = torch.empty(nb, device=m.device, dtype=torch.uint8)
        i32 = torch.empty(max(ni, 1), device=m.device, dtype=torch.int32)
        ops._lib.call("adap_key_masks", m.data_ptr(), B, h, w, n, dims, offs, u8.data_ptr(), i32.data_ptr(), ops._stream())
        for l, (H, W) in enumerate(sizes):
            N = H * W
            mask = self._by_res[(H, W)] = u8[offs[4 * l]:offs[4 * l] + B * N].view(B, N)
            if compact[l]:
                o = offs[4 * l + 1]
                self._by_res[(H, W, "compaction")] = HF.KeyCompaction(
                    mask, i32[o:o + B * N].view(B, N), i32[o + B * N:o + 2 * B * N].view(B, N), i32[o + 2 * B * N:o + 2 * B * N + B])

    def at(self, H, W):
        m = self._by_res.get((H, W))
        if m is None:
            self._build([(H, W)])
            m = self._by_res[(H, W)]
        return m

    def compaction(self, H, W):
        """(perm, inv_perm, count) of a level, all on the device (no host sync): ``perm`` [B,N] int32 lists a sample's kept
        keys first (in order), then the masked ones; ``inv_perm`` undoes it; ``count`` [B] int32 = kept keys.  A masked key's
        softmax weight is exactly 0 (attention.py:223-232 fills its score with -finfo.max), so self-attention over the first
        ``count`` rows of the permuted K / V is the same arithmetic on fewer key tiles (functional.SpatialTransformerFn).
        (A sample whose mask keeps NO key -- an image with an empty aug mask, which the data pipeline does not produce --
        would divide by an empty softmax sum: it attends to all keys instead, count = N; the reference's masked_fill gives
        it the plain average of V.)"""
        c = self._by_res.get((H, W, "compaction"))
        if c is None:
            if H * W < HF.COMPACT_KEYS_MIN_N:
                raise ValueError(f"no compaction below {HF.COMPACT_KEYS_MIN_N} keys ({H} x {W})")
            self._by_res.pop((H, W), None)
            self._build([(H, W)])
            c = self._by_res[(H, W, "compaction")]
        return c


class SpatialTransformer(nn.Module):
    """forward(x, context, mask): x pixel-major [B,H,W,C] f32; ``context`` is a tensor, a
    (v_context, k_context) tuple or -- as UNetModel passes it -- a callable returning
    ((v_context, k_context), placeholder2indices) (attention.py:184-191); ``mask`` [B,1,h,w] is
    nearest-resized to this level and masks self-attention keys (attention.py:223-232, 332)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        assert depth == 1, "SD-1.5: transformer_depth = 1"
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.n_heads = n_heads
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim)])
        self.proj_out = zero_module(nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))
        self.save_feat = False
        self._wc = HF.WeightCache()

    def _packs(self, same_ctx):
        key = None
        if HF.MODEL_STAMP is not None:       # inside UNetModel.forward: one dict per (model state, grad mode, context form)
            key = (HF.MODEL_STAMP, same_ctx, torch.is_grad_enabled(), HF.PRESCALE_Q)
            hit = self.__dict__.get("_P_cache")
            if hit is not None and hit[0] == key:
                return hit[1]
        P = self._build_packs(same_ctx)
        if key is not None:
            self.__dict__["_P_cache"] = (key, P)
        return P

    def _build_packs(self, same_ctx):
        wc, b = self._wc, self.transformer_blocks[0]
        P = {
            "norm": (self.norm.weight, self.norm.bias),
            "proj_in": wc.get("proj_in", self.proj_in.weight, self.proj_in.bias),
            "norm1": (b.norm1.weight, b.norm1.bias),
            "norm2": (b.norm2.weight, b.norm2.bias),
            "norm3": (b.norm3.weight, b.norm3.bias),
            "qkv1": wc.get("qkv1", [b.attn1.to_q.weight, b.attn1.to_k.weight, b.attn1.to_v.weight],
                           row_scales=(HF.q_prescale(b.attn1.to_q.weight.shape[0] // self.n_heads), 1.0, 1.0) if HF.PRESCALE_Q else None),
            "q1_prescaled": HF.PRESCALE_Q,
            "to_out1": wc.get("to_out1", b.attn1.to_out[0].weight, b.attn1.to_out[0].bias),
            "q2": wc.get("q2", b.attn2.to_q.weight),
            "to_out2": wc.get("to_out2", b.attn2.to_out[0].weight, b.attn2.to_out[0].bias),
            "ff1": wc.get("ff1", b.ff.net[0].proj.weight, b.ff.net[0].proj.bias),
            # the same projection with its rows in the fused-GEGLU order: built on first use (the frozen path)
            "ff1g": lambda: wc.get("ff1g", b.ff.net[0].proj.weight, b.ff.net[0].proj.bias, row_perm="geglu"),
            "ff2": wc.get("ff2", b.ff.net[2].weight, b.ff.net[2].bias),
            "proj_out": wc.get("proj_out", self.proj_out.weight, self.proj_out.bias),
        }
        P["train"] = None
        if torch.is_grad_enabled():
            P["train"] = HF.train_of(norm=self.norm, proj_in=self.proj_in, norm1=b.norm1, norm2=b.norm2, norm3=b.norm3,
                                   q1=b.attn1.to_q, k1=b.attn1.to_k, v1=b.attn1.to_v, to_out1=b.attn1.to_out[0],
                                   q2=b.attn2.to_q, k2=b.attn2.to_k, v2=b.attn2.to_v, to_out2=b.attn2.to_out[0],
                                   ff1=b.ff.net[0].proj, ff2=b.ff.net[2], proj_out=self.proj_out)
        if same_ctx:
            P["kv2"] = wc.get("kv2", [b.attn2.to_k.weight, b.attn2.to_v.weight])
        else:
            P["k2"] = wc.get("k2", b.attn2.to_k.weight)
            P["v2"] = wc.get("v2", b.attn2.to_v.weight)
        return P

    def forward(self, x, context=None, mask=None):
        B, H, W, C = x.shape
        blk = self.transformer_blocks[0]
        blk.attn2.infeat_size = (H, W)
        if callable(context):
            context, _placeholder2indices = context()
        if context is None:
            raise NotImplementedError("SpatialTransformer without a text context is not on the SD-1.5 path")
        kv = dkv_slot = None
        if isinstance(context, HF.HoistedKV):           # K | V of this layer's context already projected (HF.ContextKVFn)
            kv, dkv_slot, context = context.kv, context.dkv, context.ctx
        if isinstance(context, (list, tuple)):
            v_ctx, k_ctx = context
        else:
            v_ctx = k_ctx = context
        if blk.attn2.use_conv_attn_kernel_size is not None and blk.attn2.use_conv_attn_kernel_size > 0:
            raise NotImplementedError("conv-attn (use_conv_attn_kernel_size > 0) is off in the shipped config "
                                      "(embedding_manager.py:967) and not built")
        same = v_ctx is k_ctx
        k_ctx = k_ctx.contiguous().float()
        v_ctx = k_ctx if same else v_ctx.contiguous().float()
        key_mask = None
        if mask is not None:
            # UNetModel.forward hands over a per-resolution cache (KeyMasks): the 16 transformers run at 4 resolutions
            km = mask if hasattr(mask, "at") else KeyMasks(mask)
            # long sequences: the kept keys are moved to the front and the masked ones left out (HF.COMPACT_KEYS_MIN_N)
            key_mask = km.compaction(H, W) if H * W >= HF.COMPACT_KEYS_MIN_N else km.at(H, W)
        capture = bool(blk.attn2.save_attn_vars)
        # token weights [B, 77, G] set by UNetModel.forward when the conditioning side names the subject / background
        # token positions: the capture then also returns the per-head token maps (functional / ops.attention_capture)
        tok_w = getattr(blk.attn2, "token_weights", None) if capture else None
        if tok_w is not None and (tok_w.shape[0] != B or tok_w.shape[1] != k_ctx.shape[1]):
            tok_w = None
        res = HF.SpatialTransformerFn.apply(x.contiguous(), k_ctx, v_ctx, self._packs(same), self.n_heads, key_mask, capture,
                                            tok_w, bool(getattr(blk.attn2, "tokmap_only", False)), kv, dkv_slot)
        if capture:
            out, score, prob, qs = res[:4]
            blk.attn2.cached_activations = {"q": qs, "attn": prob, "attnscore": score}
            if tok_w is not None:
                blk.attn2.cached_activations["attnscore_tokmap"] = res[4]
        else:
            out = res
        if self.save_feat:
            raise NotImplementedError("save_feat (pre-proj_out feature tap) is not on the training hot path")
        return out

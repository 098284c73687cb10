// Block-level orchestration inside the library: the launches of one UNet ResBlock (openaimodel.py:259-279), forward and data
// gradient, issued from ONE C call -- the same entry points (adap_groupnorm_fwd/bwd, adap_conv2d_nhwc), in the order and with
// the arguments of the Python mirror (functional.ResBlockFn), so the results are bit-identical to it.  Why: the host needs
// ~26 ms to issue a training step against ~31 ms on the GPU, and a box with a 10 % slower host is host-bound; a ResBlock issued
// from Python is 5-6 ctypes calls, ~13 torch.empty and ~100 us of interpreter time each way, from here it is one call.
// Frozen weights only (no weight gradients): the training path with `unfreeze_model` stays in Python.
#include <math.h>

#include "common.h"
#include "../../include/adaprompt_hip.h"

namespace {
inline int conv3(const void* x, int x_dtype, int Cin, const void* w, const float* bias, const float* chan_add, const float* residual,
                 float* y32, void* y16, int B, int H, int W, int Cout, float* sk_ws, void* stream) {
    return adap_conv2d_nhwc(x, x_dtype, Cin, w, bias, chan_add, chan_add ? Cout : 0, residual, residual ? Cout : 0, y32, y32 ? Cout : 0,
                            y16, y16 ? Cout : 0, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 0, 1.0f, 0, sk_ws, 1, 0, 0, 0, 0, stream);
}
inline int conv1(const void* x, int x_dtype, int Cin, const void* w, const float* bias, float* y32, int B, int H, int W, int Cout,
                 float* sk_ws, void* stream) {
    return adap_conv2d_nhwc(x, x_dtype, Cin, w, bias, nullptr, 0, nullptr, 0, y32, Cout, nullptr, 0, B, H, W, Cin, H, W, Cout, 1, 1, 1, 0,
                            0, 1.0f, 0, sk_ws, 1, 0, 0, 0, 0, stream);
}
}  // namespace

// x f32 [B,H,W,Cin]; emb_out f32 [B,Cout]; packs as ops.PackedConv.fwd (c1: Cin -> Cout 3x3, c2: Cout -> Cout 3x3, sk: Cin -> Cout
// 1x1 or NULL = identity, then Cin == Cout).  Writes a1 bf16 [.., Cin], h1 bf16 [.., Cout], a2 bf16 [.., Cout], skip f32 [.., Cout]
// (only with sk), out f32 [.., Cout], stats f32 [4][B][32] = mean1, rstd1, mean2, rstd2.  gn_ws / sk_ws: the larger of the two
// GroupNorms' / three contractions' workspaces (adap_groupnorm_workspace_floats / adap_conv2d_workspace_floats); gn_sync as
// adap_groupnorm_fwd.
extern "C" int adap_resblock_fwd(const float* x, const float* emb_out, const float* g1w, const float* g1b, const float* g2w,
                                 const float* g2b, const void* c1w, const float* c1b, const void* c2w, const float* c2b, const void* skw,
                                 const float* skb, void* a1, void* h1, void* a2, float* skip, float* out, float* stats, float* gn_ws,
                                 float* sk_ws, void* gn_sync, int B, int H, int W, int Cin, int Cout, void* stream) {
    ADAP_REQUIRE(x && emb_out && g1w && g1b && g2w && g2b && c1w && c2w && a1 && h1 && a2 && out && stats && gn_ws,
                 ADAP_ERR_SHAPE, "resblock_fwd: null pointer");
    ADAP_REQUIRE((skw != nullptr) == (skip != nullptr) && (skw || Cin == Cout), ADAP_ERR_SHAPE, "resblock_fwd: skip connection");
    const int HW = H * W;
    float *m1 = stats, *r1 = stats + (size_t)B * 32, *m2 = stats + (size_t)2 * B * 32, *r2 = stats + (size_t)3 * B * 32;
    int rc;
    if ((rc = adap_groupnorm_fwd(x, 0, Cin, g1w, g1b, nullptr, 0, a1, Cin, m1, r1, gn_ws, gn_sync, B, HW, Cin, 1e-5f, 1, stream))) return rc;
    if ((rc = conv3(a1, 1, Cin, c1w, c1b, emb_out, nullptr, nullptr, h1, B, H, W, Cout, sk_ws, stream))) return rc;
    if ((rc = adap_groupnorm_fwd(h1, 1, Cout, g2w, g2b, nullptr, 0, a2, Cout, m2, r2, gn_ws, gn_sync, B, HW, Cout, 1e-5f, 1, stream)))
        return rc;
    const float* res = x;
    if (skw) {
        if ((rc = conv1(x, 0, Cin, skw, skb, skip, B, H, W, Cout, sk_ws, stream))) return rc;
        res = skip;
    }
    return conv3(a2, 1, Cout, c2w, c2b, nullptr, res, out, nullptr, B, H, W, Cout, sk_ws, stream);
}

// The data gradient of the same block.  g: the gradient of `out`, f32 (g_dtype 0) or its bf16 side copy (1), plus g32 = the f32
// gradient itself (the identity skip adds it to dx).  Packs as ops.PackedConv.bwd.  Scratch: ga2 bf16 [.., Cout], gh1 bf16
// [.., Cout], ga1 bf16 [.., Cin].  Writes gx f32 [.., Cin] and gx16 = its bf16 copy.
extern "C" int adap_resblock_bwd(const void* g, int g_dtype, const float* g32, const float* x, const void* h1, const float* stats,
                                 const float* g1w, const float* g1b, const float* g2w, const float* g2b, const void* c1wb,
                                 const void* c2wb, const void* skwb, void* ga2, void* gh1, void* ga1, float* gx, void* gx16,
                                 float* gn_ws, float* sk_ws, void* gn_sync, int B, int H, int W, int Cin, int Cout, void* stream) {
    ADAP_REQUIRE(g && g32 && x && h1 && stats && g1w && g1b && g2w && g2b && c1wb && c2wb && ga2 && gh1 && ga1 && gx && gx16 && gn_ws,
                 ADAP_ERR_SHAPE, "resblock_bwd: null pointer");
    ADAP_REQUIRE(skwb || Cin == Cout, ADAP_ERR_SHAPE, "resblock_bwd: skip connection");
    const int HW = H * W;
    const float *m1 = stats, *r1 = stats + (size_t)B * 32, *m2 = stats + (size_t)2 * B * 32, *r2 = stats + (size_t)3 * B * 32;
    int rc;
    // d a2 = conv2^T g (the flipped / transposed pack: stride 1, pad K - 1 - pad = 1)
    if ((rc = conv3(g, g_dtype, Cout, c2wb, nullptr, nullptr, nullptr, nullptr, ga2, B, H, W, Cout, sk_ws, stream))) return rc;
    if ((rc = adap_groupnorm_bwd(ga2, 1, Cout, h1, 1, Cout, g2w, g2b, m2, r2, nullptr, 0, 0, gh1, Cout, nullptr, 0, gn_ws, gn_sync, B, HW,
                                 Cout, 1, stream)))
        return rc;
    if ((rc = conv3(gh1, 1, Cout, c1wb, nullptr, nullptr, nullptr, nullptr, ga1, B, H, W, Cin, sk_ws, stream))) return rc;
    if (!skwb)          // identity skip: dx + g straight into a new tensor
        return adap_groupnorm_bwd(ga1, 1, Cin, x, 0, Cin, g1w, g1b, m1, r1, gx, Cin, 1, gx16, Cin, g32, Cin, gn_ws, gn_sync, B, HW, Cin, 1,
                                  stream);
    if ((rc = conv1(g, g_dtype, Cout, skwb, nullptr, gx, B, H, W, Cin, sk_ws, stream))) return rc;
    return adap_groupnorm_bwd(ga1, 1, Cin, x, 0, Cin, g1w, g1b, m1, r1, gx, Cin, 1, gx16, Cin, nullptr, 0, gn_ws, gn_sync, B, HW, Cin, 1,
                              stream);
}

// ---------------------------------------------------------------------------------------------
// The SpatialTransformer block (attention.py:260-341: GroupNorm -> proj_in -> [LN -> self attention] -> [LN -> cross attention]
// -> [LN -> GEGLU feed-forward] -> proj_out + x) from ONE call each way: the launches of functional.SpatialTransformerFn in its
// order and with its arguments (bit-identical to it), including the side-lane work -- the cross-attention K/V projection of the
// context tokens under the block's first half, the token-map capture, the token maps' gradient prologue, the context gradients
// -- forked and joined with events of the library's own.  Why: 16 blocks x ~40 wrapper calls, ~50 torch.empty and ~1.1 ms of
// interpreter time per step each way; with two micro-batches in flight on two streams (MicroBatchLanes) the host's issue rate
// is what bounds the step.  Frozen weights only.
// ---------------------------------------------------------------------------------------------
namespace {
// an event of the library's own for a stream fork / join.  hipStreamWaitEvent captures the record that is current when it is
// CALLED, so an event may be recorded again as soon as its waits have been issued; a small per-thread ring is plenty.
hipEvent_t next_event() {
    static thread_local hipEvent_t ring[32];
    static thread_local int made = 0, at = 0;
    if (made < 32) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        ring[made++] = e;
        return e;
    }
    at = (at + 1) & 31;
    return ring[at];
}
// `to` waits for everything issued on `from` so far
int hand_over(void* from, void* to) {
    hipEvent_t e = next_event();
    if (!e || hipEventRecord(e, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent((hipStream_t)to, e, 0) != hipSuccess)
        return adap_set_error(ADAP_ERR_HIP, "stblock: stream hand-over failed");
    return ADAP_OK;
}
// ops.linear: x [rows][Cin] (ldx) -> [rows][Cout], the 1x1 problem [1, rows, 1, Cin]
inline int lin(const void* x, int x_dtype, long ldx, const void* w, const float* bias, const float* residual, float* y32, void* y16,
               long ldy16, long rows, int Cin, int Cout, float* sk_ws, void* stream) {
    return adap_conv2d_nhwc(x, x_dtype, ldx, w, bias, nullptr, 0, residual, residual ? Cout : 0, y32, y32 ? Cout : 0, y16,
                            y16 ? ldy16 : 0, 1, (int)rows, 1, Cin, (int)rows, 1, Cout, 1, 1, 1, 0, 0, 1.0f, 0, sk_ws, 1, 0, 0, 0, 0,
                            stream);
}
}  // namespace

#define ST_TRY(call) do { if ((rc = (call))) return rc; } while (0)

extern "C" long adap_stblock_workspace_floats(int B, int N, int C, int Cctx, int M) {
    // the largest split-K workspace of the block's contractions, forward and backward (rows x channel pairs)
    const long rows = (long)B * N, crows = (long)B * M;
    long m = 0;
    auto q = [&](long r, int ci, int co) {
        const long v = r < (1L << 31) ? adap_conv2d_workspace_floats(1, (int)r, 1, ci, co, 1, 1) : 0;
        if (v > m) m = v;
    };
    q(rows, C, C); q(rows, C, 3 * C); q(rows, 3 * C, C); q(rows, C, 8 * C); q(rows, 8 * C, C); q(rows, 4 * C, C); q(rows, C, 4 * C);
    q(crows, Cctx, 2 * C); q(crows, 2 * C, Cctx); q(crows, Cctx, C); q(crows, C, Cctx);
    return m;
}

// cfg (host ints): B, H, W, C, heads, M (context tokens), Cctx, flags (ADAP_STB_*).
// w: device pointers, ADAP_STW_* order; t: device pointers (and two host handles), ADAP_STF_* order (include/adaprompt_hip.h).
extern "C" int adap_stblock_fwd(const int* cfg, const void* const* w, void* const* t, void* lane, void* stream) {
    ADAP_REQUIRE(cfg && w && t, ADAP_ERR_SHAPE, "stblock_fwd: null pointer");
    const int B = cfg[0], H = cfg[1], W = cfg[2], C = cfg[3], heads = cfg[4], M = cfg[5], Cctx = cfg[6], flags = cfg[7];
    ADAP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && heads > 0 && C % heads == 0 && M > 0 && Cctx > 0, ADAP_ERR_SHAPE,
                 "stblock_fwd: dims");
    const int N = H * W, d = C / heads;
    const long rows = (long)B * N, crows = (long)B * M;
    const bool same_ctx = flags & ADAP_STB_SAME_CTX, compact = flags & ADAP_STB_COMPACT, capture = flags & ADAP_STB_CAPTURE;
    // (the Python mirror hands float(d) ** -0.5, a double, to a C float: the same rounding here)
    const float scale2 = (float)pow((double)d, -0.5), scale1 = (flags & ADAP_STB_Q1_PRESCALED) ? 0.0f : scale2;
    const float* x = (const float*)t[ADAP_STF_X];
    const float *ctx_k = (const float*)t[ADAP_STF_CTX_K], *ctx_v = (const float*)t[ADAP_STF_CTX_V];
    uint16_t* kv2 = (uint16_t*)t[ADAP_STF_KV2];
    float* gn_stats = (float*)t[ADAP_STF_GN_STATS];
    float* tres = (float*)t[ADAP_STF_TRES];
    float* ln_stats = (float*)t[ADAP_STF_LN_STATS];
    uint16_t* qkv1 = (uint16_t*)t[ADAP_STF_QKV1];
    uint16_t* obuf = (uint16_t*)t[ADAP_STF_OBUF];
    float* lse = (float*)t[ADAP_STF_LSE];
    uint16_t* hh = (uint16_t*)t[ADAP_STF_HH];
    uint16_t* kv1c = (uint16_t*)t[ADAP_STF_KV1C];
    float* out = (float*)t[ADAP_STF_OUT];
    uint16_t* scr = (uint16_t*)t[ADAP_STF_SCRATCH16];
    float* gn_ws = (float*)t[ADAP_STF_GN_WS];
    float* sk_ws = (float*)t[ADAP_STF_SK_WS];
    float* sk_ws_lane = (float*)t[ADAP_STF_SK_WS_LANE];
    ADAP_REQUIRE(x && ctx_k && kv2 && gn_stats && tres && ln_stats && qkv1 && obuf && lse && hh && out && scr && gn_ws,
                 ADAP_ERR_SHAPE, "stblock_fwd: null tensor");
    ADAP_REQUIRE(same_ctx || ctx_v, ADAP_ERR_SHAPE, "stblock_fwd: split context without a value context");
    ADAP_REQUIRE(!compact || (kv1c && t[ADAP_STF_PERM] && t[ADAP_STF_KEY_COUNT]), ADAP_ERR_SHAPE, "stblock_fwd: key compaction inputs");
    ADAP_REQUIRE(!capture || (t[ADAP_STF_TOK_W] && t[ADAP_STF_TOKMAP]), ADAP_ERR_SHAPE, "stblock_fwd: capture needs token weights + maps");
    void* side = lane ? lane : stream;
    float* sk_side = lane ? sk_ws_lane : sk_ws;
    const size_t rc_ = (size_t)rows * C;
    float *t0 = tres, *t1 = tres + rc_, *t2 = tres + 2 * rc_;
    float *l1m = ln_stats, *l1r = ln_stats + rows, *l2m = ln_stats + 2 * rows, *l2r = ln_stats + 3 * rows, *l3m = ln_stats + 4 * rows,
          *l3r = ln_stats + 5 * rows;
    uint16_t *o1 = obuf, *q2 = obuf + rc_, *o2 = obuf + 2 * rc_;
    float *lse1 = lse, *lse2 = lse + (size_t)B * heads * N;
    uint16_t *xn = scr, *nn = scr + rc_, *gg = scr + 2 * rc_, *t3 = scr + 6 * rc_;
    int rc;
    // side lane: the cross-attention K/V projection of the context tokens (k | v halves of one [B, M, 2C] tensor) -- unless the
    // caller made it already (ADAP_STB_KV_GIVEN: the layers' projections hoisted in front of the UNet as batched launches)
    hipEvent_t kv_ready = nullptr;
    if (!(flags & ADAP_STB_KV_GIVEN)) {
        if (lane) ST_TRY(hand_over(stream, lane));
        if (same_ctx) {
            ST_TRY(lin(ctx_k, 0, Cctx, w[ADAP_STW_KV2], nullptr, nullptr, nullptr, kv2, 2 * C, crows, Cctx, 2 * C, sk_side, side));
        } else {
            ST_TRY(lin(ctx_k, 0, Cctx, w[ADAP_STW_KV2], nullptr, nullptr, nullptr, kv2, 2 * C, crows, Cctx, C, sk_side, side));
            ST_TRY(lin(ctx_v, 0, Cctx, w[ADAP_STW_V2], nullptr, nullptr, nullptr, kv2 + C, 2 * C, crows, Cctx, C, sk_side, side));
        }
        if (lane) {
            kv_ready = next_event();
            ADAP_REQUIRE(kv_ready && hipEventRecord(kv_ready, (hipStream_t)lane) == hipSuccess, ADAP_ERR_HIP, "stblock_fwd: event");
        }
    }
    // main chain
    ST_TRY(adap_groupnorm_fwd(x, 0, C, (const float*)w[ADAP_STW_GN_G], (const float*)w[ADAP_STW_GN_B], nullptr, 0, xn, C, gn_stats,
                              gn_stats + (size_t)B * 32, gn_ws, t[ADAP_STF_GN_SYNC], B, N, C, 1e-6f, 0, stream));
    ST_TRY(lin(xn, 1, C, w[ADAP_STW_PIN_W], (const float*)w[ADAP_STW_PIN_B], nullptr, t0, nullptr, 0, rows, C, C, sk_ws, stream));
    ST_TRY(adap_layernorm_fwd(t0, C, (const float*)w[ADAP_STW_LN1_G], (const float*)w[ADAP_STW_LN1_B], nn, C, l1m, l1r, rows, C, 1e-5f,
                              stream));
    ST_TRY(lin(nn, 1, C, w[ADAP_STW_QKV], nullptr, nullptr, nullptr, qkv1, 3 * C, rows, C, 3 * C, sk_ws, stream));
    if (compact) {
        ST_TRY(adap_gather_rows_bf16(qkv1 + C, 3 * C, (const int*)t[ADAP_STF_PERM], kv1c, 2 * C, B, N, N, 2 * C, stream));
        ST_TRY(adap_attention_fwd(qkv1, 3 * C, kv1c, 2 * C, kv1c + C, 2 * C, nullptr, (const int*)t[ADAP_STF_KEY_COUNT], o1, C, lse1, B,
                                  heads, N, N, d, scale1, stream));
    } else {
        ST_TRY(adap_attention_fwd(qkv1, 3 * C, qkv1 + C, 3 * C, qkv1 + 2 * C, 3 * C, (const uint8_t*)t[ADAP_STF_KEY_MASK], nullptr, o1, C,
                                  lse1, B, heads, N, N, d, scale1, stream));
    }
    ST_TRY(lin(o1, 1, C, w[ADAP_STW_OUT1_W], (const float*)w[ADAP_STW_OUT1_B], t0, t1, nullptr, 0, rows, C, C, sk_ws, stream));
    ST_TRY(adap_layernorm_fwd(t1, C, (const float*)w[ADAP_STW_LN2_G], (const float*)w[ADAP_STW_LN2_B], nn, C, l2m, l2r, rows, C, 1e-5f,
                              stream));
    ST_TRY(lin(nn, 1, C, w[ADAP_STW_Q2], nullptr, nullptr, nullptr, q2, C, rows, C, C, sk_ws, stream));
    if (kv_ready)
        ADAP_REQUIRE(hipStreamWaitEvent((hipStream_t)stream, kv_ready, 0) == hipSuccess, ADAP_ERR_HIP, "stblock_fwd: wait");
    ST_TRY(adap_attention_fwd(q2, C, kv2, 2 * C, kv2 + C, 2 * C, nullptr, nullptr, o2, C, lse2, B, heads, N, M, d, scale2, stream));
    if (capture && !(flags & ADAP_STB_CAPTURE_DEFERRED)) {
        // the side outputs are read by the losses after the UNet's forward: beside the rest of the block, joined by the caller
        if (lane) ST_TRY(hand_over(stream, lane));
        ST_TRY(adap_attention_capture(q2, C, kv2, 2 * C, nullptr, nullptr, nullptr, (const float*)t[ADAP_STF_TOK_W],
                                      (float*)t[ADAP_STF_TOKMAP], cfg[8], B, heads, N, M, d, scale2, side));
    }
    ST_TRY(lin(o2, 1, C, w[ADAP_STW_OUT2_W], (const float*)w[ADAP_STW_OUT2_B], t1, t2, nullptr, 0, rows, C, C, sk_ws, stream));
    ST_TRY(adap_layernorm_fwd(t2, C, (const float*)w[ADAP_STW_LN3_G], (const float*)w[ADAP_STW_LN3_B], nn, C, l3m, l3r, rows, C, 1e-5f,
                              stream));
    ST_TRY(adap_linear_geglu_fwd(nn, C, w[ADAP_STW_FF1G_W], (const float*)w[ADAP_STW_FF1G_B], hh, 8 * C, gg, 4 * C, rows, C, 8 * C, stream));
    ST_TRY(lin(gg, 1, 4 * C, w[ADAP_STW_FF2_W], (const float*)w[ADAP_STW_FF2_B], t2, nullptr, t3, C, rows, 4 * C, C, sk_ws, stream));
    return lin(t3, 1, C, w[ADAP_STW_POUT_W], (const float*)w[ADAP_STW_POUT_B], x, out, nullptr, 0, rows, C, C, sk_ws, stream);
}

// The data gradient of the same block (functional.SpatialTransformerFn.backward, frozen weights, fused GEGLU, bf16 storage).
// cfg as above (+ cfg[8] = G token groups, cfg[9] / cfg[10] = leading dimensions of gop / g32 in elements);  wb: ADAP_STWB_* (data-gradient packs,
// norm gains);  t: ADAP_STG_*.  Writes gx f32 / gx16 bf16 [rows][C] and, when asked for, the context gradients (side lane,
// joined before return).
extern "C" int adap_stblock_bwd(const int* cfg, const void* const* wb, void* const* t, void* lane, void* stream) {
    ADAP_REQUIRE(cfg && wb && t, ADAP_ERR_SHAPE, "stblock_bwd: null pointer");
    const int B = cfg[0], H = cfg[1], W = cfg[2], C = cfg[3], heads = cfg[4], M = cfg[5], Cctx = cfg[6], flags = cfg[7], G = cfg[8];
    const long ldg = cfg[9], ldg32 = cfg[10];
    ADAP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && heads > 0 && C % heads == 0 && M > 0 && Cctx > 0 && ldg >= C && ldg32 >= C,
                 ADAP_ERR_SHAPE, "stblock_bwd: dims");
    const int N = H * W, d = C / heads;
    const long rows = (long)B * N, crows = (long)B * M;
    const bool same_ctx = flags & ADAP_STB_SAME_CTX, compact = flags & ADAP_STB_COMPACT, tok = flags & ADAP_STB_TOKGRAD;
    const bool want_gk = flags & ADAP_STB_WANT_GK, want_gv = flags & ADAP_STB_WANT_GV, g_bf16 = flags & ADAP_STB_G_BF16;
    // the block's input needs no gradient (the UNet's first transformer block: nothing trainable lies before it): stop after the
    // cross attention -- the context gradient is all that is wanted; the self attention, proj_in and the GroupNorm are skipped
    const bool no_gx = flags & ADAP_STB_NO_GX;
    const float scale2 = (float)pow((double)d, -0.5), scale1 = (flags & ADAP_STB_Q1_PRESCALED) ? 0.0f : scale2;
    const void* gop = t[ADAP_STG_GOP];
    const float* g32 = (const float*)t[ADAP_STG_G32];
    const float* x = (const float*)t[ADAP_STG_X];
    const float* gn_stats = (const float*)t[ADAP_STG_GN_STATS];
    const float* tres = (const float*)t[ADAP_STG_TRES];
    const float* ln_stats = (const float*)t[ADAP_STG_LN_STATS];
    const uint16_t* qkv1 = (const uint16_t*)t[ADAP_STG_QKV1];
    const uint16_t* obuf = (const uint16_t*)t[ADAP_STG_OBUF];
    const float* lse = (const float*)t[ADAP_STG_LSE];
    const uint16_t* hh = (const uint16_t*)t[ADAP_STG_HH];
    const uint16_t* kv1c = (const uint16_t*)t[ADAP_STG_KV1C];
    const uint16_t* kv2 = (const uint16_t*)t[ADAP_STG_KV2];
    float* gx = (float*)t[ADAP_STG_GX];
    uint16_t* gx16 = (uint16_t*)t[ADAP_STG_GX16];
    uint16_t* dkv2 = (uint16_t*)t[ADAP_STG_DKV2];
    float* s32 = (float*)t[ADAP_STG_SCRATCH32];
    uint16_t* s16 = (uint16_t*)t[ADAP_STG_SCRATCH16];
    float* gn_ws = (float*)t[ADAP_STG_GN_WS];
    float* sk_ws = (float*)t[ADAP_STG_SK_WS];
    float* sk_ws_lane = (float*)t[ADAP_STG_SK_WS_LANE];
    float* at_ws = (float*)t[ADAP_STG_ATTN_WS];
    ADAP_REQUIRE(gop && g32 && x && gn_stats && tres && ln_stats && qkv1 && obuf && lse && hh && kv2 && (no_gx || (gx && gx16)) && dkv2 &&
                 s32 && s16 && gn_ws && at_ws, ADAP_ERR_SHAPE, "stblock_bwd: null tensor");
    ADAP_REQUIRE(!compact || (kv1c && t[ADAP_STG_INV_PERM] && t[ADAP_STG_KEY_COUNT]), ADAP_ERR_SHAPE, "stblock_bwd: key compaction inputs");
    ADAP_REQUIRE(!tok || (t[ADAP_STG_D_TOKMAP] && t[ADAP_STG_TOK_W] && t[ADAP_STG_TOK_PREP] && G >= 1 && G <= 4), ADAP_ERR_SHAPE,
                 "stblock_bwd: token-map gradient inputs");
    ADAP_REQUIRE((!want_gk || t[ADAP_STG_G_CK]) && (!want_gv || same_ctx || t[ADAP_STG_G_CV]), ADAP_ERR_SHAPE,
                 "stblock_bwd: context gradient outputs");
    void* side = lane ? lane : stream;
    float* sk_side = lane ? sk_ws_lane : sk_ws;
    const size_t rc_ = (size_t)rows * C;
    const float *t0 = tres, *t1 = tres + rc_, *t2 = tres + 2 * rc_;
    const float *l1m = ln_stats, *l1r = ln_stats + rows, *l2m = ln_stats + 2 * rows, *l2r = ln_stats + 3 * rows, *l3m = ln_stats + 4 * rows,
                *l3r = ln_stats + 5 * rows;
    const uint16_t *o1 = obuf, *q2 = obuf + rc_, *o2 = obuf + 2 * rc_;
    const float *lse1 = lse, *lse2 = lse + (size_t)B * heads * N;
    // scratch: f32 [gt | gn] (the running residual-stream gradient and a projection's input gradient);
    // bf16 [gth | go | dq2 | gxn (1 C each) | ghh (8 C) | dqkv1 (3 C) | dkvc (2 C)]
    float *gt = s32, *gn = s32 + rc_;
    uint16_t *gth = s16, *go = s16 + rc_, *dq2 = s16 + 2 * rc_, *gxn = s16 + 3 * rc_, *ghh = s16 + 4 * rc_, *dqkv1 = s16 + 12 * rc_,
             *dkvc = s16 + 15 * rc_;
    int rc;
    hipEvent_t tok_ready = nullptr;
    if (tok && !(flags & ADAP_STB_TOKPREP_GIVEN)) {
        // the token maps' gradient, its dq / dk-independent half (kw = w^T K, gq = dT^T Q): beside the feed-forward's backward
        if (lane) ST_TRY(hand_over(stream, lane));
        ST_TRY(adap_attention_tokmap_prep((const float*)t[ADAP_STG_D_TOKMAP], (const float*)t[ADAP_STG_TOK_W], q2, C, kv2, 2 * C,
                                          (float*)t[ADAP_STG_TOK_PREP], B, heads, N, M, d, G, side));
        if (lane) {
            tok_ready = next_event();
            ADAP_REQUIRE(tok_ready && hipEventRecord(tok_ready, (hipStream_t)lane) == hipSuccess, ADAP_ERR_HIP, "stblock_bwd: event");
        }
    }
    // proj_out
    ST_TRY(lin(gop, g_bf16 ? 1 : 0, ldg, wb[ADAP_STWB_POUT], nullptr, nullptr, gt, gth, C, rows, C, C, sk_ws, stream));
    // feed-forward
    ST_TRY(adap_linear_geglu_bwd(gth, C, wb[ADAP_STWB_FF2], hh, 8 * C, ghh, 8 * C, rows, C, 4 * C, stream));
    // (a long-K data gradient: where it goes out split, its reduce pass does the LayerNorm backward on the rows it sums --
    // adap_conv2d_next_ln_bwd: the same numbers, one launch less on the chain)
    ST_TRY(adap_conv2d_next_ln_bwd(t2, C, (const float*)wb[ADAP_STWB_LN3_G], l3m, l3r, gt, C, 1, gth, C));
    ST_TRY(lin(ghh, 1, 8 * C, wb[ADAP_STWB_FF1G], nullptr, nullptr, gn, nullptr, 0, rows, 8 * C, C, sk_ws, stream));
    if (!adap_conv2d_last_ln_bwd())
        ST_TRY(adap_layernorm_bwd(gn, C, t2, C, (const float*)wb[ADAP_STWB_LN3_G], l3m, l3r, gt, C, 1, gth, C, rows, C, stream));
    // cross attention
    ST_TRY(lin(gth, 1, C, wb[ADAP_STWB_OUT2], nullptr, nullptr, nullptr, go, C, rows, C, C, sk_ws, stream));
    if (tok_ready)
        ADAP_REQUIRE(hipStreamWaitEvent((hipStream_t)stream, tok_ready, 0) == hipSuccess, ADAP_ERR_HIP, "stblock_bwd: wait");
    if (tok) {
        ST_TRY(adap_attention_bwd_tok(q2, C, kv2, 2 * C, kv2 + C, 2 * C, nullptr, nullptr, o2, C, go, C, lse2, at_ws, nullptr, dq2, C, nullptr,
                                      dkv2, 2 * C, nullptr, dkv2 + C, 2 * C, B, heads, N, M, d, scale2, (const float*)t[ADAP_STG_D_TOKMAP],
                                      (const float*)t[ADAP_STG_TOK_W], (const float*)t[ADAP_STG_TOK_PREP], G, stream));
    } else {
        ST_TRY(adap_attention_bwd(q2, C, kv2, 2 * C, kv2 + C, 2 * C, nullptr, nullptr, o2, C, go, C, lse2, at_ws, nullptr, dq2, C, nullptr, dkv2,
                                  2 * C, nullptr, dkv2 + C, 2 * C, B, heads, N, M, d, scale2, stream));
    }
    if (!no_gx) {
        ST_TRY(lin(dq2, 1, C, wb[ADAP_STWB_Q2], nullptr, nullptr, gn, nullptr, 0, rows, C, C, sk_ws, stream));
        ST_TRY(adap_layernorm_bwd(gn, C, t1, C, (const float*)wb[ADAP_STWB_LN2_G], l2m, l2r, gt, C, 1, gth, C, rows, C, stream));
    }
    // the context gradient is an output of the block, not an input of anything in it: beside the self-attention backward
    const bool ctx_grads = want_gk || want_gv;
    if (ctx_grads) {
        if (lane) ST_TRY(hand_over(stream, lane));
        if (same_ctx) {             // dK Wk + dV Wv in one contraction
            ST_TRY(lin(dkv2, 1, 2 * C, wb[ADAP_STWB_KV2], nullptr, nullptr, (float*)t[ADAP_STG_G_CK], nullptr, 0, crows, 2 * C, Cctx, sk_side,
                       side));
        } else {
            if (want_gk)
                ST_TRY(lin(dkv2, 1, 2 * C, wb[ADAP_STWB_KV2], nullptr, nullptr, (float*)t[ADAP_STG_G_CK], nullptr, 0, crows, C, Cctx, sk_side,
                           side));
            if (want_gv)
                ST_TRY(lin(dkv2 + C, 1, 2 * C, wb[ADAP_STWB_V2], nullptr, nullptr, (float*)t[ADAP_STG_G_CV], nullptr, 0, crows, C, Cctx,
                           sk_side, side));
        }
    }
    if (no_gx) {
        if (ctx_grads && lane) ST_TRY(hand_over(lane, stream));
        return ADAP_OK;
    }
    // self attention
    ST_TRY(lin(gth, 1, C, wb[ADAP_STWB_OUT1], nullptr, nullptr, nullptr, go, C, rows, C, C, sk_ws, stream));
    if (compact) {
        ST_TRY(adap_attention_bwd(qkv1, 3 * C, kv1c, 2 * C, kv1c + C, 2 * C, nullptr, (const int*)t[ADAP_STG_KEY_COUNT], o1, C, go, C, lse1,
                                  at_ws, nullptr, dqkv1, 3 * C, nullptr, dkvc, 2 * C, nullptr, dkvc + C, 2 * C, B, heads, N, N, d, scale1, stream));
        // back to pixel order (masked keys: zeros)
        ST_TRY(adap_gather_rows_bf16(dkvc, 2 * C, (const int*)t[ADAP_STG_INV_PERM], dqkv1 + C, 3 * C, B, N, N, 2 * C, stream));
    } else {
        ST_TRY(adap_attention_bwd(qkv1, 3 * C, qkv1 + C, 3 * C, qkv1 + 2 * C, 3 * C, (const uint8_t*)t[ADAP_STG_KEY_MASK], nullptr, o1, C, go, C,
                                  lse1, at_ws, nullptr, dqkv1, 3 * C, nullptr, dqkv1 + C, 3 * C, nullptr, dqkv1 + 2 * C, 3 * C, B, heads, N, N, d,
                                  scale1, stream));
    }
    ST_TRY(adap_conv2d_next_ln_bwd(t0, C, (const float*)wb[ADAP_STWB_LN1_G], l1m, l1r, gt, C, 1, gth, C));
    ST_TRY(lin(dqkv1, 1, 3 * C, wb[ADAP_STWB_QKV], nullptr, nullptr, gn, nullptr, 0, rows, 3 * C, C, sk_ws, stream));
    if (!adap_conv2d_last_ln_bwd())
        ST_TRY(adap_layernorm_bwd(gn, C, t0, C, (const float*)wb[ADAP_STWB_LN1_G], l1m, l1r, gt, C, 1, gth, C, rows, C, stream));
    // proj_in, GroupNorm (dx + g: the block's residual path, no clone of g)
    ST_TRY(lin(gth, 1, C, wb[ADAP_STWB_PIN], nullptr, nullptr, nullptr, gxn, C, rows, C, C, sk_ws, stream));
    ST_TRY(adap_groupnorm_bwd(gxn, 1, C, x, 0, C, (const float*)wb[ADAP_STWB_GN_G], (const float*)wb[ADAP_STWB_GN_B], gn_stats,
                              gn_stats + (size_t)B * 32, gx, C, 1, gx16, C, g32, ldg32, gn_ws, t[ADAP_STG_GN_SYNC], B, N, C, 0, stream));
    if (ctx_grads && lane) ST_TRY(hand_over(lane, stream));
    return ADAP_OK;
}

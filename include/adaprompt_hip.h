/*
 * adaprompt_hip.h -- C ABI of libadaprompt_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * SD-1.5 UNet denoising + distillation training hot path of askerlee/adaprompt (AdaFace).
 *
 * The reference is pure Python on stock torch ops and has no FFI of its own;
 * each entry point below names the reference code (file:line under /root/reference) whose arithmetic
 * it replaces.  How a reference maintainer binds them (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: device pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - every function returns 0 on success or a negative ADAP_ERR_* code; adap_last_error() holds
 *     the message (thread local).  Nothing throws; nothing falls back to a CPU path.
 *   - the caller owns every buffer (including workspaces); kernels never allocate, free or
 *     synchronise; all work is enqueued on `stream`; entry points are re-entrant.
 *   - activations are PIXEL-MAJOR ("NHWC" / token-major): element (b, y, x, c) of a [B,H,W,C] tensor
 *     is at ((b*H + y)*W + x)*ld + c, where ld >= C is the leading dimension passed with the pointer.
 *     SpatialTransformer's 'b c h w -> b (h w) c' (attention.py:325) is therefore free.
 *   - dtype codes: 0 = f32, 1 = bf16.  The fp32 residual stream of the reference is kept in f32;
 *     bf16 is used only for MFMA operands (outputs of norm / activation / projection kernels).
 *   - conv / linear weights are pre-packed bf16, see adap_pack_conv_weight.
 */
#ifndef ADAPROMPT_HIP_H
#define ADAPROMPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAP_OK 0
#define ADAP_ERR_SHAPE (-1)
#define ADAP_ERR_ALIGN (-2)
#define ADAP_ERR_UNSUPPORTED (-3)
#define ADAP_ERR_HIP (-4)

#define ADAP_F32 0
#define ADAP_BF16 1

const char* adap_last_error(void);
int adap_abi_version(void);
int adap_device_info(char* name, int name_cap, int* num_cus, char* arch, int arch_cap);
/* Diagnostic: `blocks` workgroups of `threads` threads (~100 live registers per lane) stay resident for `usec` microseconds
 * (bounded by the 100 MHz real-time clock) -- the footprint of a collective kernel running beside the compute stream
 * (tests/test_parallel_gpu.py: the single-launch GroupNorm next to it).  sink: any device float or NULL. */
int adap_debug_occupy(int blocks, int threads, int usec, float* sink, void* stream);

/* Split-K planning scale in percent (default 100, clamped to 0..100; 0 = never split): the plan aims at ~2 workgroups per CU for
 * a launch that has the chip to itself; a caller that keeps two streams busy (the micro-batch lanes) sets a smaller target.
 * percent < 0 only reads.  Returns the previous value.  adap_conv2d_workspace_floats always sizes for 100. */
int adap_conv_ksplit_scale(int percent);
/* ---------------------------------------------------------------------------------------------
 * Contractions: nn.Conv2d 3x3 / 1x1 and nn.Linear on the matrix cores (implicit GEMM).
 * Replaces: ResBlock convs openaimodel.py:205-236,259-279; Downsample :138-164 (stride 2, pad 1);
 * Upsample :95-123 (up = 1: nearest x2 fused into the gather); SpatialTransformer proj_in/proj_out
 * attention.py:302-341; CrossAttention to_q/to_k/to_v/to_out :176-243; GEGLU/FeedForward :32-59;
 * VAE ResnetBlock / conv_in / conv_out / nin_shortcut model.py:83-142,474-499 and its Downsample
 * (pad = 0 with Hout = Hin/2 reproduces F.pad(0,1,0,1) + stride 2, model.py:73-77); quant_conv
 * autoencoder.py:302,326; and, with a mode-1 weight pack, the data gradient of each of them
 * (up = 2: zero-insert gather = transposed stride-2 conv).
 *
 *   y[b,oy,ox,co] = alpha * sum_{ky,kx,ci} x[b, oy*stride-pad+ky, ox*stride-pad+kx, ci] * w[ky,kx,co,ci]
 *                   + bias[co] + chan_add[b*ld_ca + co] + residual[pixel*ldr + co]
 *
 * x: f32 or bf16 (x_dtype); w_packed: bf16 [KH*KW][Cout][Cin]; Cin % 8 == 0, Cout % 4 == 0.
 * y32 and/or y16 receive the result.  Split-K: ksplit = 1 none; ksplit > 1 forced; ksplit = 0 lets the library
 * split the K loop over grid.z when the problem has too few pixel tiles to fill 256 CUs (only if splitk_ws is
 * given).  Splits write fp32 slabs to splitk_ws (adap_conv2d_workspace_floats floats, or ksplit*M*Cout when
 * forced) and a second kernel sums them in a fixed order and applies the epilogue: no float atomics.
 * nbatch > 1 runs independent problems (grid.y) with the given element strides (VAE mid attention).
 */
int adap_conv2d_nhwc(const void* x, int x_dtype, long ldx, const void* w_packed,
                     const float* bias, const float* chan_add, long ld_ca,
                     const float* residual, long ldr,
                     float* y32, long ldy32, void* y16, long ldy16,
                     int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                     int KH, int KW, int stride, int pad, int up,
                     float alpha, int ksplit, float* splitk_ws,
                     int nbatch, long bs_x, long bs_w, long bs_y32, long bs_y16,
                     void* stream);

long adap_conv2d_workspace_floats(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW);
/* which kernel variant the calling thread's last adap_conv2d_nhwc dispatched to (1000*variant + channel tile; profiling aid) */
int adap_conv2d_last_variant(void);
/* Diagnostic (tools/clock_probe.py): with a non-NULL device buffer of 2 * workgroups uint64, every workgroup of the
 * stencil-window kernel stores the shader-clock ticks (s_memtime) and the 100 MHz real-time ticks (s_memrealtime) its
 * K loop took -- the in-kernel clock the chip holds under this kernel's load.  NULL (the default) switches it off. */
int adap_conv2d_set_clock_probe(void* buf);
/* Diagnostic (tools/gemm_probe.py): force the kernel variant of the calling thread's next adap_conv2d_nhwc calls:
 * kind 0 = automatic (production), 1 = conv_gemm_kernel, 2 = conv_gemm_ring_kernel<256,.,3>, 3 = conv_gemm_ring_kernel<128,.,4>;
 * bn 0 = automatic, else the channel tile (64 / 128 / 160).  A combination the problem does not admit is ignored. */
int adap_conv2d_debug_force(int kind, int bn);
/* GroupNorm statistics of a contraction's OUTPUT from its epilogue (model.py:34-40 GroupNorm(32) after a conv3x3, openaimodel.py
 * normalization(): the statistics pass over the tensor -- 4 or 2 bytes per element of HBM reads on the VAE's 134-537 MB
 * tensors -- disappears).  adap_conv2d_next_gn_partial arms the calling thread's NEXT adap_conv2d_nhwc call (one shot):
 * partial f32 [B][Hout * Wout / 64][32][2] (one record per 64 output pixels -- a wave's share of a tile -- and group),
 * channels_per_group = Cout / 32 in {4, 8, 16}.  After that call adap_conv2d_last_gn_chunks() is the number of records per image
 * it wrote (Hout * Wout / 64), or 0 if the call could not provide them (split-K, a 64- or 160-wide channel tile, Cout % 128 != 0,
 * an image that is not a whole number of pixel tiles) -- the caller then runs adap_groupnorm_fwd as usual.  Records hold (sum,
 * sum of squares) of the values the epilogue stores, accumulated in a fixed order (bit-reproducible); for a bf16-only output they
 * are those of the f32 values before the rounding. */
int adap_conv2d_next_gn_partial(float* partial, int channels_per_group);
int adap_conv2d_last_gn_chunks(void);

/* The LayerNorm BACKWARD that consumes a contraction's output (attention.py:267-269: norm1 / norm3 behind the data gradients of
 * to_q|k|v and ff.net.0.proj), folded into that contraction's split-K reduce pass.  adap_conv2d_next_ln_bwd arms the calling
 * thread's NEXT adap_conv2d_nhwc call (one shot; arguments as adap_layernorm_bwd, with the call's output as dy and Cout as D).  If
 * that call goes out split (and Cout <= 1280, no chan_add), its reduce launch does the LayerNorm arithmetic on every row it has
 * just summed -- bit-identical to the reduce followed by adap_layernorm_bwd -- and the call's y32 / y16 are NOT written;
 * adap_conv2d_last_ln_bwd() is then 1.  Otherwise it is 0, the outputs are written as usual and the caller runs
 * adap_layernorm_bwd itself. */
int adap_conv2d_next_ln_bwd(const float* x, long ldx, const float* gamma, const float* mean, const float* rstd, float* dx,
                            long lddx, int accumulate, void* dx16, long lddx16);
int adap_conv2d_last_ln_bwd(void);

/* conv3x3 (stride 1, pad 1) on an RGB image -- the VAE encoder's conv_in (model.py:426, 468) -- with the whole 3 x 3 x 3 patch of a
 * pixel as ONE K step of the matrix core (k = 3 * tap + channel) instead of nine steps of a channel dimension padded 3 -> 64.
 * x_hwc f32 [B][H][W][ldx >= 3]; w_packed = adap_pack_conv_weight's forward pack [9][Cout][8] of the [Cout][3][3][3] weight;
 * Cout in {32, 64, 128}; y32 and / or y16 [B][H][W][Cout].  Honours adap_conv2d_next_gn_partial (Cout = 128, H * W % 256 == 0). */
int adap_conv3x3_rgb(const float* x_hwc, long ldx, const void* w_packed, const float* bias, float* y32, void* y16, int B,
                     int H, int W, int Cout, void* stream);

/* One UNet ResBlock (openaimodel.py:259-279: GroupNorm+SiLU -> conv3x3 + emb -> GroupNorm+SiLU -> conv3x3 + skip) issued from
 * ONE call, forward and data gradient: the launches of adap_groupnorm_fwd/bwd and adap_conv2d_nhwc in the order and with the
 * arguments of the per-op sequence, so the numbers are bit-identical to it -- what changes is the host: one call instead of 5-6
 * and their wrappers (csrc/blocks.hip).  Frozen weights (no weight gradients).
 * fwd: x f32 [B,H,W,Cin], emb_out f32 [B,Cout], forward packs (sk* NULL = identity skip, Cin == Cout) -> a1 bf16 [..,Cin],
 * h1 bf16 [..,Cout], a2 bf16 [..,Cout], skip f32 [..,Cout] (with sk only), out f32 [..,Cout], stats f32 [4][B][32] = mean1,
 * rstd1, mean2, rstd2.  gn_ws / sk_ws: the largest adap_groupnorm_workspace_floats / adap_conv2d_workspace_floats of the
 * block's calls (sk_ws may be NULL when all are 0); gn_sync as for adap_groupnorm_fwd.
 * bwd: g = d out as f32 (g_dtype 0) or its bf16 copy (1), g32 = the f32 gradient (added to dx by an identity skip), data-gradient
 * packs -> gx f32 [..,Cin] and gx16 its bf16 copy; ga2 / gh1 bf16 [..,Cout] and ga1 bf16 [..,Cin] are scratch. */
int adap_resblock_fwd(const float* x, const float* emb_out, const float* g1w, const float* g1b, const float* g2w,
                      const float* g2b, const void* c1w, const float* c1b, const void* c2w, const float* c2b, const void* skw,
                      const float* skb, void* a1, void* h1, void* a2, float* skip, float* out, float* stats, float* gn_ws,
                      float* sk_ws, void* gn_sync, int B, int H, int W, int Cin, int Cout, void* stream);
int adap_resblock_bwd(const void* g, int g_dtype, const float* g32, const float* x, const void* h1, const float* stats,
                      const float* g1w, const float* g1b, const float* g2w, const float* g2b, const void* c1wb,
                      const void* c2wb, const void* skwb, void* ga2, void* gh1, void* ga1, float* gx, void* gx16,
                      float* gn_ws, float* sk_ws, void* gn_sync, int B, int H, int W, int Cin, int Cout, void* stream);

/* One SpatialTransformer block (attention.py:260-341: GroupNorm -> proj_in -> [LN -> self attention] -> [LN -> cross attention] ->
 * [LN -> GEGLU feed-forward] -> proj_out + x; CrossAttention.forward attention.py:147-257) issued from ONE call each way: the
 * launches of the per-op sequence in its order and with its arguments, so the numbers are bit-identical to it; the work off
 * the block's dependency chain -- the cross-attention K/V projection of the context tokens, the token-map capture, the token
 * maps' gradient prologue, the context gradients -- goes to `lane` (a second stream, or NULL = all on `stream`), forked and
 * joined with events of the library's own (the capture's join is the caller's: it is only read after the UNet's forward).
 * Frozen weights (no weight gradients), bf16 storage, fused GEGLU.
 *   cfg (host ints): [0] B, [1] H, [2] W, [3] C, [4] heads, [5] M context tokens, [6] Cctx, [7] ADAP_STB_* flags,
 *                    [8] G token groups (capture / token-map gradient), bwd only: [9] / [10] leading dims of gop / g32
 *   w / wb: host arrays of DEVICE pointers in ADAP_STW_* / ADAP_STWB_* order (bf16 packs of adap_pack_conv_weight, f32 biases and
 *           norm gains; with a split context KV2 / V2 are the to_k / to_v packs, otherwise KV2 is their concatenation)
 *   t: host array of DEVICE pointers in ADAP_STF_* (fwd) / ADAP_STG_* (bwd) order; rows = B * H * W:
 *     fwd in : X f32 [rows][C]; CTX_K (CTX_V) f32 [B][M][Cctx]; KEY_MASK u8 [B][N] or NULL; PERM int32 [B][N] + KEY_COUNT int32 [B]
 *              (ADAP_STB_COMPACT); TOK_W f32 [B][M][G] (ADAP_STB_CAPTURE)
 *     fwd out: KV2 bf16 [B][M][2C] (k | v); GN_STATS f32 [2][B][32]; TRES f32 [3][rows][C] = t0, t1, t2 (the residual stream after
 *              proj_in / attn1 / attn2); LN_STATS f32 [6][rows]; QKV1 bf16 [rows][3C]; OBUF bf16 [3][rows][C] = o1, q2, o2;
 *              LSE f32 [2][B][heads][N]; HH bf16 [rows][8C] (permuted GEGLU pre-activation); KV1C bf16 [rows][2C] (compaction);
 *              TOKMAP f32 [B][heads][N][G] (capture); OUT f32 [rows][C]
 *     scratch: SCRATCH16 bf16 7 * rows * C; GN_WS adap_groupnorm_workspace_floats(B, N, C) floats; SK_WS / SK_WS_LANE
 *              adap_stblock_workspace_floats(...) floats each (either may be NULL when that is 0); GN_SYNC as adap_groupnorm_fwd
 *     bwd in : GOP = d out as bf16 (ADAP_STB_G_BF16) or f32, G32 = the f32 gradient (added to dx); the forward's saved tensors;
 *              INV_PERM; D_TOKMAP f32 [B][heads][N][G] + TOK_W + TOK_PREP (adap_attention_tokmap_prep_workspace_floats floats
 *              of scratch) with ADAP_STB_TOKGRAD
 *     bwd out: GX f32 / GX16 bf16 [rows][C] (not with ADAP_STB_NO_GX); DKV2 bf16 [B][M][2C]; G_CK (G_CV) f32 [B][M][Cctx] with
 *              ADAP_STB_WANT_GK / _GV
 *     scratch: SCRATCH32 f32 2 * rows * C; SCRATCH16 bf16 17 * rows * C; ATTN_WS max of adap_attention_bwd_workspace_floats over
 *              the self (M = N) and cross attention; GN_WS, SK_WS, SK_WS_LANE, GN_SYNC as above */
#define ADAP_STB_SAME_CTX 1
#define ADAP_STB_COMPACT 2
#define ADAP_STB_CAPTURE 4
#define ADAP_STB_Q1_PRESCALED 8
#define ADAP_STB_TOKGRAD 16
#define ADAP_STB_WANT_GK 32
#define ADAP_STB_WANT_GV 64
#define ADAP_STB_G_BF16 128
#define ADAP_STB_NO_GX 256          /* bwd: the block's input needs no gradient -- stop after the cross attention (GX / GX16 may be NULL) */
#define ADAP_STB_KV_GIVEN 512       /* fwd: KV2 already holds the context's K | V projection (hoisted, batched over layers): not recomputed */
#define ADAP_STB_CAPTURE_DEFERRED 1024   /* fwd (with ADAP_STB_CAPTURE): the token maps are NOT made here -- the caller fills TOKMAP with
                                            adap_attention_tokmap_fwd_batched behind the UNet's forward */
#define ADAP_STB_TOKPREP_GIVEN 2048      /* bwd (with ADAP_STB_TOKGRAD): TOK_PREP already holds adap_attention_tokmap_prep's result
                                            (adap_attention_tokmap_prep_batched in front of the backward) */
enum { ADAP_STW_GN_G, ADAP_STW_GN_B, ADAP_STW_PIN_W, ADAP_STW_PIN_B, ADAP_STW_LN1_G, ADAP_STW_LN1_B, ADAP_STW_QKV, ADAP_STW_OUT1_W,
       ADAP_STW_OUT1_B, ADAP_STW_LN2_G, ADAP_STW_LN2_B, ADAP_STW_Q2, ADAP_STW_KV2, ADAP_STW_V2, ADAP_STW_OUT2_W, ADAP_STW_OUT2_B,
       ADAP_STW_LN3_G, ADAP_STW_LN3_B, ADAP_STW_FF1G_W, ADAP_STW_FF1G_B, ADAP_STW_FF2_W, ADAP_STW_FF2_B, ADAP_STW_POUT_W,
       ADAP_STW_POUT_B, ADAP_STW_COUNT };
enum { ADAP_STWB_GN_G, ADAP_STWB_GN_B, ADAP_STWB_PIN, ADAP_STWB_LN1_G, ADAP_STWB_QKV, ADAP_STWB_OUT1, ADAP_STWB_LN2_G, ADAP_STWB_Q2,
       ADAP_STWB_KV2, ADAP_STWB_V2, ADAP_STWB_OUT2, ADAP_STWB_LN3_G, ADAP_STWB_FF1G, ADAP_STWB_FF2, ADAP_STWB_POUT, ADAP_STWB_COUNT };
enum { ADAP_STF_X, ADAP_STF_CTX_K, ADAP_STF_CTX_V, ADAP_STF_KEY_MASK, ADAP_STF_PERM, ADAP_STF_KEY_COUNT, ADAP_STF_TOK_W, ADAP_STF_KV2,
       ADAP_STF_GN_STATS, ADAP_STF_TRES, ADAP_STF_LN_STATS, ADAP_STF_QKV1, ADAP_STF_OBUF, ADAP_STF_LSE, ADAP_STF_HH, ADAP_STF_KV1C,
       ADAP_STF_TOKMAP, ADAP_STF_OUT, ADAP_STF_SCRATCH16, ADAP_STF_GN_WS, ADAP_STF_SK_WS, ADAP_STF_SK_WS_LANE, ADAP_STF_GN_SYNC,
       ADAP_STF_COUNT };
enum { ADAP_STG_GOP, ADAP_STG_G32, ADAP_STG_X, ADAP_STG_GN_STATS, ADAP_STG_TRES, ADAP_STG_LN_STATS, ADAP_STG_QKV1, ADAP_STG_OBUF,
       ADAP_STG_LSE, ADAP_STG_HH, ADAP_STG_KV1C, ADAP_STG_KV2, ADAP_STG_KEY_MASK, ADAP_STG_INV_PERM, ADAP_STG_KEY_COUNT,
       ADAP_STG_D_TOKMAP, ADAP_STG_TOK_W, ADAP_STG_TOK_PREP, ADAP_STG_GX, ADAP_STG_GX16, ADAP_STG_DKV2, ADAP_STG_G_CK, ADAP_STG_G_CV,
       ADAP_STG_SCRATCH32, ADAP_STG_SCRATCH16, ADAP_STG_ATTN_WS, ADAP_STG_GN_WS, ADAP_STG_SK_WS, ADAP_STG_SK_WS_LANE, ADAP_STG_GN_SYNC,
       ADAP_STG_COUNT };
long adap_stblock_workspace_floats(int B, int N, int C, int Cctx, int M);
int adap_stblock_fwd(const int* cfg, const void* const* w, void* const* t, void* lane, void* stream);
int adap_stblock_bwd(const int* cfg, const void* const* wb, void* const* t, void* lane, void* stream);

/* FeedForward with its GEGLU fused into the two contractions (attention.py:32-59: proj -> chunk -> a * gelu(gate) -> Linear).
 * The 8C pre-activation h is stored in a PERMUTED channel order -- 16 value channels, then their 16 gate channels, then the next
 * 16 values ... (new row 32 k + j <- value channel 16 k + j, new row 32 k + 16 + j <- gate channel 16 k + j) -- and w_packed /
 * bias of adap_linear_geglu_fwd are ff.net.0.proj's packs with their rows in that order (so is the data-gradient pack that
 * consumes dh).  fwd: x16 bf16 [rows][Cin] -> h16 bf16 [rows][C8] (kept for the backward) and out16 = a * gelu(gate) bf16
 * [rows][C8 / 2].  bwd: g16 = d(ff.net.2 output) bf16 [rows][C], w_packed_bwd = ff.net.2's data-gradient pack [1][C4][C] ->
 * dh16 bf16 [rows][2 * C4] (permuted).  Same arithmetic as adap_geglu_fwd / _bwd around adap_conv2d_nhwc, bit for bit. */
int adap_linear_geglu_fwd(const void* x16, long ldx, const void* w_packed, const float* bias, void* h16, long ldh,
                          void* out16, long ldo, long rows, int Cin, int C8, void* stream);
int adap_linear_geglu_bwd(const void* g16, long ldg, const void* w_packed_bwd, const void* h16, long ldh, void* dh16,
                          long lddh, long rows, int C, int C4, void* stream);

/* OIHW f32 (checkpoint layout, ddpm.py:321-344) -> bf16 [KH*KW][rows][cols].
 * mode 0 (forward): rows >= O, cols >= I, out[t][o][i] = w[o][i][ky][kx] (zero padded).
 * mode 1 (data gradient): rows >= I, cols >= O, out[t][i][o] = w[o][i][KH-1-ky][KW-1-kx]. */
int adap_pack_conv_weight(const float* w_oihw, void* out_bf16, int O, int I, int KH, int KW, int mode,
                          int rows, int cols, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm(32) [+ SiLU]: GroupNorm32 util.py:217-219 (+ nn.SiLU), Normalize attention.py:71-72 and
 * model.py:39-40 (+ nonlinearity model.py:34-36).  x [B][HW][C] f32 (residual stream) or bf16 (block-internal
 * tensors), C % 32 == 0; statistics in fp32/fp64; mean/rstd [B][32] are saved for
 * the backward; workspace holds adap_groupnorm_workspace_floats(B,HW,C) floats.  act: 0 none, 1 SiLU.
 *
 * sync: NULL, or a device buffer of adap_groupnorm_sync_ints() int32 that the caller zero-initialised ONCE and then
 * hands to every call issued on the same stream (one buffer per stream).  With it, tensors whose per-workgroup slab
 * fits the register file (every UNet shape at the training batch sizes) take a SINGLE launch that reads x once: the
 * workgroups of a sample exchange their group partials through tagged 8-byte records in `sync` between the statistics
 * and the normalisation (norms.hip).  sync[1] is a poison word, set if a bounded wait ever gave up (never, unless
 * more such kernels are in flight than the chip can hold: hand `sync` to ONE stream per device only);
 * adap_groupnorm_last_variant(): 0 = two launches, N = single launch with N pixel rows per thread.
 */
long adap_groupnorm_workspace_floats(int B, int HW, int C);
long adap_groupnorm_sync_ints(void);
int adap_groupnorm_last_variant(void);
int adap_groupnorm_fwd(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta,
                       float* y32, long ldy32, void* y16, long ldy16,
                       float* mean, float* rstd, float* workspace, void* sync,
                       int B, int HW, int C, float eps, int act, void* stream);
/* The same normalisation from statistics records a contraction's epilogue left (adap_conv2d_next_gn_partial): one small
 * launch finishes the stats_chunks records per sample into mean / rstd (fp64, fixed order), then the apply pass of the
 * two-launch form runs; no statistics pass over x. */
int adap_groupnorm_fwd_stats(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta,
                             float* y32, long ldy32, void* y16, long ldy16, float* mean, float* rstd,
                             const float* partial, int stats_chunks, int B, int HW, int C, float eps, int act, void* stream);
/* dx (f32 and/or bf16) from dy (f32 or bf16).  accumulate: dx32 = dx + add_src (the residual-stream gradient
 * coming down the skip path); add_src NULL = dx32 itself (in place), otherwise any f32 tensor of the same shape,
 * so the block's incoming gradient need not be cloned first.  dx16 is the bf16 copy of what dx32 receives. */
int adap_groupnorm_bwd(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                       const float* gamma, const float* beta, const float* mean, const float* rstd,
                       float* dx32, long lddx32, int accumulate, void* dx16, long lddx16,
                       const float* add_src, long ldadd,
                       float* workspace, void* sync, int B, int HW, int C, int act, void* stream);

/* LayerNorm over the last dim: BasicTransformerBlock.norm1/2/3, attention.py:267-269,275-285.
 * bwd: dx (f32, optionally accumulated into) and optionally dx16, a bf16 copy of the final dx. */
int adap_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y16, long ldy,
                       float* mean, float* rstd, long rows, int D, float eps, void* stream);
int adap_layernorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                       const float* mean, const float* rstd, float* dx, long lddx, int accumulate,
                       void* dx16, long lddx16, long rows, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CrossAttention core, attention.py:195-243: softmax((q k^T) * scale [+ key mask]) v, fused.
 * q [B][N][H*d], k/v [B][M][H*d] bf16; key_mask [B][M] bytes (0 = masked with -finfo.max, :223-232)
 * or NULL; out bf16 [B][N][H*d]; lse f32 [B][H][N].  d % 8 == 0, d <= 160.
 * key_count [B] int32 (device) or NULL: sample b attends to its FIRST key_count[b] keys only -- the form a key mask takes after
 * the kept keys have been compacted to the front (adap_gather_rows_bf16): masked keys contribute exactly 0 to the softmax
 * (attention.py:223-232 fills them with -finfo.max), so leaving them out is the same arithmetic on fewer tiles.  The
 * backward writes zeros into dk / dv rows >= key_count[b].  key_count[b] must be >= 1 (a softmax over no key has no
 * denominator); values < 1 are treated as 1 and values > M as M.  A sample whose mask keeps NO key is the caller's case to
 * resolve before compaction: the reference's masked_fill form then averages V uniformly, and KeyMasks.compaction (the host
 * side) keeps every key for such a sample, which is the same result.
 * scale = 0 means "q is PRE-SCALED": it already carries d^-1/2 * log2(e) (folded into to_q's weight pack before its one bf16
 * rounding), so the scores leave the matrix core in the exp2 domain and need no multiply.  adap_attention_bwd with scale = 0
 * returns dq with respect to that pre-scaled q (= ln 2 * dS K) and dk = ln 2 * dS^T q: pushed through the same scaled pack they
 * give the unscaled projection's input gradient.  out / lse are the same numbers in both forms.
 */
int adap_attention_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                       const uint8_t* key_mask, const int* key_count, void* out, long ldo, float* lse,
                       int B, int H, int N, int M, int d, float scale, void* stream);
/* which forward kernel the last adap_attention_fwd dispatched to: 1 / 2 = query-stationary kernel with 1 / 2 query blocks per
 * wave, 3 = the ping-pong kernel (512-query workgroups, SIMD partners half a tile apart; N >= 512 keys, d <= 64), 4 / 6 = the
 * interleaved kernel (P V of tile t-1, softmax of t, Q K^T of t+1 in one instruction stream; d = 40-like heads with a spare
 * V column, M >= 256) with 4 / 8 waves per workgroup */
int adap_attention_fwd_last_variant(void);
/* Diagnostic (tools/attn_stamps.py): with a non-NULL device buffer of 6 * (2 * ceil(M/64) + 1) uint64, workgroup (0,0) of the
 * ping-pong forward stores shader-clock stamps (entry / work done / barrier passed) per phase and wave half.  NULL = off. */
int adap_attention_set_stamp_buffer(void* buf);
/* Kernel-selection switches for tests and tuning tools (defaults come from the environment, read once: ADAP_ATTN_PP /
 * ADAP_ATTN_FORCE_PP, ADAP_ATTN_PP_PRIO, ADAP_ATTN_QB1, ADAP_ATTN_DKV_QSPLIT); -1 leaves a switch unchanged.
 * pp_mode 0 = default choice (query-stationary forward), 1 = ping-pong forward where it applies, 2 = always, 3 / 4 = the
 * interleaved forward with 8 / 4 waves per workgroup where it applies, 5 = query-stationary always (also ADAP_ATTN_FWD_MODE);
 * pp_prio: the ping-pong kernel's
 * raised-priority phase (1 matrix, 2 vector, 0 neither); qb1 = 1: one query block per wave; dkv_qsplit: 0 = heuristic. */
int adap_attention_set_debug(int pp_mode, int pp_prio, int qb1, int dkv_qsplit);
/* dq/dk/dv as f32 and/or bf16; workspace: adap_attention_bwd_workspace_floats(...) floats of scratch (the row
 * dots delta = sum(dO * O) and, when few key blocks exist -- cross attention, M = 77 -- the f32 partials of the
 * query-split dK/dV pass, summed in a fixed order). */
long adap_attention_bwd_workspace_floats(int B, int H, int N, int M, int d);
int adap_attention_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                       const uint8_t* key_mask, const int* key_count, const void* out, long ldo, const void* dout,
                       long lddo, const float* lse, float* workspace,
                       float* dq32, void* dq16, long lddq, float* dk32, void* dk16, long lddk,
                       float* dv32, void* dv16, long lddv,
                       int B, int H, int N, int M, int d, float scale, void* stream);
/* side outputs of the distillation layers, attention.py:245-255: attnscore / attn [B][H][N][M] f32,
 * q_scaled = q * scale^0.5 [B][H][N][d] f32 (any of them may be NULL).  M <= 192. */
int adap_attention_capture(const void* q, long ldq, const void* k, long ldk, float* attnscore, float* attn,
                           float* q_scaled, const float* tok_w, float* tokmap, int G, int B, int H, int N, int M, int d,
                           float scale, void* stream);
/* Token maps (optional outputs of the call above: tok_w f32 [B][M][G] -> tokmap f32 [B][H][N][G] =
 * sum_m attnscore[b][h][n][m] * tok_w[b][m][g], G <= 4): all the cross-layer consistency loss reads of attnscore
 * (ddpm.py:4323-4339: mean over heads, sum over the subject / background tokens).  Their gradient is applied to the
 * layer's bf16 dq / dk without ever forming the dense [B][H][N][M] gradient.  workspace:
 * adap_attention_tokmap_bwd_workspace_floats(...) floats. */
/* The same gradient folded into the attention backward itself (no read-modify-write pass over dq / dk): adap_attention_tokmap_prep
 * forms kw = w^T K and gq = d_tokmap^T Q (workspace: adap_attention_tokmap_prep_workspace_floats floats; it may run on another
 * stream as soon as d_tokmap is known), adap_attention_bwd_tok = adap_attention_bwd + scale * d_tokmap . kw into dq and
 * scale * w . gq into dk inside its epilogues. */
long adap_attention_tokmap_prep_workspace_floats(int B, int H, int N, int d, int G);
/* The token maps of n <= 16 layers in ONE launch, and their gradient prologue (adap_attention_tokmap_prep) for n layers in THREE:
 * the distillation layers' maps are read by the losses only after the UNet's forward, and the prologue depends only on what the
 * forward saved and on the losses' gradients, so neither has to sit on a transformer block's dependency chain
 * (ldm/modules/diffusionmodules/openaimodel.py capture of the 12 distillation layers, ddpm.py:3246-3270).  Flat HOST arrays:
 * ptrs [n][6] = {q, k, tok_w, tokmap (fwd) | NULL, d_tokmap (prep) | NULL, workspace (prep) | NULL} (device pointers),
 * lds [n][2] = {ldq, ldk}, dims [n][6] = {B, H, N, M, d, G}, scales [n].  Per layer the arithmetic of the single-layer calls. */
int adap_attention_tokmap_fwd_batched(int n, const void* const* ptrs, const long* lds, const int* dims, const float* scales,
                                      void* stream);
int adap_attention_tokmap_prep_batched(int n, const void* const* ptrs, const long* lds, const int* dims, void* stream);
int adap_attention_tokmap_prep(const float* d_tokmap, const float* tok_w, const void* q, long ldq, const void* k, long ldk,
                               float* workspace, int B, int H, int N, int M, int d, int G, void* stream);
int adap_attention_bwd_tok(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                           const uint8_t* key_mask, const int* key_count, const void* out, long ldo, const void* dout,
                           long lddo, const float* lse, float* workspace,
                           float* dq32, void* dq16, long lddq, float* dk32, void* dk16, long lddk,
                           float* dv32, void* dv16, long lddv,
                           int B, int H, int N, int M, int d, float scale,
                           const float* d_tokmap, const float* tok_w, const float* tok_prep, int G, void* stream);
long adap_attention_tokmap_bwd_workspace_floats(int B, int H, int N, int d, int G);
int adap_attention_tokmap_bwd(const float* d_tokmap, const float* tok_w, const void* q, long ldq, const void* k,
                              long ldk, void* dq16, long lddq, void* dk16, long lddk, float* workspace, int B, int H,
                              int N, int M, int d, int G, float scale, void* stream);
/* Gradient of those side outputs (ddpm.py:3246-3270: the recon iteration's cross-layer consistency loss reads
 * `attnscore` with gradient; stage 2 also `q`): dq16 / dk16 are the bf16 gradients adap_attention_bwd has written for
 * the same layer, and receive  += scale * dS k  (+ dim_head^-1/4 * dQs)  and  += scale * dS^T q  (f32 sum, rounded
 * once).  d_attnscore f32 [B][H][N][M] and d_q_scaled f32 [B][H][N][d], either may be NULL; workspace (needed with
 * d_attnscore): adap_attention_capture_bwd_workspace_floats(...) floats -- dk is reduced over 128-row chunks of the
 * queries in two stages, fixed order. */
long adap_attention_capture_bwd_workspace_floats(int B, int H, int N, int M, int d);
int adap_attention_capture_bwd(const float* d_attnscore, const float* d_q_scaled, const void* q, long ldq,
                               const void* k, long ldk, void* dq16, long lddq, void* dk16, long lddk,
                               float* workspace, int B, int H, int N, int M, int d, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GEGLU, attention.py:32-40: h = [a | gate] bf16 [rows][2*inner] -> a * gelu(gate) bf16 [rows][inner].
 */
int adap_geglu_fwd(const void* h, long ldh, void* out, long ldo, long rows, int inner, void* stream);
int adap_geglu_bwd(const void* dout, long lddo, const void* h, long ldh, void* dh, long lddh, long rows,
                   int inner, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MLP activation of the zero-shot front end's CLIP vision encoder (ddpm.py:904-914 loads HF `CLIPVisionModel`; its
 * `CLIPMLP` applies ACT2FN[config.hidden_act] between fc1 and fc2): x f32 [rows][cols] -> bf16 [rows][cols].
 * kind 0 = "quick_gelu" x * sigmoid(1.702 x) (openai/clip-vit-large-patch14), 1 = "gelu" (erf; the LAION checkpoints).
 */
int adap_act_fwd(const float* x, long ldx, void* out, long ldo, long rows, int cols, int kind, void* stream);

/* dst[b][i][0..cols) = src[b][idx[b][i]][0..cols): row gather of bf16 rows (cols % 8 == 0, 16-byte aligned rows) with leading
 * dimensions on both sides -- moves the keys a mask keeps to the front (and dk / dv back) for adap_attention_*'s key_count. */
int adap_gather_rows_bf16(const void* src, long lds, const int* idx, void* dst, long ldd, int B, int rows_src, int rows_dst,
                          int cols, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The first stage's encode as ONE call: image -> moments [-> z] (encode_first_stage / get_first_stage_encoding, ddpm.py:1381-1419,
 * 955-962; AutoencoderKL.encode autoencoder.py:324-328; Encoder.forward model.py:408-499).  Host code that issues the library's
 * own launches (conv / GroupNorm / the mid AttnBlock's two batched products and softmax / quant_conv / posterior sample) on
 * `stream` in Encoder.forward's order -- the same launches the Python mirror issues, hence the same bits.
 *   cfg (host ints): ch, num_levels, num_res_blocks, 2*z_channels, 2*embed_dim, ch_mult[num_levels]
 *   tensors (host array of DEVICE pointers; adap_vae_encode_tensor_count(cfg) of them), in this order:
 *     conv_in {w, b}; per level: per ResnetBlock {norm1 g, b; conv1 w, b; norm2 g, b; conv2 w, b; [nin_shortcut w, b when the
 *     block changes the channel count]}; {downsample w, b} for all but the last level; mid.block_1 (8); mid.attn_1 {norm g, b;
 *     q|k|v fused w, b; proj_out w, b}; mid.block_2 (8); norm_out {g, b}; conv_out {w, b}; quant_conv {w, b}
 *     -- every w is the bf16 forward pack of adap_pack_conv_weight ([taps][O4][I8]), every b / g f32.
 *   x_hwc [B][H][W][3] f32 in [-1,1] (the dataloader's layout); pixel_class [B][(H/8)*(W/8)] bytes (0 outside the aug mask,
 *   1 foreground, 2 background: the mid attention's hetero-pair zero fill, model.py:196-232) or NULL;
 *   moments f32 [B][h][w][2*embed_dim] out; z f32 [B][h][w][embed_dim] out or NULL, = scale * (mean + std * noise) with
 *   noise [B][h][w][embed_dim]; workspace: adap_vae_encode_workspace_bytes(cfg, B, H, W) bytes, 256-byte aligned;
 *   gn_sync as in adap_groupnorm_fwd (NULL = two-pass GroupNorm kernels). */
int adap_vae_encode_tensor_count(const int* cfg);
long adap_vae_encode_workspace_bytes(const int* cfg, int B, int H, int W);
int adap_vae_encode(const int* cfg, const void* const* tensors, int n_tensors, const float* x_hwc,
                    const uint8_t* pixel_class, const float* noise, float scale, float* moments, float* z,
                    void* workspace, long workspace_bytes, void* gn_sync, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The data-parallel gradient exchange as C entries (main.py:829 strategy="ddp": the mean of the trainable gradients after every
 * micro-batch backward).  RCCL (librccl.so.1, resolved with dlopen at the first call) on a communicator of the library's own:
 * rank 0 draws a unique id (adap_comm_unique_id_bytes() bytes), the host passes it to the other processes by any means, every
 * process calls adap_comm_init(&comm, id, nranks, rank) with its GPU current, then adap_allreduce_bucket(comm, buf, count,
 * dtype (0 = f32, 1 = bf16), average (0 = sum, 1 = mean), stream) in place on slices of its flat gradient buffer -- asynchronous
 * on `stream`, like every other entry.  (adaprompt_amd/parallel.py::GradReducer drives the same exchange through
 * torch.distributed by default and through these entries with backend="c_abi".) */
int adap_comm_unique_id_bytes(void);
int adap_comm_unique_id(void* id_out);
int adap_comm_init(void** comm_out, const void* id, int nranks, int rank);
int adap_allreduce_bucket(void* comm, void* buf, long count, int dtype, int average, void* stream);
int adap_comm_destroy(void* comm);

/* time_embed MLP openaimodel.py:518-522 and ResBlock emb_layers :217-223 (R <= 8 rows, exact f32):
 * y[r][n] = post( bias[n] + sum_k pre(x[r][k]) w[n][k] ), pre/post = SiLU when the flag is set. */
int adap_linear_small(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                      int R, int K, int N, int pre_silu, int post_silu, void* stream);

/* timestep_embedding, util.py:154-174 (t int64 [B] -> [B][dim]). */
int adap_timestep_embedding(const long long* t, float* out, int B, int dim, void* stream);

/* q_sample, ddpm.py:416-419 with extract_into_tensor util.py:99-102.  sqrt_ac / sqrt_1mac hold num_timesteps
 * entries; a t outside [0, num_timesteps) yields NaN for that sample instead of an out-of-bounds read. */
int adap_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ac,
                  const float* sqrt_1mac, float* out, int B, long per_sample, int num_timesteps, void* stream);

/* DiagonalGaussianDistribution.sample + scale_factor, distributions.py:24-37, ddpm.py:955-962.
 * moments pixel-major [pixels][2*zch] (mean | logvar), noise / z [pixels][zch]. */
int adap_posterior_sample(const float* moments, long ldm, const float* noise, float* z, long pixels, int zch,
                          float scale, void* stream);

/* calc_recon_loss, ddpm.py:3571-3595: masked, fg/bg-weighted MSE; writes loss[0] and (optionally) d loss / d out. */
int adap_masked_mse(const float* out, const float* tgt, const float* img_mask, const float* fg_mask, float w_fg,
                    float w_bg, long pixels, int C, float* loss, float* grad, void* stream);

/* torch.cat([h, hs.pop()], dim=1), openaimodel.py:1018, for pixel-major f32 tensors. */
int adap_concat2(const float* a, long lda, int Ca, const float* b, long ldb, int Cb, float* out, long ldo,
                 long rows, void* stream);

/* adjoint of F.interpolate(scale_factor=2, mode='nearest') (openaimodel.py:118): [B][2H][2W][C] -> [B][H][W][C]. */
int adap_sumpool2x2(const float* in, float* out, int B, int H, int W, int C, void* stream);

/* f32 [rows][Cin] -> bf16 [rows][Cout >= Cin], zero padded (image 3->8 / latent 4->8 channels, or a plain cast). */
int adap_pad_cast_bf16(const float* in, long ldi, int Cin, void* out, long ldo, int Cout, long rows, void* stream);
/* nearest x2 upsample + cast: f32 [B][H][W][C] (row stride ldi) -> bf16 [B][2H][2W][C] -- the interpolate in front of Upsample's
 * conv3x3 (openaimodel.py:95-123), so that the conv reads a plain bf16 image.  C % 8 == 0. */
int adap_upsample2x_bf16(const float* in, long ldi, void* out, int B, int H, int W, int C, void* stream);

/* batched bf16 transpose [R][C] -> [C][R] (V^T of the VAE mid attention, model.py:236-239). */
int adap_transpose_bf16(const void* in, void* out, int batch, int R, int C, void* stream);

/* The self-attention key masks of up to 8 UNet levels in one launch (reference: attention.py:223-232 and :332 -- img_mask
 * [B,1,h,w] resized to each level with F.interpolate(mode="nearest") and compared != 0; there once per transformer).
 * img f32 [B][h][w]; dims (host) int [2*nlev] = (H, W) per level; offs (host) long [4*nlev] = per level the byte offset of
 * its mask u8 [B][H*W] in out_u8, then the int32 offsets in out_i32 of perm [B][H*W], inv_perm [B][H*W] and count [B] --
 * the stable partition "kept keys first" that self-attention over the kept keys alone runs on (perm offset < 0: mask
 * only).  count = kept keys, or H*W for a sample that keeps none (it attends to every key). */
int adap_key_masks(const float* img, int B, int h, int w, int nlev, const int* dims, const long* offs, void* out_u8,
                   int* out_i32, void* stream);

/* Pixel classes of the VAE's masked mid-block attention (model.py:196-232): fg [B][h][w] and aug [B][ha][wa] (NULL = ones)
 * nearest-resized to H x W; out u8 [B][H*W] = 1 where fg*aug != 0, 2 where (1-fg)*aug != 0, else 0. */
int adap_pixel_classes(const float* fg, int h, int w, const float* aug, int ha, int wa, void* out, int B, int H, int W,
                       void* stream);

/* VAE mid AttnBlock softmax with the post-softmax hetero-pair zero fill, model.py:190-232.
 * pixel_class [batch][N] bytes: 0 outside aug mask, 1 fg, 2 bg; NULL = no masking. */
int adap_vae_softmax(const float* S, long lds, void* P, long ldp, const uint8_t* pixel_class, long rows, int N,
                     int rows_per_batch, float scale, void* stream);

/* y += a * x (f32; n % 4 == 0). */
int adap_axpy(const float* x, float* y, float a, long n, void* stream);
/* y32 [rows][C] = a + b (rows of a / b lda / ldb apart), optionally also as bf16 (y16, same packed layout): the
 * meeting point of the two gradients of a skip connection (openaimodel.py:1018), with the operand copy of the sum. */
int adap_add2(const float* a, long lda, const float* b, long ldb, float* y32, void* y16, long rows, int C,
              void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Optimiser step over flat fp32 buffers: Prodigy (ldm/prodigy.py:97-252) + the global gradient-norm clip that
 * precedes it (Lightning clip_gradients(0.5, "norm") -> torch clip_grad_norm_, ddpm.py:606-633).
 *
 * The D-adaptation state lives in DEVICE memory, so a step needs no device->host sync (the reference does two
 * .item() calls per parameter, prodigy.py:179,189):
 *   state[16] doubles: [0] d  [1] d_max  [2] d_numerator  [3] d_denom  [4] d_hat  [5] k  [6] clip coefficient
 *                      [7] gradient 2-norm  [8] 1.0 if the last step was skipped (d_denom == 0)  [9] d*lr*bias_corr
 *   workspace: adap_optim_workspace_doubles(nslots) doubles (fp64 partial sums; fixed-order combine = deterministic)
 * One step = adap_grad_clip_coef (optional; without it state[6] keeps its value, 1.0 after state_init)
 *          -> adap_prodigy_moments once per contiguous range of parameters whose group has lr > 0 (slot = 0..nslots-1)
 *          -> adap_prodigy_finish -> adap_prodigy_update over all parameters.
 * The clipped gradient is consumed on the fly and NOT written back (the reference zeroes it right after the step). */
long adap_optim_workspace_doubles(int nslots);
int adap_prodigy_state_init(double* state, double d0, void* stream);
/* state[7] = ||g||_2 (fp32, as torch), state[6] = min(1, max_norm / (||g|| + 1e-6)). g 16-byte aligned. */
int adap_grad_clip_coef(const float* g, long n, double max_norm, double* state, double* workspace, void* stream);
/* prodigy.py:160-192 for one range: exp_avg (m), exp_avg_sq (v), s updated in place from g * state[6]
 * (+ weight_decay_coupled * p when decouple=False); partial sums of <g, p0 - p> and |s| go to the slot. */
int adap_prodigy_moments(const float* p, const float* p0, const float* g, float* m, float* v, float* s, long n,
                         const double* state, double* workspace, int slot, double lr, double beta1, double beta2,
                         double beta3, double d0, double weight_decay_coupled, int use_bias_correction,
                         int safeguard_warmup, void* stream);
/* prodigy.py:194-229: d_numerator, d_denom, d_hat, d, d_max, k <- the nslots slots; a zero d_denom skips the step. */
int adap_prodigy_finish(double* state, const double* workspace, int nslots, double lr, double beta1, double beta2,
                        double beta3, double d0, double d_coef, double growth_rate, int use_bias_correction,
                        void* stream);
/* prodigy.py:231-248: p -= dlr * m / (sqrt(v) + d_new * eps), after p *= 1 - weight_decay_decoupled * dlr. */
int adap_prodigy_update(float* p, const float* m, const float* v, long n, const double* state, double eps,
                        double weight_decay_decoupled, void* stream);
/* The reference's other optimizer_type values, torch.optim.AdamW / torch.optim.NAdam (ddpm.py:5134-5142, 5188-5196), as one
 * pass over a parameter group's range of the flat buffers (state = the 16-double array above: only [6], the clip
 * coefficient, is read).  The step-dependent scalars depend on the step count alone and come from the host:
 *   g' = state[6] * g + weight_decay_coupled * p;  p *= 1 - decay;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 *   p -= (coef_grad * g' + coef_moment * m) / (sqrt(v * inv_bias_correction2) + eps)
 * AdamW: decay = lr * weight_decay, coef_grad = 0, coef_moment = lr / (1 - b1^t), inv_bias_correction2 = 1 / (1 - b2^t).
 * NAdam: coef_grad = lr (1 - mu_t) / (1 - prod_i mu_i), coef_moment = lr mu_{t+1} / (1 - mu_{t+1} prod_i mu_i). */
int adap_adam_update(float* p, const float* g, float* m, float* v, long n, const double* state, double beta1,
                     double beta2, double eps, double decay, double weight_decay_coupled, double inv_bias_correction2,
                     double coef_grad, double coef_moment, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Weight gradients (`unfreeze_model: True`, ddpm.py:775-786: the UNet's own parameters train; what torch autograd's
 * conv2d / linear / group_norm / layer_norm backward computes for nn.Conv2d openaimodel.py:229-247, nn.Linear
 * attention.py:157-165, GroupNorm32 util.py:217-219, nn.LayerNorm attention.py:267-269).
 *
 * adap_conv2d_bwd_weight: dw f32 [Cout][Cin][KH][KW] (OIHW; a Linear is KH = KW = 1 with Hout = rows, Wout = 1),
 * dbias f32 [Cout] or NULL, from x [B][Hin][Win][Cin] and dy [B][Hout][Wout][Cout] (f32 or bf16, row pitches ldx /
 * lddy); stride / pad / up as in adap_conv2d_nhwc (up 0 or 1).  accumulate != 0 adds to dw / dbias.  workspace: 256-byte
 * aligned device scratch of adap_conv2d_bwd_weight_workspace_bytes(...) bytes (transposed operands + split-K slabs). */
long adap_conv2d_bwd_weight_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW);
int adap_conv2d_bwd_weight(const void* x, int x_dtype, long ldx, const void* dy, int dy_dtype, long lddy,
                           float* dw, float* dbias, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                           int KH, int KW, int stride, int pad, int up, int accumulate, void* workspace,
                           long workspace_bytes, void* stream);
/* out[seg][c] (+)= sum over the seg_rows rows of segment seg of dy[row][c] (rows % seg_rows == 0): bias gradients
 * (one segment) and d emb_out of a ResBlock (one segment per image, openaimodel.py:264-268).  workspace:
 * adap_colsum_workspace_floats(rows, seg_rows, C) floats.  Two-stage, fixed order, fp64 finish. */
long adap_colsum_workspace_floats(long rows, long seg_rows, int C);
int adap_colsum(const void* dy, int dy_dtype, long lddy, long rows, long seg_rows, int C, float* out,
                int accumulate, float* workspace, void* stream);
/* dgamma[c] (+)= sum dz * xhat, dbeta[c] (+)= sum dz over all rows, dz = dy (act 0) or dy * silu'(xhat*gamma+beta)
 * (act 1).  kind 0: GroupNorm32, mean / rstd [B][32], rows = B*HW; kind 1: LayerNorm, mean / rstd [rows].
 * workspace: adap_colsum_workspace_floats(rows, rows, C) floats. */
int adap_norm_affine_bwd(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                         const float* gamma, const float* beta, const float* mean, const float* rstd, int kind,
                         int act, float* dgamma, float* dbeta, int accumulate, float* workspace, long rows, int HW,
                         int C, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Row-wise demeaned cosine loss against a sign-preserving power of the reference (ldm/util.py:437-535 calc_ref_cosine_loss,
 * exponent 1, 2 or 3; F.cosine_embedding_loss with margin 0): per row of x, r f32 [R][D]
 *   loss = 1 - cos(x - mean x, t)  (align) or max(0, cos) (align == 0),  t = (r - mean r) * |r - mean r|^(exponent - 1).
 * Forward: loss != NULL.  Backward: dx and / or dr != NULL receive gl[row] * d loss / d x (d r scaled by
 * ref_grad_scale, the reference's ScaleGrad). */
int adap_cosine_rows(const float* x, long ldx, const float* r, long ldr, const float* gl, float* loss, float* dx,
                     long lddx, float* dr, long lddr, long R, int D, int demean, int align, float ref_grad_scale,
                     int exponent, void* stream);
/* ortho_subtract (ldm/util.py:280): per row of a, b f32 [R][D] (rows contiguous, leading dims lda / ldb):
 * out = a - c b with c = <a,b> / (<b,b> + 1e-6).  g NULL: forward (out).  g = d L / d out [R][D]: backward, da and / or db
 * (NULL = not wanted):  da = g - s b,  db = -c g - s (a - 2 c b),  s = <g,b> / (<b,b> + 1e-6). */
int adap_ortho_rows(const float* a, long lda, const float* b, long ldb, const float* g, long ldg, float* out, long ldo,
                    float* da, long ldda, float* db, long lddb, long R, int D, void* stream);
/* The same with a workspace of adap_ortho_rows_workspace_floats(R, D) floats (0: none needed): rows of >= 8192 elements (Stage 2's
 * pooled feature maps: 4 rows of 72 000) are cut into 2048-element slices over two launches, the slices' sums added in a fixed
 * order; shorter rows, or workspace NULL: the one-workgroup-per-row form. */
long adap_ortho_rows_workspace_floats(long R, int D);
int adap_ortho_rows_ws(const float* a, long lda, const float* b, long ldb, const float* g, long ldg, float* out, long ldo,
                       float* da, long ldda, float* db, long lddb, long R, int D, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Stage-2 elastic matching loss of ONE distillation layer, value and gradient (ldm/util.py:2241-2368
 * calc_elastic_matching_loss; caller ddpm.py:4389-4551).  The batch is the four blocks (subject single, subject comp, mix
 * single, mix comp) of one instance:  q f32 [4][Cq][N] pooled queries, f f32 [4][Cf][N] pooled output features, fg f32 [N]
 * (non-zero = foreground token of the single instance).  All sums run in a fixed order (no split K, no atomics): two runs are
 * bit-equal.  Forward fills
 *   P2  f32 [2][N][N]  the two correspondence softmaxes P[z][i][j] (i single token, j comp token; z = 0 subject, 1 mix),
 *   RT  f32 [Cf][N]    the comp features carried onto the single tokens,
 *   tok f32 [ADAP_EM_TOK_ROWS][N]   per-token records (rows below; SC_BELOW / MC_BELOW are the two returned weight vectors),
 *   out f32 [ADAP_EM_OUT_FLOATS]    the three losses, the weight sum and the foreground count.
 * Backward takes device pointers to the incoming gradients of the three losses (scalars) and of the two `below` vectors
 * ([N]); NULL = zero.  Scratch: dS2 [2][N][N], dRT [Cf][N], coef [ADAP_EM_COEF_ROWS][N].  Results: dq [4][Cq][N], df [4][Cf][N]
 * with the reference's gradient scales applied (gs_q on the single queries, gs_feat on the single features, gs_mix on the
 * mix-comp features). */
enum { ADAP_EM_P_SC, ADAP_EM_P_MC, ADAP_EM_DOT_FG, ADAP_EM_NA_FG, ADAP_EM_NB_FG, ADAP_EM_DOT_BG, ADAP_EM_NA_BG, ADAP_EM_NB_BG,
       ADAP_EM_SC_BELOW, ADAP_EM_MC_BELOW, ADAP_EM_ROWPART, ADAP_EM_TOK_ROWS };
enum { ADAP_EM_OUT_MAP, ADAP_EM_OUT_FG, ADAP_EM_OUT_BG, ADAP_EM_OUT_W, ADAP_EM_OUT_NFG, ADAP_EM_OUT_FLOATS = 8 };
enum { ADAP_EM_COEF_ROWS = 10 };
int adap_elastic_match_fwd(const float* q, int Cq, const float* f, int Cf, const float* fg, int N, float cutoff, float* P2,
                           float* RT, float* tok, float* out, void* stream);
int adap_elastic_match_bwd(const float* q, int Cq, const float* f, int Cf, const float* fg, int N, float cutoff, float gs_q,
                           float gs_feat, float gs_mix, const float* P2, const float* RT, const float* tok, const float* out,
                           const float* g_map, const float* g_fg, const float* g_bg, const float* g_scb, const float* g_mcb,
                           float* dS2, float* dRT, float* coef, float* dq, float* df, void* stream);

/* calc_prompt_mix_loss's per-layer terms on the subject tokens' score maps (ddpm.py:3714-3930; calc_delta_alignment_loss
 * ldm/util.py:543-594 "feat_to_ref", cosine exponent 3; and the L1 between mean scores): a f32 [4][H][N] = the (subject single,
 * subject comp, mix single, mix comp) maps of one instance.  out != NULL: forward, out[0] = subj_attn_delta_align,
 * out[1] = subj_attn_norm_distill of this layer, rec f32 [H][ADAP_PM_REC] the record for the backward.  out == NULL: backward
 * for the two losses' incoming gradients (device scalars, NULL = 0) -> da [4][H][N], the mix maps' share scaled by gs_mix. */
enum { ADAP_PM_REC = 16 };
int adap_promptmix_attn_terms(const float* a, int H, int N, float gs_mix, float* rec, float* out, const float* g_delta,
                              const float* g_norm, float* da, void* stream);
/* convert_attn_to_spatial_weight (ldm/util.py:1718) for one instance at the feature map's own resolution: a0 (and a1, or NULL)
 * f32 [H][N] subject score maps -> sw f32 [N] = average over the sources of min(exp(-+(a - mean) / max(std + 0.001, mean / 2)), 1)
 * normalised to mean 1 (a = mean over heads; `reversed` != 0: small where the subject attends). */
int adap_attn_spatial_weight(const float* a0, const float* a1, int H, int N, int reversed, float* sw, void* stream);
/* The two background-suppression terms of calc_comp_fg_bg_preserve_loss (ddpm.py:4520-4545): a f32 [4][H][N] pooled subject
 * score maps, scb / mcb f32 [N] the elastic matching's `below` weights.  da == NULL: forward, out f32 [4] = (comp_subj_bg_attn_
 * suppress, comp_mix_bg_attn_suppress, the two mask counts), col f32 [2][N] kept for the backward.  da != NULL: backward for the
 * two incoming gradients (device scalars, NULL = 0) -> da [4][H][N] (mix share scaled by gs_mix), dscb, dmcb [N]. */
int adap_bg_suppress(const float* a, const float* scb, const float* mcb, int H, int N, float gs_mix, float* col, float* out,
                     const float* g_s, const float* g_m, float* da, float* dscb, float* dmcb, void* stream);

/* The four mask hinge terms of calc_fg_bg_complementary_loss (ddpm.py:4143-4238) for L same-resolution layers:
 * S / G f32 [L][B][H][N] = per-head score maps of the subject / background tokens (element stride `estride`: they
 * are columns of a token-map tensor; G NULL = subject-only, calc_fg_mb_suppress_loss), fmask f32 [B][N] in {0,1},
 * iw f32 [B] instance weights or NULL.  out f32 [4][L] = (subj_mb_suppress, bg_mf_suppress, subj_bg_contrast_at_mf,
 * bg_subj_contrast_at_mb) BEFORE the layer weights and the 0.05 / 0.1 scales.  The backward reads the forward's
 * workspace (adap_mask_hinges_workspace_floats(L, B) floats) and gout f32 [4][L]; it includes the 0.5 ScaleGrad on the
 * subject's foreground mean (ddpm.py:4093, 4163). */
long adap_mask_hinges_workspace_floats(int L, int B);
int adap_mask_hinges_fwd(const float* S, const float* G, long estride, const float* fmask, const float* iw, float* out,
                         float* workspace, int L, int B, int H, int N, float margin, float margin_bg_at_mf, void* stream);
int adap_mask_hinges_bwd(const float* S, const float* G, long estride, const float* fmask, const float* iw,
                         const float* gout, const float* workspace, float* dS, float* dG, long dstride, int L, int B,
                         int H, int N, float margin, float margin_bg_at_mf, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The recon iteration's attention regularisers, values and gradients in one call (csrc/regloss.hip):
 * calc_fg_bg_xlayer_consist_loss ddpm.py:4259-4387 and calc_fg_bg_complementary_loss ddpm.py:4043-4258 on the token maps
 * adap_attention_capture emits, with the weighting of ddpm.py:2921-2950 / 3246-3270 folded into six coefficients.
 *
 * tm / dtm: HOST arrays of L device pointers: token map of layer l, f32 [Bt][H][layer_N[l]][G] (G = 1 subject, 2 = + background),
 * and where its gradient goes (same shape; every element is written).  layer_N, complem_w (the layer's normalised
 * complementary-loss weight, 0 = takes no part), pair_x / pair_r / pair_w (cross-layer pair i compares the head-mean map of layer
 * pair_x[i] -- 2 x 2 mean-pooled when it has 4x the pixels -- with layer pair_r[i]'s as the reference, weight pair_w[i]) are HOST
 * arrays too.  fg_mask: f32 [Bt][Hm][Hm] in {0,1} at the latent resolution or NULL (no hinge terms); inst_w f32 [Bk] or NULL.
 * Only the first Bk instances count.  The total is
 *   cx_fg * L_fg + cx_bg * L_bg + cc_complem * L_complem + cc_smb * L_subj_mb_suppress + cc_bmf * L_bg_mf_suppress + cc_con * L_mask_contrast
 * parts (device, 8 floats) receives {L_fg, L_bg, L_complem, L_smb, L_bmf, L_con, total, 0}; dtm the gradient of the total.
 */
long adap_reg_losses_workspace_floats(const int* layer_N, int L, const int* pair_x, const int* pair_r, int npairs, int Bk,
                                      int H, int G);
int adap_reg_losses(const void* const* tm, void* const* dtm, const int* layer_N, const float* complem_w, int L,
                    const int* pair_x, const int* pair_r, const float* pair_w, int npairs,
                    const float* fg_mask, int Hm, const float* inst_w, int Bt, int Bk, int H, int G, int have_bg,
                    float margin, float margin_bg_at_mf, float fg_grad_scale,
                    float cx_fg, float cx_bg, float cc_complem, float cc_smb, float cc_bmf, float cc_con,
                    float* parts, float* workspace, long ws_floats, void* stream);

/* The static prompt-delta loss (ldm/util.py:2037 calc_prompt_emb_delta_loss -> ortho_subtract :280, calc_ref_cosine_loss :437),
 * value and gradient in one call.  emb4 f32 [4*Bs][L][T][D] contiguous: subject-single, subject-comp, class-single, class-comp
 * static embeddings (L layers, T tokens); mask4 f32 [4*Bs][T]: the prompt mask, whose start-token column is zeroed IN PLACE as
 * the reference does.  out2 (device) <- {loss, coef * loss}; demb4 (same shape as emb4) <- d (coef * loss) / d emb4, the class
 * delta's share scaled by cls_grad_scale (0.05).  workspace: adap_prompt_delta_loss_workspace_floats floats. */
long adap_prompt_delta_loss_workspace_floats(int Bs, int L, int T);
int adap_prompt_delta_loss(const float* emb4, float* demb4, float* mask4, int Bs, int L, int T, int D, float coef,
                           float cls_grad_scale, float* out2, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAPROMPT_HIP_H */

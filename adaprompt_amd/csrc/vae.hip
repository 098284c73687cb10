// adap_vae_encode: the first stage's encode as ONE C-ABI call (SURVEY.md 8b "minimum surface": vae_encode(x, masks, weights*,
// noise, z)) -- the launch sequence of Encoder.forward (model.py:408-499: conv_in, four levels of two ResnetBlocks with a
// stride-2 Downsample between them, mid block / AttnBlock / block, norm_out + SiLU, conv_out), quant_conv
// (autoencoder.py:302, 324-328) and the posterior sample scaled by scale_factor (distributions.py:24-37, ddpm.py:955-962),
// issued on the caller's stream from host code.  No kernels of its own: every launch goes through the library's other entry
// points (adap_conv2d_nhwc, adap_groupnorm_fwd, ...), in the order and with the arguments of the Python mirror
// (ldm/modules/diffusionmodules/model.py::Encoder.forward_nhwc), so the two produce the same bits.
#include "common.h"
#include <stdlib.h>

#include <string.h>

namespace {

struct Cursor {                     // walks the caller's table of device pointers in the documented order
    const void* const* t;
    int n, i;
    const void* next() { return i < n ? t[i++] : nullptr; }
};

struct ConvW { const void* w; const float* b; };
struct NormW { const float* g; const float* b; };

ConvW conv_w(Cursor& c) { ConvW r; r.w = c.next(); r.b = (const float*)c.next(); return r; }
NormW norm_w(Cursor& c) { NormW r; r.g = (const float*)c.next(); r.b = (const float*)c.next(); return r; }

inline long up8(long x) { return (x + 7) & ~7L; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Arena {                      // carves the caller's workspace; dry = size it only
    char* base; size_t off, cap; bool dry;
    void* take(size_t bytes) {
        size_t o = off;
        off = al256(off + bytes);
        return dry ? (void*)(uintptr_t)256 : (void*)(base + o);
    }
};

struct Ctx {
    int B; void* sync; hipStream_t s;
    float* gn_ws; float* mean; float* rstd; float* sk_ws;
    float* partA; float* partB;       // GroupNorm statistics records out of conv1's / conv2's epilogue (resblock)
};

// a conv3x3's output this large gets its GroupNorm statistics from the conv's own epilogue (the Python mirror's rule:
// model.py ResnetBlock.forward).  2^23 elements = every level of the SD-1.5 encoder at bs 4 (round 4: with two micro-batch lanes
// beside the prefetch stream a statistics pass costs ~48 us there, the records' finish ~16)
bool stats_from_epilogue(int B, int H, int W, int C) {
    static const bool off = [] { const char* e = getenv("ADAP_GN_EPILOGUE_STATS"); return e && atoi(e) == 0; }();   // A/B switch
    static const int lg = [] { const char* e = getenv("ADAP_GN_EPILOGUE_MIN_LOG2"); return e ? atoi(e) : 23; }();         // tuning
    return !off && (long)B * H * W * C >= (1L << lg) && (H * W) % 256 == 0 && C % 32 == 0;
}

// y = conv(x) [+ bias] [+ residual]; f32 and / or bf16 output (ops.conv2d)
int conv(const Ctx& c, const void* x, int x_bf16, int H, int W, int Cin, const ConvW& w, int Cout, int K, int stride, int pad,
         int Ho, int Wo, const float* residual, float* y32, void* y16) {
    return adap_conv2d_nhwc(x, x_bf16 ? 1 : 0, Cin, w.w, w.b, nullptr, 0, residual, residual ? Cout : 0, y32, y32 ? Cout : 0, y16,
                            y16 ? Cout : 0, c.B, H, W, Cin, Ho, Wo, Cout, K, K, stride, pad, 0, 1.0f, 0, c.sk_ws, 1, 0, 0, 0, 0,
                            (void*)c.s);
}

// conv3x3 stride 1 whose output feeds a GroupNorm: asks for the statistics records; *chunks = records per image (0: none)
int conv_stats_any(const Ctx& c, const void* x, int x_bf16, int H, int W, int Cin, const ConvW& w, int Cout, int stride, int pad,
                   int Ho, int Wo, const float* residual, float* y32, void* y16, float* part, int* chunks) {
    *chunks = 0;
    const bool want = stats_from_epilogue(c.B, Ho, Wo, Cout);
    int rc;
    if (want && (rc = adap_conv2d_next_gn_partial(part, Cout / 32))) return rc;
    if ((rc = conv(c, x, x_bf16, H, W, Cin, w, Cout, 3, stride, pad, Ho, Wo, residual, y32, y16))) return rc;
    if (want) *chunks = adap_conv2d_last_gn_chunks();
    return ADAP_OK;
}

int conv_stats(const Ctx& c, const void* x, int H, int W, int Cin, const ConvW& w, int Cout, const float* residual, float* y32,
               void* y16, float* part, int* chunks) {
    return conv_stats_any(c, x, 1, H, W, Cin, w, Cout, 1, 1, H, W, residual, y32, y16, part, chunks);
}

int gn_stats(const Ctx& c, const void* x, int x_bf16, int HW, int C, const NormW& n, int act, void* y16, const float* part,
             int chunks) {
    return adap_groupnorm_fwd_stats(x, x_bf16 ? 1 : 0, C, n.g, n.b, nullptr, 0, y16, C, c.mean, c.rstd, part, chunks, c.B, HW, C,
                                    1e-6f, act, (void*)c.s);
}

int gn(const Ctx& c, const void* x, int x_bf16, int HW, int C, const NormW& n, int act, void* y16) {
    return adap_groupnorm_fwd(x, x_bf16 ? 1 : 0, C, n.g, n.b, nullptr, 0, y16, C, c.mean, c.rstd, c.gn_ws, c.sync, c.B, HW, C, 1e-6f,
                              act, (void*)c.s);
}

struct ResW { NormW n1; ConvW c1; NormW n2; ConvW c2; ConvW nin; bool has_nin; };

ResW res_w(Cursor& cur, int cin, int cout) {
    ResW r;
    r.n1 = norm_w(cur); r.c1 = conv_w(cur); r.n2 = norm_w(cur); r.c2 = conv_w(cur);
    r.has_nin = cin != cout;
    if (r.has_nin) r.nin = conv_w(cur); else { r.nin.w = nullptr; r.nin.b = nullptr; }
    return r;
}

// ResnetBlock.forward (model.py:108-142): x f32 [B,H,W,cin] -> y f32 [B,H,W,cout]; a16 / h16 bf16 scratch, skip f32 scratch
// *xc: statistics records per image that the producer of x left in c.partB (0: none); on return, those of y
// y16 (instead of y): the block's output goes to a Downsample's contraction alone -- written as its bf16 operand only
int resblock(const Ctx& c, const ResW& w, const float* x, int H, int W, int cin, int cout, void* a16, void* h16, float* skip,
             float* y, int* xc, void* y16 = nullptr) {
    int rc, hc = 0;
    if (*xc > 0) rc = gn_stats(c, x, 0, H * W, cin, w.n1, 1, a16, c.partB, *xc);
    else rc = gn(c, x, 0, H * W, cin, w.n1, 1, a16);
    if (rc) return rc;
    if ((rc = conv_stats(c, a16, H, W, cin, w.c1, cout, nullptr, nullptr, h16, c.partA, &hc))) return rc;   // block-internal: bf16
    if (hc > 0) rc = gn_stats(c, h16, 1, H * W, cout, w.n2, 1, a16, c.partA, hc);
    else rc = gn(c, h16, 1, H * W, cout, w.n2, 1, a16);
    if (rc) return rc;
    const float* sk = x;
    if (w.has_nin) {
        if ((rc = conv(c, x, 0, H, W, cin, w.nin, cout, 1, 1, 0, H, W, nullptr, skip, nullptr))) return rc;
        sk = skip;
    }
    if (y16) {
        *xc = 0;
        return conv(c, a16, 1, H, W, cout, w.c2, cout, 3, 1, 1, H, W, sk, nullptr, y16);
    }
    return conv_stats(c, a16, H, W, cout, w.c2, cout, sk, y, nullptr, c.partB, xc);
}

bool downsample_bf16() {
    static const bool on = [] { const char* e = getenv("ADAP_VAE_DOWNSAMPLE_BF16"); return !(e && atoi(e) == 0); }();   // A/B switch
    return on;
}

struct Plan {                       // what the configuration implies
    int ch, levels, nres, zc2, emb2;
    int mult[8];
};

int read_plan(const int* cfg, Plan& p) {
    p.ch = cfg[0]; p.levels = cfg[1]; p.nres = cfg[2]; p.zc2 = cfg[3]; p.emb2 = cfg[4];
    if (p.ch <= 0 || p.ch % 32 || p.levels < 1 || p.levels > 8 || p.nres < 1 || p.nres > 4 || p.zc2 % 4 || p.emb2 % 4) return 1;
    for (int i = 0; i < p.levels; ++i) { p.mult[i] = cfg[5 + i]; if (p.mult[i] < 1) return 1; }
    return 0;
}

int expected_tensors(const Plan& p) {
    int n = 2, cin = p.ch;
    for (int l = 0; l < p.levels; ++l) {
        int cout = p.ch * p.mult[l];
        for (int r = 0; r < p.nres; ++r) { n += 8 + (cin != cout ? 2 : 0); cin = cout; }
        if (l != p.levels - 1) n += 2;
    }
    return n + 8 + 2 + 2 + 2 + 8 + 2 + 2 + 2;      // mid.block_1, attn norm, qkv, proj_out, mid.block_2, norm_out, conv_out, quant_conv
}

// the whole sequence; with a dry arena it only sizes the workspace
int run(const Plan& p, Cursor cur, Arena& ar, const float* x_hwc, const uint8_t* pixel_class, const float* noise, float scale,
        float* moments, float* z, void* sync, int B, int H, int W, hipStream_t s) {
    const bool dry = ar.dry;
    int rc;
    // ---- sizes of the widest tensors
    long px0 = (long)B * H * W;
    int cmax = 0, cmid = p.ch * p.mult[p.levels - 1];
    long f32_max = 0, b16_max = 0, gn_ws = 0, sk_ws = 0;
    {
        int h = H, w = W, cin = p.ch;
        auto see = [&](int hh, int ww, int ci, int co, int k) {
            long px = (long)B * hh * ww;
            if (px * co > f32_max) f32_max = px * co;
            if (px * (ci > co ? ci : co) > b16_max) b16_max = px * (ci > co ? ci : co);
            long g1 = ci % 32 == 0 ? adap_groupnorm_workspace_floats(B, hh * ww, ci) : 0,
                 g2 = co % 32 == 0 ? adap_groupnorm_workspace_floats(B, hh * ww, co) : 0;
            if (g1 > gn_ws) gn_ws = g1;
            if (g2 > gn_ws) gn_ws = g2;
            long s1 = adap_conv2d_workspace_floats(B, hh, ww, ci, co, k, k);
            if (s1 > sk_ws) sk_ws = s1;
            if (co > cmax) cmax = co;
        };
        see(h, w, 8, p.ch, 3);
        for (int l = 0; l < p.levels; ++l) {
            int cout = p.ch * p.mult[l];
            for (int r = 0; r < p.nres; ++r) {
                see(h, w, cin, cout, 3);
                see(h, w, cout, cout, 3);
                if (cin != cout) see(h, w, cin, cout, 1);
                cin = cout;
            }
            if (l != p.levels - 1) { h /= 2; w /= 2; see(h, w, cin, cin, 3); }
        }
        see(h, w, cmid, 3 * cmid, 1);
        see(h, w, cmid, p.zc2, 3);
        // the 1x1 contractions issued as [1, B*N, 1, C] rows
        const int rows = B * h * w;
        long q1 = adap_conv2d_workspace_floats(1, rows, 1, cmid, 3 * cmid, 1, 1), q2 = adap_conv2d_workspace_floats(1, rows, 1, cmid, cmid, 1, 1),
             q3 = adap_conv2d_workspace_floats(1, rows, 1, p.zc2, p.emb2, 1, 1);
        if (q1 > sk_ws) sk_ws = q1;
        if (q2 > sk_ws) sk_ws = q2;
        if (q3 > sk_ws) sk_ws = q3;
    }
    Ctx c;
    c.B = B; c.sync = sync; c.s = s;
    void* x16 = ar.take((size_t)px0 * 8 * 2);
    float* hA = (float*)ar.take((size_t)f32_max * 4);
    float* hB = (float*)ar.take((size_t)f32_max * 4);
    float* hS = (float*)ar.take((size_t)f32_max * 4);
    void* a16 = ar.take((size_t)b16_max * 2);
    void* h16 = ar.take((size_t)b16_max * 2);
    c.gn_ws = (float*)ar.take((size_t)(gn_ws > 0 ? gn_ws : 1) * 4);
    c.sk_ws = (float*)ar.take((size_t)(sk_ws > 0 ? sk_ws : 1) * 4);
    c.partA = (float*)ar.take(((size_t)px0 / 64 + 1) * 64 * 4);           // [B][H * W / 64][32][2]: the full-resolution level
    c.partB = (float*)ar.take(((size_t)px0 / 64 + 1) * 64 * 4);
    c.mean = (float*)ar.take((size_t)B * 32 * 4);
    c.rstd = (float*)ar.take((size_t)B * 32 * 4);
    const int hl = H >> (p.levels - 1), wl = W >> (p.levels - 1);
    const long N = (long)hl * wl;
    void* qkv = ar.take((size_t)B * N * 3 * cmid * 2);
    void* qc = ar.take((size_t)B * N * cmid * 2);
    void* kc = ar.take((size_t)B * N * cmid * 2);
    void* vc = ar.take((size_t)B * N * cmid * 2);
    void* vT = ar.take((size_t)B * N * cmid * 2);
    float* S = (float*)ar.take((size_t)B * N * N * 4);
    void* P = ar.take((size_t)B * N * N * 2);
    void* o16 = ar.take((size_t)B * N * cmid * 2);
    float* hout = (float*)ar.take((size_t)B * N * up8(p.zc2) * 4);
    if (dry) return ADAP_OK;
    if (ar.off > ar.cap) return adap_set_error(ADAP_ERR_SHAPE, "vae_encode: workspace of %zu bytes needed, %zu given", ar.off, ar.cap);

    // ---- conv_in on the 3 -> 8 channel padded bf16 image (model.py:426, 468)
    ConvW cin_w = conv_w(cur);
    int xc = 0;                    // statistics records per image the producer of h left in c.partB (0: none)
    if (p.ch == 32 || p.ch == 64 || p.ch == 128) {
        // the 27-column single-K-step kernel straight from the f32 image (conv_gemm.hip conv3x3_rgb_kernel; the Python mirror's
        // rule: model.py Encoder.forward_nhwc)
        const bool want = stats_from_epilogue(B, H, W, p.ch);
        if (want && (rc = adap_conv2d_next_gn_partial(c.partB, p.ch / 32))) return rc;
        if ((rc = adap_conv3x3_rgb(x_hwc, 3, cin_w.w, cin_w.b, hA, nullptr, B, H, W, p.ch, (void*)s))) return rc;
        if (want) xc = adap_conv2d_last_gn_chunks();
    } else {
        if ((rc = adap_pad_cast_bf16(x_hwc, 3, 3, x16, 8, 8, px0, (void*)s))) return rc;
        if ((rc = conv_stats_any(c, x16, 1, H, W, 8, cin_w, p.ch, 1, 1, H, W, nullptr, hA, nullptr, c.partB, &xc))) return rc;
    }
    float* h = hA;
    float* other = hB;
    int hh = H, ww = W, cin = p.ch;
    for (int l = 0; l < p.levels; ++l) {
        int cout = p.ch * p.mult[l];
        // a level's last block feeds the Downsample alone: handed over as that contraction's bf16 operand (the same rounding the
        // contraction applies to an f32 input in registers; half the bytes written there and read here), in the f32 buffer's memory
        const bool to_down = l != p.levels - 1 && downsample_bf16();
        for (int r = 0; r < p.nres; ++r) {
            ResW w = res_w(cur, cin, cout);
            const bool hand16 = to_down && r == p.nres - 1;
            if ((rc = resblock(c, w, h, hh, ww, cin, cout, a16, h16, hS, other, &xc, hand16 ? (void*)other : nullptr))) return rc;
            float* t = h; h = other; other = t;
            cin = cout;
        }
        if (l != p.levels - 1) {       // Downsample: F.pad(0,1,0,1) + conv3x3 stride 2 pad 0 (model.py:151-178)
            ConvW dw = conv_w(cur);
            if ((rc = conv_stats_any(c, h, to_down ? 1 : 0, hh, ww, cin, dw, cin, 2, 0, hh / 2, ww / 2, nullptr, other, nullptr, c.partB,
                                     &xc)))
                return rc;
            float* t = h; h = other; other = t;
            hh /= 2; ww /= 2;
        }
    }
    // ---- mid: block_1, AttnBlock (model.py:179-242), block_2
    {
        ResW w = res_w(cur, cin, cin);
        if ((rc = resblock(c, w, h, hh, ww, cin, cin, a16, h16, hS, other, &xc))) return rc;
        float* t = h; h = other; other = t;
    }
    {
        const int C = cin;
        NormW an = norm_w(cur);
        ConvW wqkv = conv_w(cur), wproj = conv_w(cur);
        if (xc > 0) rc = gn_stats(c, h, 0, (int)N, C, an, 0, a16, c.partB, xc);
        else rc = gn(c, h, 0, (int)N, C, an, 0, a16);
        if (rc) return rc;
        xc = 0;                    // block_2's input comes out of proj_out: no records
        // (the 1x1 contractions see [1, B*N, 1, C])
        if ((rc = adap_conv2d_nhwc(a16, 1, C, wqkv.w, wqkv.b, nullptr, 0, nullptr, 0, nullptr, 0, qkv, 3 * C, 1, (int)(B * N), 1, C,
                                   (int)(B * N), 1, 3 * C, 1, 1, 1, 0, 0, 1.0f, 0, c.sk_ws, 1, 0, 0, 0, 0, (void*)s))) return rc;
        const size_t row = (size_t)C * 2, pitch = (size_t)3 * C * 2;
        if (hipMemcpy2DAsync(qc, row, qkv, pitch, row, (size_t)(B * N), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpy2DAsync(kc, row, (const char*)qkv + row, pitch, row, (size_t)(B * N), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpy2DAsync(vc, row, (const char*)qkv + 2 * row, pitch, row, (size_t)(B * N), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return adap_set_error(ADAP_ERR_HIP, "vae_encode: hipMemcpy2DAsync failed");
        if ((rc = adap_transpose_bf16(vc, vT, B, (int)N, C, (void*)s))) return rc;
        // S = q k^T (unscaled, f32), P = softmax(S * C^-1/2) with the fg / bg hetero-pair zero fill, o = P v
        if ((rc = adap_conv2d_nhwc(qc, 1, C, kc, nullptr, nullptr, 0, nullptr, 0, S, N, nullptr, 0, 1, (int)N, 1, C, (int)N, 1, (int)N, 1, 1,
                                   1, 0, 0, 1.0f, 1, nullptr, B, N * C, N * C, N * N, N * N, (void*)s))) return rc;
        if ((rc = adap_vae_softmax(S, N, P, N, pixel_class, (long)B * N, (int)N, (int)N, 1.0f / sqrtf((float)C), (void*)s))) return rc;
        if ((rc = adap_conv2d_nhwc(P, 1, N, vT, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, o16, C, 1, (int)N, 1, (int)N, (int)N, 1, C, 1, 1,
                                   1, 0, 0, 1.0f, 1, nullptr, B, N * N, (long)C * N, N * C, N * C, (void*)s))) return rc;
        if ((rc = adap_conv2d_nhwc(o16, 1, C, wproj.w, wproj.b, nullptr, 0, h, C, other, C, nullptr, 0, 1, (int)(B * N), 1, C, (int)(B * N),
                                   1, C, 1, 1, 1, 0, 0, 1.0f, 0, c.sk_ws, 1, 0, 0, 0, 0, (void*)s))) return rc;
        float* t = h; h = other; other = t;
    }
    {
        ResW w = res_w(cur, cin, cin);
        if ((rc = resblock(c, w, h, hh, ww, cin, cin, a16, h16, hS, other, &xc))) return rc;
        float* t = h; h = other; other = t;
    }
    // ---- norm_out + SiLU, conv_out, quant_conv (1x1), posterior sample
    NormW no = norm_w(cur);
    ConvW co = conv_w(cur), qcw = conv_w(cur);
    if (xc > 0) rc = gn_stats(c, h, 0, (int)N, cin, no, 1, a16, c.partB, xc);
    else rc = gn(c, h, 0, (int)N, cin, no, 1, a16);
    if (rc) return rc;
    if ((rc = conv(c, a16, 1, hh, ww, cin, co, p.zc2, 3, 1, 1, hh, ww, nullptr, hout, nullptr))) return rc;
    if ((rc = adap_conv2d_nhwc(hout, 0, p.zc2, qcw.w, qcw.b, nullptr, 0, nullptr, 0, moments, p.emb2, nullptr, 0, 1, (int)(B * N), 1, p.zc2,
                               (int)(B * N), 1, p.emb2, 1, 1, 1, 0, 0, 1.0f, 0, c.sk_ws, 1, 0, 0, 0, 0, (void*)s))) return rc;
    if (z) {
        if (!noise) return adap_set_error(ADAP_ERR_SHAPE, "vae_encode: z wanted but no noise given");
        if ((rc = adap_posterior_sample(moments, p.emb2, noise, z, (long)B * N, p.emb2 / 2, scale, (void*)s))) return rc;
    }
    return ADAP_OK;
}

}  // namespace

extern "C" int adap_vae_encode_tensor_count(const int* cfg) {
    Plan p;
    if (read_plan(cfg, p)) return -1;
    return expected_tensors(p);
}

extern "C" long adap_vae_encode_workspace_bytes(const int* cfg, int B, int H, int W) {
    Plan p;
    if (read_plan(cfg, p)) return -1;
    Arena ar = {nullptr, 0, 0, true};
    Cursor cur = {nullptr, 0, 0};
    run(p, cur, ar, nullptr, nullptr, nullptr, 1.0f, nullptr, nullptr, nullptr, B, H, W, nullptr);
    return (long)ar.off;
}

extern "C" int adap_vae_encode(const int* cfg, const void* const* tensors, int n_tensors, const float* x_hwc,
                               const uint8_t* pixel_class, const float* noise, float scale, float* moments, float* z,
                               void* workspace, long workspace_bytes, void* gn_sync, int B, int H, int W, void* stream) {
    Plan p;
    ADAP_REQUIRE(cfg && !read_plan(cfg, p), ADAP_ERR_SHAPE, "vae_encode: bad configuration");
    ADAP_REQUIRE(tensors && n_tensors == expected_tensors(p), ADAP_ERR_SHAPE, "vae_encode: %d tensors expected", expected_tensors(p));
    for (int i = 0; i < n_tensors; ++i) ADAP_REQUIRE(tensors[i], ADAP_ERR_SHAPE, "vae_encode: tensor %d is null", i);
    ADAP_REQUIRE(x_hwc && moments && workspace, ADAP_ERR_SHAPE, "vae_encode: null pointer");
    ADAP_REQUIRE(B > 0 && H > 0 && W > 0 && H % (1 << (p.levels - 1)) == 0 && W % (1 << (p.levels - 1)) == 0, ADAP_ERR_SHAPE,
                 "vae_encode: image %dx%d not divisible by the encoder's stride", H, W);
    ADAP_REQUIRE(((uintptr_t)workspace % 256) == 0, ADAP_ERR_ALIGN, "vae_encode: workspace must be 256-byte aligned");
    Arena ar = {(char*)workspace, 0, (size_t)workspace_bytes, false};
    Cursor cur = {tensors, n_tensors, 0};
    return run(p, cur, ar, x_hwc, pixel_class, noise, scale, moments, z, gn_sync, B, H, W, (hipStream_t)stream);
}

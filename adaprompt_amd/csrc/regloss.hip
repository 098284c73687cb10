// The recon iteration's attention regularisers on the captured token maps, values AND gradients, as ONE call of eight
// launches (the host-side torch expression of the same arithmetic costs ~400 element-wise launches per micro-batch once
// autograd has walked it backwards: ldm/models/diffusion/ddpm.py calc_fg_bg_xlayer_consist_loss / calc_fg_bg_complementary_loss
// are the readable form, and the checker of this file in tests/):
//
//   * cross-layer consistency (reference ddpm.py:4259-4387): per aligned layer pair the head-mean token map of the finer
//     layer, bilinearly resized to the coarser one (an exact 2 x 2 mean for the UNet's factor-2 pairs), against the coarser
//     layer's map: demeaned cosine with a sign-preserving squared reference (util.py:437-535), mean over the instances,
//     weighted per pair;
//   * fg / bg complementary loss (ddpm.py:4043-4258): per layer and head, cosine_embedding(bg map, subj map * |subj map|, -1)
//     with the reference's gradient scaled by fg_grad_scale, and the four mask hinge terms with their masked means (the
//     arithmetic of misc.hip's hinge kernels, here over layers of different resolutions in one grid), a resolution whose
//     resized mask has an instance without foreground or without background contributing nothing (a 0/1 device factor).
//
// The total is linear in these terms with coefficients the host knows at call time (the loss weights of the yaml), so the
// gradient with respect to every token map is produced in the same call (as adap_masked_mse does for the MSE): nothing is
// saved for a backward pass and autograd never sees the small tensors.
//
// Token map of layer l: f32 [Bt][H][N_l][G] (adap_attention_capture's side output), G = 1 (subject) or 2 (+ background).
// Every element of every gradient tensor is written by exactly one thread, sums run in a fixed order: bit-reproducible.
#include "common.h"
#include <string.h>

#define REG_MAX_LAYERS 16
#define REG_MAX_PAIRS 16
#define REG_MAX_RES 4

struct RegParams {
    const float* tm[REG_MAX_LAYERS];
    float* dtm[REG_MAX_LAYERS];
    int N[REG_MAX_LAYERS];           // pixels of layer l (side^2)
    int side[REG_MAX_LAYERS];
    int res[REG_MAX_LAYERS];         // index of the layer's resolution in the mask tables
    float lw[REG_MAX_LAYERS];        // complementary-loss weight of the layer (0: not part of it)
    int crow0[REG_MAX_LAYERS];       // first complementary-cosine row of layer l (rows: (b, h))
    long crowoff[REG_MAX_LAYERS];    // float offset of that row's gradient storage
    int elem0[REG_MAX_LAYERS + 1];   // prefix of Bt*H*N_l: the scatter kernel's grid decode, one thread per (l, b, h, n)
    int L;
    // cross-layer pairs: x = source `px` (pooled 2 x 2 when ppool), reference = source `pr`
    int px[REG_MAX_PAIRS], pr[REG_MAX_PAIRS], ppool[REG_MAX_PAIRS], pN[REG_MAX_PAIRS];
    long prowoff[REG_MAX_PAIRS];     // float offset of the pair's first gradient row (rows: (g, b))
    float pw[REG_MAX_PAIRS];
    int npairs;
    // resolutions
    int rside[REG_MAX_RES];
    long rmaskoff[REG_MAX_RES];      // float offset of fgm[res] [Bk][side^2] in the workspace
    int nres;
    int Bt, Bk, H, G;
    int have_bg, have_mask;
    const float* fg_mask;            // [Bt][Hm][Wm] (latent resolution), or NULL
    int Hm;
    const float* iw;                 // [Bk] instance weights or NULL
    float m, m3, fg_grad_scale;
    float cx[2];                     // d total / d (fg, bg) cross-layer loss
    float cc, c_smb, c_bmf, c_con;   // d total / d (complementary cosine, subj_mb_suppress, bg_mf_suppress, mask_contrast)
    // workspace pieces
    float* fgm;                      // masks
    float* valid;                    // [nres]
    float* nfg;                      // [nres][Bk] foreground pixels of the resized mask
    float* xloss;                    // [npairs][G][Bk]
    float* closs;                    // [crows]
    float* xdx; float* xdr;          // cross-layer gradient rows
    float* cdx; float* cdr;          // complementary-cosine gradient rows
    float* havg;                     // [L*Bk][2]
    float* hpart;                    // [L*Bk][8]
    float* hout;                     // [4][L]
    float* hcnt;                     // [4][L]
    float* parts;                    // [8] out
    int ncrows;
};

__device__ __forceinline__ float reg_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- K0: the foreground mask at every attention resolution: max(nearest, bilinear) > 1e-6 (util.py:1570
// resize_mask_for_feat_or_attn, mode "nearest|bilinear"); grid (nres, Bk).  For the integer factors f = Hm / side of the
// UNet (1, 2, 4, 8) the bilinear sample (align_corners = False) sits at f*d + (f-1)/2: the mean of the central 2 x 2.
__global__ __launch_bounds__(256) void reg_mask_kernel(RegParams p) {
    const int r = blockIdx.x, b = blockIdx.y, s = p.rside[r], f = p.Hm / s;
    float* out = p.fgm + p.rmaskoff[r] + (long)b * s * s;
    const float* src = p.fg_mask + (long)b * p.Hm * p.Hm;
    __shared__ float red[4];
    float cnt = 0.f;
    for (int i = threadIdx.x; i < s * s; i += 256) {
        const int y = i / s, x = i - y * s;
        float v = src[(y * f) * p.Hm + x * f];
        if (f > 1) {
            const int c = f / 2 - 1;
            const float* q = src + (y * f + c) * p.Hm + x * f + c;
            v = fmaxf(v, 0.25f * ((q[0] + q[1]) + (q[p.Hm] + q[p.Hm + 1])));
        }
        const float m = v > 1e-6f ? 1.f : 0.f;
        out[i] = m;
        cnt += m;
    }
    cnt = reg_block_sum(cnt, red);
    if (threadIdx.x == 0) p.nfg[r * p.Bk + b] = cnt;
}

// ---- K1: the cosine rows.  Blocks [0, npairs*G*Bk): cross-layer rows (pair, g, b); then the complementary rows (l, b, h).
// A row's two operands are gathered into LDS once (head mean / 2 x 2 pooling on the way), the row loss goes to xloss /
// closs and, scaled by the row's known coefficient, its gradient rows to the workspace (reg_scatter_kernel adds them
// into the token maps' gradients).
__global__ __launch_bounds__(256) void reg_rows_kernel(RegParams p) {
    extern __shared__ float rows[];                 // x [N], r [N]
    __shared__ float red[4];
    const int t = threadIdx.x;
    const int nx = p.npairs * p.G * p.Bk;
    int N, demean, align;
    float gl, rgs;
    float *dxo, *dro, *lo;
    float* xs = rows;
    float* rs;
    if ((int)blockIdx.x < nx) {
        const int pi = blockIdx.x / (p.G * p.Bk), rem = blockIdx.x - pi * p.G * p.Bk, g = rem / p.Bk, b = rem - g * p.Bk;
        N = p.pN[pi];
        rs = rows + N;
        const int lx = p.px[pi], lr = p.pr[pi];
        const float invH = 1.0f / p.H;
        const float* tx = p.tm[lx] + (long)b * p.H * p.N[lx] * p.G + g;
        const float* tr = p.tm[lr] + (long)b * p.H * p.N[lr] * p.G + g;
        const int sx = p.side[lx], sc = p.side[lr];
        for (int n = t; n < N; n += 256) {
            float ax = 0.f, ar = 0.f;
            if (p.ppool[pi]) {
                const int y = n / sc, x = n - y * sc;
                const int n0 = (2 * y) * sx + 2 * x;
                for (int h = 0; h < p.H; ++h) {
                    const float* q = tx + ((long)h * p.N[lx] + n0) * p.G;
                    ax += 0.25f * ((q[0] + q[p.G]) + (q[(long)sx * p.G] + q[(long)(sx + 1) * p.G]));
                }
            } else {
                for (int h = 0; h < p.H; ++h) ax += tx[((long)h * p.N[lx] + n) * p.G];
            }
            for (int h = 0; h < p.H; ++h) ar += tr[((long)h * p.N[lr] + n) * p.G];
            xs[n] = ax * invH;
            rs[n] = ar * invH;
        }
        demean = 1; align = 1; rgs = 1.0f;
        gl = p.cx[g] * p.pw[pi] / p.Bk;
        lo = p.xloss + blockIdx.x;
        dxo = p.xdx + p.prowoff[pi] + (long)rem * N;
        dro = p.xdr + p.prowoff[pi] + (long)rem * N;
    } else {
        const int row = blockIdx.x - nx;
        int l = 0;
        while (l + 1 < p.L && row >= p.crow0[l + 1]) ++l;
        const int rem = row - p.crow0[l], b = rem / p.H, h = rem - b * p.H;
        N = p.N[l];
        rs = rows + N;
        const float* base = p.tm[l] + ((long)b * p.H + h) * N * p.G;
        for (int n = t; n < N; n += 256) {
            xs[n] = base[(long)n * p.G + 1];          // background map
            rs[n] = base[(long)n * p.G];              // subject map
        }
        demean = 0; align = 0; rgs = p.fg_grad_scale;
        gl = p.cc * p.lw[l] / (p.Bk * p.H);
        lo = p.closs + row;
        dxo = p.cdx + p.crowoff[l] + (long)rem * N;
        dro = p.cdr + p.crowoff[l] + (long)rem * N;
    }
    __syncthreads();
    float mx_ = 0.f, mr_ = 0.f;
    if (demean) {
        float a = 0.f, c = 0.f;
        for (int i = t; i < N; i += 256) { a += xs[i]; c += rs[i]; }
        mx_ = reg_block_sum(a, red) / N;
        mr_ = reg_block_sum(c, red) / N;
    }
    float P = 0.f, A = 0.f, Bq = 0.f;
    for (int i = t; i < N; i += 256) {
        const float xt = xs[i] - mx_, rt = rs[i] - mr_, tt = rt * fabsf(rt);
        P += xt * tt; A += xt * xt; Bq += tt * tt;
    }
    P = reg_block_sum(P, red);
    A = reg_block_sum(A, red) + 1e-12f;
    Bq = reg_block_sum(Bq, red) + 1e-12f;
    const float inv = 1.0f / sqrtf(A * Bq);
    const float c = P * inv;
    if (t == 0) *lo = align ? 1.0f - c : fmaxf(c, 0.f);
    const float gc = gl * (align ? -1.0f : (c > 0.f ? 1.0f : 0.f));
    float gmx = 0.f, gmr = 0.f;
    if (demean) {
        float a = 0.f, b2 = 0.f;
        for (int i = t; i < N; i += 256) {
            const float xt = xs[i] - mx_, rt = rs[i] - mr_, tt = rt * fabsf(rt);
            a += (tt - (P / A) * xt) * inv;
            b2 += (xt - (P / Bq) * tt) * inv * 2.0f * fabsf(rt);
        }
        gmx = reg_block_sum(a, red) / N;
        gmr = reg_block_sum(b2, red) / N;
    }
    for (int i = t; i < N; i += 256) {
        const float xt = xs[i] - mx_, rt = rs[i] - mr_, tt = rt * fabsf(rt);
        dxo[i] = gc * ((tt - (P / A) * xt) * inv - gmx);
        dro[i] = gc * rgs * ((xt - (P / Bq) * tt) * inv * 2.0f * fabsf(rt) - gmr);
    }
}

// ---- K2a-d: the mask hinges (see misc.hip hinge_*_kernel for the arithmetic), one block per (layer, instance), layers of any
// resolution.  S = column 0, G = column 1 of the layer's token map.
__global__ __launch_bounds__(256) void reg_hinge_avg_kernel(RegParams p) {
    __shared__ float red[4];
    const int lb = blockIdx.x, l = lb / p.Bk, b = lb - l * p.Bk, t = threadIdx.x;
    if (p.lw[l] == 0.f) return;
    const int N = p.N[l];
    const float* base = p.tm[l] + (long)b * p.H * N * p.G;
    const float* f = p.fgm + p.rmaskoff[p.res[l]] + (long)b * N;
    float sS = 0.f, sG = 0.f, nf = 0.f;
    for (int i = t; i < p.H * N; i += 256) {
        const int n = i % N;
        const float fv = f[n];
        sS += base[(long)i * p.G] * fv;
        if (p.have_bg) sG += base[(long)i * p.G + 1] * (1.f - fv);
        nf += fv;
    }
    sS = reg_block_sum(sS, red);
    sG = reg_block_sum(sG, red);
    nf = reg_block_sum(nf, red);
    if (t == 0) {
        p.havg[2 * lb] = sS / fmaxf(nf, 1e-6f);
        p.havg[2 * lb + 1] = sG / fmaxf((float)p.H * N - nf, 1e-6f);
    }
}

__global__ __launch_bounds__(256) void reg_hinge_sum_kernel(RegParams p) {
    __shared__ float red[4];
    const int lb = blockIdx.x, l = lb / p.Bk, b = lb - l * p.Bk, t = threadIdx.x;
    if (p.lw[l] == 0.f) return;
    const int N = p.N[l];
    const float* base = p.tm[l] + (long)b * p.H * N * p.G;
    const float* f = p.fgm + p.rmaskoff[p.res[l]] + (long)b * N;
    const float aS = p.havg[2 * lb], aG = p.havg[2 * lb + 1];
    float s[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < p.H * N; i += 256) {
        const int n = i % N;
        const float fv = f[n];
        const float S = base[(long)i * p.G], G = p.have_bg ? base[(long)i * p.G + 1] : 0.f;
        float x[4] = {S * (1.f - fv) + p.m - aS, G * fv + p.m - aG, G * fv + p.m3 - aS, S * (1.f - fv) + p.m - aG};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x[j] > 0.f) { s[j] += x[j]; c[j] += 1.f; }
    }
    const float w = p.iw ? p.iw[b] : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sj = reg_block_sum(s[j], red), cj = reg_block_sum(c[j], red);
        if (t == 0) { p.hpart[8 * lb + j] = sj * w; p.hpart[8 * lb + 4 + j] = cj; }
    }
}

// one block: per resolution the validity factor; per (term, layer) the hinge value and its count; then every reported
// part and the total, in a fixed order (partial sums by 256 threads, combined by thread 0)
__global__ __launch_bounds__(256) void reg_finish_kernel(RegParams p) {
    __shared__ float valid[REG_MAX_RES];
    __shared__ float xs[REG_MAX_PAIRS * 2], cs[REG_MAX_LAYERS], hs[4 * REG_MAX_LAYERS];
    const int t = threadIdx.x;
    if (t < REG_MAX_RES) {
        float v = 1.f;
        if (p.have_mask && t < p.nres) {
            const float s2 = (float)(p.rside[t] * p.rside[t]);
            for (int b = 0; b < p.Bk; ++b) {
                const float nf = p.nfg[t * p.Bk + b];
                if (!(nf > 0.f) || !(s2 - nf > 0.f)) v = 0.f;     // an instance without foreground or without background
            }
            p.valid[t] = v;
        }
        valid[t] = v;
    }
    if (t < p.npairs * p.G) {                                      // (pair, group): mean over the instances
        float s = 0.f;
        for (int b = 0; b < p.Bk; ++b) s += p.xloss[t * p.Bk + b];
        xs[t] = s / p.Bk;
    }
    if (t >= 64 && t < 64 + p.L) {                                 // layer: mean over (instance, head) of the complementary cosine
        const int l = t - 64;
        float s = 0.f;
        if (p.have_bg && p.lw[l] != 0.f)
            for (int r = 0; r < p.Bk * p.H; ++r) s += p.closs[p.crow0[l] + r];
        cs[l] = s / (p.Bk * p.H);
    }
    if (p.have_mask && t >= 128 && t < 128 + 4 * p.L) {
        const int i = t - 128, j = i / p.L, l = i - j * p.L;
        float s = 0.f, c = 0.f;
        if (p.lw[l] != 0.f)
            for (int b = 0; b < p.Bk; ++b) { s += p.hpart[8 * (l * p.Bk + b) + j]; c += p.hpart[8 * (l * p.Bk + b) + 4 + j]; }
        c = fmaxf(c, 1e-6f);
        const bool live = (p.have_bg || j == 0) && p.lw[l] != 0.f;
        p.hout[i] = live ? s / c : 0.f;
        hs[i] = live ? s / c : 0.f;
        p.hcnt[i] = c;
    }
    __syncthreads();
    if (t == 0) {
        float lx[2] = {0.f, 0.f};
        for (int pi = 0; pi < p.npairs; ++pi)
            for (int g = 0; g < p.G; ++g) lx[g] += xs[pi * p.G + g] * p.pw[pi];
        float lc = 0.f;
        for (int l = 0; l < p.L; ++l) lc += cs[l] * p.lw[l];
        float h4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.have_mask)
            for (int j = 0; j < 4; ++j)
                for (int l = 0; l < p.L; ++l) h4[j] += hs[j * p.L + l] * p.lw[l] * valid[p.res[l]];
        const float smb = h4[0] * 0.05f, bmf = h4[1] * 0.1f, con = (h4[2] + h4[3]) * 0.05f;
        p.parts[0] = lx[0]; p.parts[1] = lx[1]; p.parts[2] = lc; p.parts[3] = smb; p.parts[4] = bmf; p.parts[5] = con;
        p.parts[6] = p.cx[0] * lx[0] + p.cx[1] * lx[1] + p.cc * lc + p.c_smb * smb + p.c_bmf * bmf + p.c_con * con;
        p.parts[7] = 0.f;
    }
}

// gradient of the hinge terms into the token maps' gradients (plain stores: every element of the counted instances of a
// layer that takes part; reg_scatter_kernel adds the cosine rows afterwards)
__global__ __launch_bounds__(256) void reg_hinge_bwd_kernel(RegParams p) {
    __shared__ float red[4];
    const int lb = blockIdx.x, l = lb / p.Bk, b = lb - l * p.Bk, t = threadIdx.x;
    const int N = p.N[l];
    float* d = p.dtm[l] + (long)b * p.H * N * p.G;
    if (p.lw[l] == 0.f || !p.have_mask) {
        for (int i = t; i < p.H * N * p.G; i += 256) d[i] = 0.f;
        return;
    }
    const float* base = p.tm[l] + (long)b * p.H * N * p.G;
    const float* f = p.fgm + p.rmaskoff[p.res[l]] + (long)b * N;
    const float aS = p.havg[2 * lb], aG = p.havg[2 * lb + 1];
    const float w = p.iw ? p.iw[b] : 1.f;
    const float coef[4] = {p.c_smb * 0.05f, p.c_bmf * 0.1f, p.c_con * 0.05f, p.c_con * 0.05f};
    const float vl = p.valid[p.res[l]] * p.lw[l];
    float c[4], P[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = (p.have_bg || j == 0) ? coef[j] * vl / p.hcnt[j * p.L + l] * w : 0.f;
        P[j] = p.hpart[8 * lb + 4 + j];
    }
    float nf = 0.f;
    for (int n = t; n < N; n += 256) nf += f[n];
    nf = reg_block_sum(nf, red) * p.H;
    const float denS = fmaxf(nf, 1e-6f), denG = fmaxf((float)p.H * N - nf, 1e-6f);
    const float viaS = 0.5f * (c[0] * P[0] + c[2] * P[2]) / denS;
    const float viaG = (c[1] * P[1] + c[3] * P[3]) / denG;
    for (int i = t; i < p.H * N; i += 256) {
        const int n = i % N;
        const float fv = f[n];
        const float S = base[(long)i * p.G], G = p.have_bg ? base[(long)i * p.G + 1] : 0.f;
        const float x0 = S * (1.f - fv) + p.m - aS, x1 = G * fv + p.m - aG, x2 = G * fv + p.m3 - aS, x3 = S * (1.f - fv) + p.m - aG;
        const float p0 = x0 > 0.f, p1 = x1 > 0.f, p2 = x2 > 0.f, p3 = x3 > 0.f;
        d[(long)i * p.G] = (1.f - fv) * (c[0] * p0 + c[3] * p3) - fv * viaS;
        if (p.G > 1) d[(long)i * p.G + 1] = p.have_bg ? fv * (c[1] * p1 + c[2] * p2) - (1.f - fv) * viaG : 0.f;
    }
}

// ---- K3: every element of every gradient tensor, once: hinge gradient (already there) + complementary-cosine row +
// cross-layer rows (as x of its pairs, then as reference), instances beyond Bk get zeros.  One thread per (l, b, h, n).
__global__ __launch_bounds__(256) void reg_scatter_kernel(RegParams p, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int l = 0;
    while (l + 1 < p.L && idx >= p.elem0[l + 1]) ++l;
    int e = idx - p.elem0[l];
    const int N = p.N[l];
    const int n = e % N;
    e /= N;
    const int h = e % p.H, b = e / p.H;
    float* d = p.dtm[l] + (((long)b * p.H + h) * N + n) * p.G;
    if (b >= p.Bk) {
        for (int g = 0; g < p.G; ++g) d[g] = 0.f;
        return;
    }
    float v[2] = {d[0], p.G > 1 ? d[1] : 0.f};
    if (p.have_bg && p.lw[l] != 0.f) {
        const long o = p.crowoff[l] + ((long)b * p.H + h) * N + n;
        v[1] += p.cdx[o];
        v[0] += p.cdr[o];
    }
    const float invH = 1.0f / p.H;
    for (int pi = 0; pi < p.npairs; ++pi) {
        if (p.px[pi] == l) {
            int m = n;
            float sc = invH;
            if (p.ppool[pi]) {
                const int s = p.side[l], y = n / s, x = n - y * s;
                m = (y >> 1) * (s >> 1) + (x >> 1);
                sc *= 0.25f;
            }
            for (int g = 0; g < p.G; ++g) v[g] += p.xdx[p.prowoff[pi] + ((long)g * p.Bk + b) * p.pN[pi] + m] * sc;
        }
        if (p.pr[pi] == l)
            for (int g = 0; g < p.G; ++g) v[g] += p.xdr[p.prowoff[pi] + ((long)g * p.Bk + b) * p.pN[pi] + n] * invH;
    }
    for (int g = 0; g < p.G; ++g) d[g] = v[g];
}

static long reg_align4(long v) { return (v + 3) & ~3L; }

// workspace floats for adap_reg_losses (layer_N: host array of the L layers' pixel counts)
extern "C" long adap_reg_losses_workspace_floats(const int* layer_N, int L, const int* pair_x, const int* pair_r, int npairs, int Bk,
                                                 int H, int G) {
    if (!layer_N || L <= 0 || L > REG_MAX_LAYERS || npairs < 0 || npairs > REG_MAX_PAIRS) return -1;
    long tot = 0;
    for (int l = 0; l < L; ++l) tot += reg_align4((long)Bk * layer_N[l]);                 // masks (one per layer at most)
    tot += 4 + reg_align4((long)REG_MAX_RES * Bk);                                          // valid, foreground counts
    tot += reg_align4((long)npairs * G * Bk);                                              // xloss
    long crows = (long)L * Bk * H, cgrad = 0;
    for (int l = 0; l < L; ++l) cgrad += (long)Bk * H * layer_N[l];
    tot += reg_align4(crows) + 2 * reg_align4(cgrad);
    long xgrad = 0;
    for (int i = 0; i < npairs; ++i) {
        const int n = layer_N[pair_r[i]];
        xgrad += (long)G * Bk * n;
    }
    tot += 2 * reg_align4(xgrad);
    tot += reg_align4(2L * L * Bk) + reg_align4(8L * L * Bk) + 2 * reg_align4(4L * L);
    return tot;
}

extern "C" int adap_reg_losses(const void* const* tm, void* const* dtm, const int* layer_N, const float* complem_w, int L,
                               const int* pair_x, const int* pair_r, const float* pair_w, int npairs,
                               const float* fg_mask, int Hm, const float* inst_w, int Bt, int Bk, int H, int G, int have_bg,
                               float margin, float margin_bg_at_mf, float fg_grad_scale,
                               float cx_fg, float cx_bg, float cc_complem, float cc_smb, float cc_bmf, float cc_con,
                               float* parts, float* workspace, long ws_floats, void* stream) {
    ADAP_REQUIRE(tm && dtm && layer_N && complem_w && parts && workspace, ADAP_ERR_SHAPE, "reg_losses: null pointer");
    ADAP_REQUIRE(L >= 1 && L <= REG_MAX_LAYERS && npairs >= 0 && npairs <= REG_MAX_PAIRS, ADAP_ERR_UNSUPPORTED,
                 "reg_losses: %d layers / %d pairs (at most %d / %d)", L, npairs, REG_MAX_LAYERS, REG_MAX_PAIRS);
    ADAP_REQUIRE(npairs == 0 || (pair_x && pair_r && pair_w), ADAP_ERR_SHAPE, "reg_losses: pair tables");
    ADAP_REQUIRE(Bt >= 1 && Bk >= 1 && Bk <= Bt && H >= 1 && (G == 1 || G == 2), ADAP_ERR_SHAPE, "reg_losses: Bt %d Bk %d H %d G %d",
                 Bt, Bk, H, G);
    ADAP_REQUIRE(!have_bg || G == 2, ADAP_ERR_SHAPE, "reg_losses: the background column needs G == 2");
    ADAP_REQUIRE(ws_floats >= adap_reg_losses_workspace_floats(layer_N, L, pair_x, pair_r, npairs, Bk, H, G), ADAP_ERR_SHAPE,
                 "reg_losses: workspace too small");
    RegParams p;
    memset(&p, 0, sizeof(p));
    p.L = L; p.npairs = npairs; p.Bt = Bt; p.Bk = Bk; p.H = H; p.G = G; p.have_bg = have_bg ? 1 : 0;
    p.have_mask = fg_mask != nullptr;
    p.fg_mask = fg_mask; p.Hm = Hm; p.iw = inst_w;
    p.m = margin; p.m3 = margin_bg_at_mf; p.fg_grad_scale = fg_grad_scale;
    p.cx[0] = cx_fg; p.cx[1] = cx_bg; p.cc = cc_complem; p.c_smb = cc_smb; p.c_bmf = cc_bmf; p.c_con = cc_con;
    p.parts = parts;
    long off = 0, elems = 0;
    int maxN = 0;
    for (int l = 0; l < L; ++l) {
        ADAP_REQUIRE(tm[l] && dtm[l] && layer_N[l] >= 1, ADAP_ERR_SHAPE, "reg_losses: layer %d", l);
        int s = 1;
        while (s * s < layer_N[l]) ++s;
        ADAP_REQUIRE(s * s == layer_N[l], ADAP_ERR_SHAPE, "reg_losses: layer %d has %d pixels (not a square)", l, layer_N[l]);
        p.tm[l] = (const float*)tm[l]; p.dtm[l] = (float*)dtm[l]; p.N[l] = layer_N[l]; p.side[l] = s; p.lw[l] = complem_w[l];
        if (layer_N[l] > maxN) maxN = layer_N[l];
        int r = 0;
        while (r < p.nres && p.rside[r] != s) ++r;
        if (r == p.nres) {
            ADAP_REQUIRE(p.nres < REG_MAX_RES, ADAP_ERR_UNSUPPORTED, "reg_losses: more than %d resolutions", REG_MAX_RES);
            ADAP_REQUIRE(!fg_mask || (Hm >= s && Hm % s == 0 && ((Hm / s) & (Hm / s - 1)) == 0), ADAP_ERR_UNSUPPORTED,
                         "reg_losses: mask side %d vs attention side %d", Hm, s);
            p.rside[r] = s;
            p.rmaskoff[r] = off;
            off += reg_align4((long)Bk * layer_N[l]);
            ++p.nres;
        }
        p.res[l] = r;
        p.elem0[l] = (int)elems;
        elems += (long)Bt * H * layer_N[l];
    }
    ADAP_REQUIRE(elems < (1L << 31), ADAP_ERR_SHAPE, "reg_losses: too many elements");
    p.elem0[L] = (int)elems;
    // (the workspace query reserves one mask per LAYER; distinct resolutions need at most that)
    long fixed = 0;
    for (int l = 0; l < L; ++l) fixed += reg_align4((long)Bk * layer_N[l]);
    float* w = workspace;
    p.fgm = w; w += fixed;
    p.valid = w; w += 4;
    p.nfg = w; w += reg_align4((long)REG_MAX_RES * Bk);
    p.xloss = w; w += reg_align4((long)npairs * G * Bk);
    long crows = 0, cgrad = 0;
    for (int l = 0; l < L; ++l) {
        p.crow0[l] = (int)crows;
        p.crowoff[l] = cgrad;
        crows += (long)Bk * H;
        cgrad += (long)Bk * H * layer_N[l];
    }
    p.ncrows = (int)crows;
    p.closs = w; w += reg_align4((long)L * Bk * H);
    p.cdx = w; w += reg_align4(cgrad);
    p.cdr = w; w += reg_align4(cgrad);
    long xgrad = 0;
    for (int i = 0; i < npairs; ++i) {
        ADAP_REQUIRE(pair_x[i] >= 0 && pair_x[i] < L && pair_r[i] >= 0 && pair_r[i] < L, ADAP_ERR_SHAPE, "reg_losses: pair %d", i);
        const int nx = layer_N[pair_x[i]], nr = layer_N[pair_r[i]];
        ADAP_REQUIRE(nx == nr || nx == 4 * nr, ADAP_ERR_UNSUPPORTED,
                     "reg_losses: pair %d resizes %d -> %d pixels (same size or a factor-2 downscale)", i, nx, nr);
        p.px[i] = pair_x[i]; p.pr[i] = pair_r[i]; p.ppool[i] = nx != nr; p.pN[i] = nr; p.pw[i] = pair_w[i];
        p.prowoff[i] = xgrad;
        xgrad += (long)G * Bk * nr;
    }
    p.xdx = w; w += reg_align4(xgrad);
    p.xdr = w; w += reg_align4(xgrad);
    p.havg = w; w += reg_align4(2L * L * Bk);
    p.hpart = w; w += reg_align4(8L * L * Bk);
    p.hout = w; w += reg_align4(4L * L);
    p.hcnt = w; w += reg_align4(4L * L);
    ADAP_REQUIRE(w - workspace <= ws_floats, ADAP_ERR_SHAPE, "reg_losses: workspace layout overflow");
    static_assert(128 + 4 * REG_MAX_LAYERS <= 256 && REG_MAX_PAIRS * 2 <= 64, "reg_finish_kernel's thread roles");
    hipStream_t s = (hipStream_t)stream;
    if (p.have_mask) {
        hipLaunchKernelGGL(reg_mask_kernel, dim3(p.nres, Bk), dim3(256), 0, s, p);
        hipLaunchKernelGGL(reg_hinge_avg_kernel, dim3(L * Bk), dim3(256), 0, s, p);
        hipLaunchKernelGGL(reg_hinge_sum_kernel, dim3(L * Bk), dim3(256), 0, s, p);
    }
    const int nrows = npairs * G * Bk + (p.have_bg ? p.ncrows : 0);
    if (nrows > 0) {
        const size_t lds = (size_t)2 * maxN * sizeof(float);
        ADAP_REQUIRE(lds <= 64 * 1024, ADAP_ERR_UNSUPPORTED, "reg_losses: %d pixels per row", maxN);
        hipLaunchKernelGGL(reg_rows_kernel, dim3(nrows), dim3(256), lds, s, p);
    }
    hipLaunchKernelGGL(reg_finish_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(reg_hinge_bwd_kernel, dim3(L * Bk), dim3(256), 0, s, p);
    hipLaunchKernelGGL(reg_scatter_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, p, (int)elems);
    return adap_check_launch("reg_losses");
}

// =============================================================================================
// The static prompt-delta loss of the recon iteration (reference ldm/util.py:2037 calc_prompt_emb_delta_loss with
// ortho_subtract :280 and calc_ref_cosine_loss :437), value and gradient in one call -- ~45 torch launches otherwise.
// emb [4 Bs][R = layers x tokens][D]: subject-single, subject-comp, class-single, class-comp blocks.  Per row (b, r):
//   x = ortho(sc, ss),  ref = ortho(cc, cs),   ortho(a, b) = a - <a,b> / (<b,b> + 1e-6) b
//   loss_row = 1 - cos(x - mean x, t),  t = (ref - mean ref) |ref - mean ref|;  the reference's gradient scaled by cls_grad_scale
//   loss = mean_b [ sum_r loss_row w[b, tok(r)] / (sum_r w + 1e-8) ],  w = (m_single + m_comp)^2 / 4 with the start token's mask
//   zeroed IN the caller's mask tensor, as the reference does
// =============================================================================================
struct DeltaParams {
    const float* emb; float* demb; float* mask;       // mask [4 Bs][T] (the trailing 1 dropped)
    int Bs, L, T, D;
    float coef, cls_scale;
    float* wsum;      // [Bs]
    float* lossw;     // [Bs * L * T]  loss_row * w
    float* out;       // [2]: loss, coef * loss
};

__global__ __launch_bounds__(256) void delta_prep_kernel(DeltaParams p) {
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    float s = 0.f;
    for (int i = t; i < p.T; i += 256) {
        const float ms = i == 0 ? 0.f : p.mask[(long)b * p.T + i], mc = i == 0 ? 0.f : p.mask[(long)(p.Bs + b) * p.T + i];
        const float w = fmaxf((ms + mc) * (ms + mc) * 0.25f, 0.f);
        s += w;
    }
    s = reg_block_sum(s, red);
    if (t == 0) p.wsum[b] = s * p.L;
}

// after every reader of the old start-token entries is done (the rows kernel treats token 0 as masked by index anyway)
__global__ void delta_zero_start_kernel(DeltaParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * p.Bs) p.mask[(long)i * p.T] = 0.f;
}

#define DELTA_EPT 4           // elements per thread: D <= 1024
__global__ __launch_bounds__(256) void delta_rows_kernel(DeltaParams p) {
    __shared__ float red[4];
    const int row = blockIdx.x, t = threadIdx.x;
    const int R = p.L * p.T, b = row / R, r = row - b * R, tok = r % p.T;
    const long blk = (long)p.Bs * R * p.D, off = (long)row * p.D;
    const float* ss = p.emb + off;
    const float* sc = p.emb + blk + off;
    const float* cs = p.emb + 2 * blk + off;
    const float* cc = p.emb + 3 * blk + off;
    float a[DELTA_EPT], bb[DELTA_EPT], ca[DELTA_EPT], cb[DELTA_EPT];
    float ab = 0.f, b2 = 0.f, cab = 0.f, cb2 = 0.f;
#pragma unroll
    for (int e = 0; e < DELTA_EPT; ++e) {
        const int i = t + 256 * e;
        const bool ok = i < p.D;
        a[e] = ok ? sc[i] : 0.f; bb[e] = ok ? ss[i] : 0.f; ca[e] = ok ? cc[i] : 0.f; cb[e] = ok ? cs[i] : 0.f;
        ab += a[e] * bb[e]; b2 += bb[e] * bb[e]; cab += ca[e] * cb[e]; cb2 += cb[e] * cb[e];
    }
    ab = reg_block_sum(ab, red); b2 = reg_block_sum(b2, red) + 1e-6f;
    cab = reg_block_sum(cab, red); cb2 = reg_block_sum(cb2, red) + 1e-6f;
    const float c1 = ab / b2, c2 = cab / cb2;
    float x[DELTA_EPT], rr[DELTA_EPT];
    float sx = 0.f, sr = 0.f;
#pragma unroll
    for (int e = 0; e < DELTA_EPT; ++e) {
        x[e] = a[e] - c1 * bb[e];
        rr[e] = ca[e] - c2 * cb[e];
        sx += x[e]; sr += rr[e];                       // (lanes beyond D hold zeros)
    }
    sx = reg_block_sum(sx, red) / p.D;
    sr = reg_block_sum(sr, red) / p.D;
    float P = 0.f, A = 0.f, Bq = 0.f;
#pragma unroll
    for (int e = 0; e < DELTA_EPT; ++e) {
        if (t + 256 * e < p.D) {
            const float xt = x[e] - sx, rt = rr[e] - sr, tt = rt * fabsf(rt);
            P += xt * tt; A += xt * xt; Bq += tt * tt;
        }
    }
    P = reg_block_sum(P, red); A = reg_block_sum(A, red) + 1e-12f; Bq = reg_block_sum(Bq, red) + 1e-12f;
    const float inv = 1.0f / sqrtf(A * Bq), c = P * inv;
    const float ms = tok == 0 ? 0.f : p.mask[(long)b * p.T + tok], mc = tok == 0 ? 0.f : p.mask[(long)(p.Bs + b) * p.T + tok];
    const float w = fmaxf((ms + mc) * (ms + mc) * 0.25f, 0.f);
    if (t == 0) p.lossw[row] = (1.0f - c) * w;
    // d total / d loss_row, then through the cosine (align: d loss / d cos = -1) to x and ref
    const float gl = p.coef * w / (p.wsum[b] + 1e-8f) / p.Bs;
    const float gc = -gl;
    float gx[DELTA_EPT], gr[DELTA_EPT];
    float mx = 0.f, mr = 0.f;
#pragma unroll
    for (int e = 0; e < DELTA_EPT; ++e) {
        gx[e] = 0.f; gr[e] = 0.f;
        if (t + 256 * e < p.D) {
            const float xt = x[e] - sx, rt = rr[e] - sr, tt = rt * fabsf(rt);
            gx[e] = (tt - (P / A) * xt) * inv;
            gr[e] = (xt - (P / Bq) * tt) * inv * 2.0f * fabsf(rt);
            mx += gx[e]; mr += gr[e];
        }
    }
    mx = reg_block_sum(mx, red) / p.D;
    mr = reg_block_sum(mr, red) / p.D;
    float gxb = 0.f, grb = 0.f;
#pragma unroll
    for (int e = 0; e < DELTA_EPT; ++e) {
        if (t + 256 * e < p.D) {
            gx[e] = gc * (gx[e] - mx);
            gr[e] = gc * p.cls_scale * (gr[e] - mr);
        }
        gxb += gx[e] * bb[e];
        grb += gr[e] * cb[e];
    }
    gxb = reg_block_sum(gxb, red) / b2;          // (g . b) / (<b,b> + eps)
    grb = reg_block_sum(grb, red) / cb2;
    // out = a - c b, c = <a,b> / (<b,b> + eps):  dL/da = g - (g.b)/(bb) b ;  dL/db = -c g - (g.b)/(bb) a + 2 c (g.b)/(bb) b
    float* d_ss = p.demb + off;
    float* d_sc = p.demb + blk + off;
    float* d_cs = p.demb + 2 * blk + off;
    float* d_cc = p.demb + 3 * blk + off;
#pragma unroll
    for (int e = 0; e < DELTA_EPT; ++e) {
        const int i = t + 256 * e;
        if (i < p.D) {
            d_sc[i] = gx[e] - gxb * bb[e];
            d_ss[i] = -c1 * gx[e] - gxb * a[e] + 2.0f * c1 * gxb * bb[e];
            d_cc[i] = gr[e] - grb * cb[e];
            d_cs[i] = -c2 * gr[e] - grb * ca[e] + 2.0f * c2 * grb * cb[e];
        }
    }
}

__global__ __launch_bounds__(256) void delta_finish_kernel(DeltaParams p) {
    __shared__ float red[4];
    const int t = threadIdx.x, R = p.L * p.T;
    float tot = 0.f;
    for (int b = 0; b < p.Bs; ++b) {
        float s = 0.f;
        for (int i = t; i < R; i += 256) s += p.lossw[(long)b * R + i];
        s = reg_block_sum(s, red);
        tot += s / (p.wsum[b] + 1e-8f);
    }
    if (t == 0) {
        p.out[0] = tot / p.Bs;
        p.out[1] = p.coef * tot / p.Bs;
    }
}

extern "C" long adap_prompt_delta_loss_workspace_floats(int Bs, int L, int T) { return reg_align4(Bs) + reg_align4((long)Bs * L * T); }

extern "C" int adap_prompt_delta_loss(const float* emb4, float* demb4, float* mask4, int Bs, int L, int T, int D, float coef,
                                      float cls_grad_scale, float* out2, float* workspace, void* stream) {
    ADAP_REQUIRE(emb4 && demb4 && mask4 && out2 && workspace, ADAP_ERR_SHAPE, "prompt_delta_loss: null pointer");
    ADAP_REQUIRE(Bs >= 1 && L >= 1 && T >= 1 && D >= 1 && D <= 256 * DELTA_EPT, ADAP_ERR_UNSUPPORTED,
                 "prompt_delta_loss: Bs %d L %d T %d D %d (D <= %d)", Bs, L, T, D, 256 * DELTA_EPT);
    DeltaParams p = {emb4, demb4, mask4, Bs, L, T, D, coef, cls_grad_scale, workspace, workspace + reg_align4(Bs), out2};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(delta_prep_kernel, dim3(Bs), dim3(256), 0, s, p);
    hipLaunchKernelGGL(delta_rows_kernel, dim3((unsigned)((long)Bs * L * T)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(delta_finish_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(delta_zero_start_kernel, dim3((4 * Bs + 63) / 64), dim3(64), 0, s, p);
    return adap_check_launch("prompt_delta_loss");
}

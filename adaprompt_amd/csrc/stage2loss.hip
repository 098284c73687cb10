// Stage-2 (compositional distillation) elastic matching loss, value and gradient, as fixed-order f32 kernels
// (reference ldm/util.py:2241-2368 calc_elastic_matching_loss, called per distillation layer from
// calc_comp_fg_bg_preserve_loss ddpm.py:4389-4551).
//
// The batch is four blocks of ONE instance: (subject single, subject comp, mix single, mix comp) = (ss, sc, ms, mc).
//   q  f32 [4][Cq][N]   pooled attention queries,        f  f32 [4][Cf][N]   pooled output features,
//   fg f32 [N]          1 where the single instance's pooled foreground mask is non-zero.
// With i a token of the single instance and j a token of the comp instance:
//   S[i][j]  = sum_c ss_q[c][i] sc_q[c][j]      P  = softmax_j S      (sc_map_ss_prob[j][i] of the reference)
//   Sm[i][j] = sum_c ms_q[c][i] mc_q[c][j]      Pm = softmax_j Sm
//   map_align = sum_{i, j in fg} |P - Pm| / max(n_fg^2, 1e-6)
//   R[:, i]   = sum_j P[i][j] sc_f[:, j]        sc_ss_fg = mean_{i in fg} (1 - cos(R[:, i], t(ss_f[:, i]))),  t(x) = x |x|
//   p_sc[j]   = sum_{i in fg} P[i][j]           sc_below = max(cutoff - p_sc, 0)         (mc_below from Pm likewise)
//   sc_mc_bg  = sum_j w_j (1 - cos(sc_f[:, j], t(mc_f[:, j]))) / (sum_j w_j + 1e-8),   w = mc_below
// (cos with the 1e-12 floors of F.cosine_embedding_loss).  Gradient scales of the reference's ScaleGrad: ss_q and ms_q
// gs_q, ss_f gs_feat, mc_f (as the cosine's reference) gs_mix; the weights w carry gradient into Pm, and the two `below`
// vectors are outputs with incoming gradients of their own (the background-suppression terms use them as masks).
//
// Why not the torch expressions: the products went to a vendor GEMM whose summation order moves between runs (the two
// absolute floors the Stage-2 parity gates carried), and the chain was ~70 launches per layer each way.  Here every sum has
// a fixed order: the products are a plain LDS-tiled f32 MFMA kernel (v_mfma_f32_16x16x4_f32, K walked in order, no split
// K, no atomics), the reductions are per-thread strided sums + a fixed tree.  6 launches forward, 9 backward.
#include "common.h"

#define EM_THREADS 256

struct EmGemm {
    const float* A;
    const float* B;
    float* C;
    long sam, sak, sbk, sbn, ldc;       // A(m, k) = A[m sam + k sak], B(k, n) = B[k sbk + n sbn], C[m ldc + n]
    long za, zb, zc;                    // offsets per blockIdx.z
    int M, N, K;
    float alpha;
    int accumulate;                     // C += alpha A B instead of C = alpha A B
};

// C = alpha A B, 64 x 64 tile per workgroup, 4 waves of 32 x 32 (2 x 2 MFMA 16x16x4 f32 tiles), K in chunks of 16 through LDS.
// A_MC: A's m index is the contiguous one (else k); B_NC: B's n index is the contiguous one (else k) -- only chooses which
// index the lanes of a wave walk when staging, so that the global loads coalesce.
template <bool A_MC, bool B_NC>
__global__ __launch_bounds__(EM_THREADS) void em_gemm_kernel(EmGemm g) {
    __shared__ float sA[16][80];        // [k][m]; 80: the 4 k rows one MFMA operand read touches fall in 4 x 16 distinct banks
    __shared__ float sB[16][80];        // [k][n]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* A = g.A + blockIdx.z * g.za;
    const float* B = g.B + blockIdx.z * g.zb;
    float* C = g.C + blockIdx.z * g.zc;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this thread's four (m, k) / (k, n) positions of a chunk, and the loads of one chunk into registers
    int am[4], ak[4], bn_[4], bk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (A_MC) { am[r] = t & 63; ak[r] = (t >> 6) + 4 * r; }
        else { ak[r] = t & 15; am[r] = (t >> 4) + 16 * r; }
        if (B_NC) { bn_[r] = t & 63; bk[r] = (t >> 6) + 4 * r; }
        else { bk[r] = t & 15; bn_[r] = (t >> 4) + 16 * r; }
    }
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ra[r] = (m0 + am[r] < g.M && k0 + ak[r] < g.K) ? A[(long)(m0 + am[r]) * g.sam + (long)(k0 + ak[r]) * g.sak] : 0.f;
            rb[r] = (n0 + bn_[r] < g.N && k0 + bk[r] < g.K) ? B[(long)(k0 + bk[r]) * g.sbk + (long)(n0 + bn_[r]) * g.sbn] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sA[ak[r]][am[r]] = ra[r];
            sB[bk[r]][bn_[r]] = rb[r];
        }
        __syncthreads();
        if (k0 + 16 < g.K) fetch(k0 + 16);             // the next chunk's loads are in flight under this chunk's MFMAs
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            const int kr = kk + (lane >> 4), c = lane & 15;
            const float a0 = sA[kr][wm + c], a1 = sA[kr][wm + 16 + c];
            const float b0 = sB[kr][wn + c], b1 = sB[kr][wn + 16 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // accumulator r of tile (a, b): row wm + 16 a + 4 (lane / 16) + r, column wn + 16 b + lane % 16
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * a + 4 * (lane >> 4) + r, n = n0 + wn + 16 * b + (lane & 15);
                if (m < g.M && n < g.N) {
                    float* dst = C + (long)m * g.ldc + n;
                    const float v = g.alpha * acc[a][b][r];
                    *dst = g.accumulate ? *dst + v : v;
                }
            }
}

static void em_gemm(bool a_mc, bool b_nc, const EmGemm& g, int nz, hipStream_t s) {
    dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, nz);
    if (a_mc && b_nc) hipLaunchKernelGGL((em_gemm_kernel<true, true>), grid, dim3(EM_THREADS), 0, s, g);
    else if (!a_mc && !b_nc) hipLaunchKernelGGL((em_gemm_kernel<false, false>), grid, dim3(EM_THREADS), 0, s, g);
    else if (!a_mc && b_nc) hipLaunchKernelGGL((em_gemm_kernel<false, true>), grid, dim3(EM_THREADS), 0, s, g);
    else hipLaunchKernelGGL((em_gemm_kernel<true, false>), grid, dim3(EM_THREADS), 0, s, g);
}

__device__ __forceinline__ float em_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

__device__ __forceinline__ float em_block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one workgroup per single-instance token i: both rows S[i][:], Sm[i][:] -> softmax in place; then this row's share of
// map_align.  Thread t owns columns t, t + 256, ... in every loop, so the in-place rewrite needs no barrier of its own.
__global__ __launch_bounds__(EM_THREADS) void em_softmax_kernel(float* __restrict__ P2, const float* __restrict__ fg,
                                                                float* __restrict__ rowpart, int N) {
    __shared__ float red[4];
    const int i = blockIdx.x, t = threadIdx.x;
    float* p0 = P2 + (long)i * N;
    float* p1 = p0 + (long)N * N;
#pragma unroll
    for (int z = 0; z < 2; ++z) {
        float* p = z ? p1 : p0;
        float mx = -INFINITY;
        for (int j = t; j < N; j += EM_THREADS) mx = fmaxf(mx, p[j]);
        mx = em_block_max(mx, red);
        float s = 0.f;
        for (int j = t; j < N; j += EM_THREADS) s += expf(p[j] - mx);
        s = em_block_sum(s, red);
        const float inv = 1.0f / s;
        for (int j = t; j < N; j += EM_THREADS) p[j] = expf(p[j] - mx) * inv;
    }
    float d = 0.f;
    if (fg[i] != 0.f)
        for (int j = t; j < N; j += EM_THREADS)
            if (fg[j] != 0.f) d += fabsf(p0[j] - p1[j]);
    d = em_block_sum(d, red);
    if (t == 0) rowpart[i] = d;
}

// p[z][j] = sum_{i in fg} P2[z][i][j]: 64 columns x 4 row slices per workgroup, the slices added in a fixed order
__global__ __launch_bounds__(EM_THREADS) void em_colsum_kernel(const float* __restrict__ P2, const float* __restrict__ fg,
                                                               float* __restrict__ out, int N) {
    __shared__ float part[4][64];
    const int t = threadIdx.x, col = t & 63, sl = t >> 6, z = blockIdx.y;
    const int j = blockIdx.x * 64 + col;
    const float* P = P2 + (long)z * N * N;
    float acc = 0.f;
    if (j < N)
        for (int i = sl; i < N; i += 4)
            if (fg[i] != 0.f) acc += P[(long)i * N + j];
    part[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && j < N) out[(long)z * N + j] = ((part[0][col] + part[1][col]) + part[2][col]) + part[3][col];
}

// per token (column of [C][N] arrays): <a, t(b)>, |a|^2 + 1e-12, |t(b)|^2 + 1e-12 -> three rows of `dst`
#define EM_CS 16                                        // channel slices per workgroup (64 tokens x 16 slices = 1024 threads)
__global__ __launch_bounds__(64 * EM_CS) void em_coscols_kernel(const float* __restrict__ a0, const float* __restrict__ b0,
                                                                float* __restrict__ d0, const float* __restrict__ a1,
                                                                const float* __restrict__ b1, float* __restrict__ d1, int C,
                                                                int N) {
    __shared__ float part[3][EM_CS][64];
    const int t = threadIdx.x, col = t & 63, sl = t >> 6;
    const int i = blockIdx.x * 64 + col;
    const float* a = blockIdx.y ? a1 : a0;
    const float* b = blockIdx.y ? b1 : b0;
    float* dst = blockIdx.y ? d1 : d0;
    float P = 0.f, A = 0.f, Bq = 0.f;
    if (i < N)
        for (int c = sl; c < C; c += EM_CS) {
            const float x = a[(long)c * N + i], r = b[(long)c * N + i], tt = r * fabsf(r);
            P += x * tt; A += x * x; Bq += tt * tt;
        }
    part[0][sl][col] = P; part[1][sl][col] = A; part[2][sl][col] = Bq;
    __syncthreads();
    if (sl < 3 && i < N) {                              // slice threads 0, 1, 2 each sum one of the three quantities, in order
        float v = 0.f;
        for (int k = 0; k < EM_CS; ++k) v += part[sl][k][col];
        dst[(long)sl * N + i] = v + (sl ? 1e-12f : 0.f);
    }
}

// one workgroup: the three losses, the two `below` vectors, the weight sum and the foreground count
__global__ __launch_bounds__(EM_THREADS) void em_finalize_kernel(const float* __restrict__ fg, float* __restrict__ tok,
                                                                 float* __restrict__ out, int N, float cutoff) {
    __shared__ float red[4];
    const int t = threadIdx.x;
    float nfg = 0.f, lmap = 0.f, lfg = 0.f, W = 0.f, lbg = 0.f;
    for (int i = t; i < N; i += EM_THREADS) {
        const float f = fg[i] != 0.f ? 1.f : 0.f;
        nfg += f;
        lmap += tok[(long)ADAP_EM_ROWPART * N + i];
        const float cf = tok[(long)ADAP_EM_DOT_FG * N + i] /
                         sqrtf(tok[(long)(ADAP_EM_DOT_FG + 1) * N + i] * tok[(long)(ADAP_EM_DOT_FG + 2) * N + i]);
        lfg += f * (1.0f - cf);
        const float scb = fmaxf(cutoff - tok[(long)ADAP_EM_P_SC * N + i], 0.f);
        const float mcb = fmaxf(cutoff - tok[(long)ADAP_EM_P_MC * N + i], 0.f);
        tok[(long)ADAP_EM_SC_BELOW * N + i] = scb;
        tok[(long)ADAP_EM_MC_BELOW * N + i] = mcb;
        const float cb = tok[(long)ADAP_EM_DOT_BG * N + i] /
                         sqrtf(tok[(long)(ADAP_EM_DOT_BG + 1) * N + i] * tok[(long)(ADAP_EM_DOT_BG + 2) * N + i]);
        W += mcb;
        lbg += mcb * (1.0f - cb);
    }
    nfg = em_block_sum(nfg, red);
    lmap = em_block_sum(lmap, red);
    lfg = em_block_sum(lfg, red);
    W = em_block_sum(W, red);
    lbg = em_block_sum(lbg, red);
    if (t == 0) {
        out[ADAP_EM_OUT_MAP] = lmap / fmaxf(nfg * nfg, 1e-6f);
        out[ADAP_EM_OUT_FG] = nfg > 0.f ? lfg / nfg : 0.f;
        out[ADAP_EM_OUT_BG] = lbg / (W + 1e-8f);
        out[ADAP_EM_OUT_W] = W;
        out[ADAP_EM_OUT_NFG] = nfg;
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// per-token coefficients.  For a cosine row loss l = 1 - <x, t> / sqrt(A B) with incoming weight g:
//   dl/dx = g (-(1/sqrt(AB)) t + (P/A)/sqrt(AB) x),   dl/dt = g (-(1/sqrt(AB)) x + (P/B)/sqrt(AB) t)
// rows of coef: 0 / 1 multiply (t, x) in d/dx of the foreground term, 2 / 3 multiply (x, t) in its d/dt, 4..7 the same for
// the background term, 8 / 9 d total / d p_sc, d total / d p_mc.
__global__ __launch_bounds__(EM_THREADS) void em_coef_kernel(const float* __restrict__ fg, const float* __restrict__ tok,
                                                             const float* __restrict__ out, const float* __restrict__ g_fg,
                                                             const float* __restrict__ g_bg, const float* __restrict__ g_scb,
                                                             const float* __restrict__ g_mcb, float* __restrict__ coef, int N,
                                                             float cutoff) {
    const int i = blockIdx.x * EM_THREADS + threadIdx.x;
    if (i >= N) return;
    const float nfg = out[ADAP_EM_OUT_NFG], W = out[ADAP_EM_OUT_W], Lbg = out[ADAP_EM_OUT_BG];
    const float gfg = g_fg ? *g_fg : 0.f, gbg = g_bg ? *g_bg : 0.f;
    {
        const float P = tok[(long)ADAP_EM_DOT_FG * N + i], A = tok[(long)(ADAP_EM_DOT_FG + 1) * N + i],
                    B = tok[(long)(ADAP_EM_DOT_FG + 2) * N + i];
        const float inv = 1.0f / sqrtf(A * B);
        const float gi = (fg[i] != 0.f && nfg > 0.f) ? gfg / nfg : 0.f;
        coef[0 * (long)N + i] = -gi * inv;
        coef[1 * (long)N + i] = gi * (P / A) * inv;
        coef[2 * (long)N + i] = -gi * inv;
        coef[3 * (long)N + i] = gi * (P / B) * inv;
    }
    const float P = tok[(long)ADAP_EM_DOT_BG * N + i], A = tok[(long)(ADAP_EM_DOT_BG + 1) * N + i],
                B = tok[(long)(ADAP_EM_DOT_BG + 2) * N + i];
    const float inv = 1.0f / sqrtf(A * B);
    const float w = tok[(long)ADAP_EM_MC_BELOW * N + i];
    const float gi = gbg * w / (W + 1e-8f);
    coef[4 * (long)N + i] = -gi * inv;
    coef[5 * (long)N + i] = gi * (P / A) * inv;
    coef[6 * (long)N + i] = -gi * inv;
    coef[7 * (long)N + i] = gi * (P / B) * inv;
    const float lb = 1.0f - P * inv;
    const float dW = gbg * (lb - Lbg) / (W + 1e-8f);
    // clamp(cutoff - p, min = 0): the gradient passes where cutoff - p >= 0 (torch's clamp backward)
    const float psc = tok[(long)ADAP_EM_P_SC * N + i], pmc = tok[(long)ADAP_EM_P_MC * N + i];
    coef[8 * (long)N + i] = (cutoff - psc >= 0.f) ? -(g_scb ? g_scb[i] : 0.f) : 0.f;
    coef[9 * (long)N + i] = (cutoff - pmc >= 0.f) ? -((g_mcb ? g_mcb[i] : 0.f) + dW) : 0.f;
}

// every element of the [Cf][N] feature arrays once: dR^T, and the cosine terms' share of d ss_f, d sc_f, d mc_f (d ms_f = 0)
__global__ __launch_bounds__(EM_THREADS) void em_featgrad_kernel(const float* __restrict__ f, const float* __restrict__ RT,
                                                                 const float* __restrict__ coef, float* __restrict__ dRT,
                                                                 float* __restrict__ df, int C, int N, float gs_feat,
                                                                 float gs_mix) {
    const int i = blockIdx.x * EM_THREADS + threadIdx.x, c = blockIdx.y;
    if (i >= N) return;
    const long CN = (long)C * N, e = (long)c * N + i;
    const float ss = f[e], sc = f[CN + e], mc = f[3 * CN + e], R = RT[e];
    const float t = ss * fabsf(ss), tm = mc * fabsf(mc);
    dRT[e] = coef[i] * t + coef[(long)N + i] * R;
    df[e] = gs_feat * 2.0f * fabsf(ss) * (coef[2 * (long)N + i] * R + coef[3 * (long)N + i] * t);
    df[CN + e] = coef[4 * (long)N + i] * tm + coef[5 * (long)N + i] * sc;
    df[2 * CN + e] = 0.f;
    df[3 * CN + e] = gs_mix * 2.0f * fabsf(mc) * (coef[6 * (long)N + i] * sc + coef[7 * (long)N + i] * tm);
}

// one workgroup per row i: dP[z][i][:] assembled on the fly (z = 0: the recon product already in dS2[0] + map_align + p_sc;
// z = 1: map_align with the other sign + p_mc), softmax backward, written over dS2
__global__ __launch_bounds__(EM_THREADS) void em_softmax_bwd_kernel(const float* __restrict__ P2, const float* __restrict__ fg,
                                                                    const float* __restrict__ coef, const float* __restrict__ out,
                                                                    const float* __restrict__ g_map, float* __restrict__ dS2,
                                                                    int N) {
    __shared__ float red[4];
    const int i = blockIdx.x, t = threadIdx.x;
    const float nfg = out[ADAP_EM_OUT_NFG];
    const float fi = fg[i] != 0.f ? 1.f : 0.f;
    const float gm = (g_map ? *g_map : 0.f) * fi / fmaxf(nfg * nfg, 1e-6f);
    const float* p0 = P2 + (long)i * N;
    const float* p1 = p0 + (long)N * N;
    float* d0 = dS2 + (long)i * N;
    float* d1 = d0 + (long)N * N;
    const float* c_sc = coef + 8 * (long)N;
    const float* c_mc = coef + 9 * (long)N;
    auto dp0 = [&](int j) {
        const float df = p0[j] - p1[j];
        const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        return d0[j] + (fg[j] != 0.f ? gm * sg : 0.f) + fi * c_sc[j];
    };
    auto dp1 = [&](int j) {
        const float df = p0[j] - p1[j];
        const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        return (fg[j] != 0.f ? -gm * sg : 0.f) + fi * c_mc[j];
    };
    float s0 = 0.f, s1 = 0.f;
    for (int j = t; j < N; j += EM_THREADS) { s0 += p0[j] * dp0(j); s1 += p1[j] * dp1(j); }
    s0 = em_block_sum(s0, red);
    s1 = em_block_sum(s1, red);
    for (int j = t; j < N; j += EM_THREADS) {
        const float a = p0[j] * (dp0(j) - s0), b = p1[j] * (dp1(j) - s1);
        d0[j] = a;
        d1[j] = b;
    }
}

extern "C" int adap_elastic_match_fwd(const float* q, int Cq, const float* f, int Cf, const float* fg, int N, float cutoff,
                                      float* P2, float* RT, float* tok, float* out, void* stream) {
    ADAP_REQUIRE(q && f && fg && P2 && RT && tok && out, ADAP_ERR_SHAPE, "elastic_match_fwd: null pointer");
    ADAP_REQUIRE(Cq >= 1 && Cf >= 1 && N >= 1 && N <= 16384, ADAP_ERR_SHAPE, "elastic_match_fwd: Cq %d Cf %d N %d", Cq, Cf, N);
    hipStream_t s = (hipStream_t)stream;
    const long NN = (long)N * N, qN = (long)Cq * N, fN = (long)Cf * N;
    // S[z][i][j] = sum_c q[2z][c][i] q[2z + 1][c][j]
    EmGemm g{};
    g.A = q; g.B = q + qN; g.C = P2; g.sam = 1; g.sak = N; g.sbk = N; g.sbn = 1; g.ldc = N;
    g.za = 2 * qN; g.zb = 2 * qN; g.zc = NN; g.M = N; g.N = N; g.K = Cq; g.alpha = 1.f; g.accumulate = 0;
    em_gemm(true, true, g, 2, s);
    hipLaunchKernelGGL(em_softmax_kernel, dim3(N), dim3(EM_THREADS), 0, s, P2, fg, tok + (long)ADAP_EM_ROWPART * N, N);
    hipLaunchKernelGGL(em_colsum_kernel, dim3((N + 63) / 64, 2), dim3(EM_THREADS), 0, s, P2, fg, tok + (long)ADAP_EM_P_SC * N, N);
    // R^T[c][i] = sum_j sc_f[c][j] P[i][j]
    EmGemm r{};
    r.A = f + fN; r.B = P2; r.C = RT; r.sam = N; r.sak = 1; r.sbk = 1; r.sbn = N; r.ldc = N;
    r.M = Cf; r.N = N; r.K = N; r.alpha = 1.f; r.accumulate = 0;
    em_gemm(false, false, r, 1, s);
    hipLaunchKernelGGL(em_coscols_kernel, dim3((N + 63) / 64, 2), dim3(64 * EM_CS), 0, s, RT, f, tok + (long)ADAP_EM_DOT_FG * N,
                       f + fN, f + 3 * fN, tok + (long)ADAP_EM_DOT_BG * N, Cf, N);
    hipLaunchKernelGGL(em_finalize_kernel, dim3(1), dim3(EM_THREADS), 0, s, fg, tok, out, N, cutoff);
    return adap_check_launch("elastic_match_fwd");
}

extern "C" int adap_elastic_match_bwd(const float* q, int Cq, const float* f, int Cf, const float* fg, int N, float cutoff,
                                      float gs_q, float gs_feat, float gs_mix, const float* P2, const float* RT,
                                      const float* tok, const float* out, const float* g_map, const float* g_fg,
                                      const float* g_bg, const float* g_scb, const float* g_mcb, float* dS2, float* dRT,
                                      float* coef, float* dq, float* df, void* stream) {
    ADAP_REQUIRE(q && f && fg && P2 && RT && tok && out && dS2 && dRT && coef && dq && df, ADAP_ERR_SHAPE,
                 "elastic_match_bwd: null pointer");
    ADAP_REQUIRE(Cq >= 1 && Cf >= 1 && N >= 1 && N <= 16384, ADAP_ERR_SHAPE, "elastic_match_bwd: Cq %d Cf %d N %d", Cq, Cf, N);
    hipStream_t s = (hipStream_t)stream;
    const long NN = (long)N * N, qN = (long)Cq * N, fN = (long)Cf * N;
    hipLaunchKernelGGL(em_coef_kernel, dim3((N + EM_THREADS - 1) / EM_THREADS), dim3(EM_THREADS), 0, s, fg, tok, out, g_fg, g_bg,
                       g_scb, g_mcb, coef, N, cutoff);
    hipLaunchKernelGGL(em_featgrad_kernel, dim3((N + EM_THREADS - 1) / EM_THREADS, Cf), dim3(EM_THREADS), 0, s, f, RT, coef, dRT,
                       df, Cf, N, gs_feat, gs_mix);
    // the recon term's d P[i][j] = sum_c dR^T[c][i] sc_f[c][j]  -> dS2[0]
    EmGemm a{};
    a.A = dRT; a.B = f + fN; a.C = dS2; a.sam = 1; a.sak = N; a.sbk = N; a.sbn = 1; a.ldc = N;
    a.M = N; a.N = N; a.K = Cf; a.alpha = 1.f; a.accumulate = 0;
    em_gemm(true, true, a, 1, s);
    // d sc_f[c][j] += sum_i dR^T[c][i] P[i][j]
    EmGemm b{};
    b.A = dRT; b.B = P2; b.C = df + fN; b.sam = N; b.sak = 1; b.sbk = N; b.sbn = 1; b.ldc = N;
    b.M = Cf; b.N = N; b.K = N; b.alpha = 1.f; b.accumulate = 1;
    em_gemm(false, true, b, 1, s);
    hipLaunchKernelGGL(em_softmax_bwd_kernel, dim3(N), dim3(EM_THREADS), 0, s, P2, fg, coef, out, g_map, dS2, N);
    // d q[2z][c][i] = gs_q sum_j q[2z + 1][c][j] dS[z][i][j];   d q[2z + 1][c][j] = sum_i q[2z][c][i] dS[z][i][j]
    EmGemm c{};
    c.A = q + qN; c.B = dS2; c.C = dq; c.sam = N; c.sak = 1; c.sbk = 1; c.sbn = N; c.ldc = N;
    c.za = 2 * qN; c.zb = NN; c.zc = 2 * qN; c.M = Cq; c.N = N; c.K = N; c.alpha = gs_q; c.accumulate = 0;
    em_gemm(false, false, c, 2, s);
    EmGemm d{};
    d.A = q; d.B = dS2; d.C = dq + qN; d.sam = N; d.sak = 1; d.sbk = N; d.sbn = 1; d.ldc = N;
    d.za = 2 * qN; d.zb = NN; d.zc = 2 * qN; d.M = Cq; d.N = N; d.K = N; d.alpha = 1.f; d.accumulate = 0;
    em_gemm(false, true, d, 2, s);
    return adap_check_launch("elastic_match_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------
// calc_prompt_mix_loss's per-layer terms on the subject tokens' score maps (ddpm.py:3714-3930; ldm/util.py:543-594
// calc_delta_alignment_loss "feat_to_ref" with the cosine of exponent 3, and the L1 between mean scores).
//   a f32 [4][H][N] = (ss, sc, ms, mc) score maps summed over the subject tokens, one instance, H heads.  Per head:
//     src = ss - c1 ms, c1 = <ss,ms> / (<ms,ms> + 1e-6);   tgt = sc - c2 mc, c2 = <sc,mc> / (<mc,mc> + 1e-6)
//     delta = 1 - cos(tgt, src^3)            norm = |mean sc - mean mc| + |mean ss - mean ms|
//   out[0] = mean_h delta, out[1] = mean_h norm.  The mix maps (ms, mc) carry gradient scaled by gs_mix (0.05).
// One workgroup per head (fixed summation order inside it), the heads averaged in order by a one-thread finish.
// rec f32 [H][ADAP_PM_REC] is the forward's record for the backward.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EM_THREADS) void pm_attnterms_fwd_kernel(const float* __restrict__ a, int H, int N,
                                                                      float* __restrict__ rec) {
    __shared__ float red[4];
    const int t = threadIdx.x;
    const long HN = (long)H * N;
    {
        const int h = blockIdx.x;                      // one workgroup per head; pm_attnterms_finish_kernel averages them in order
        const float* ss = a + (long)h * N;
        const float* sc = ss + HN;
        const float* ms = ss + 2 * HN;
        const float* mc = ss + 3 * HN;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int n = t; n < N; n += EM_THREADS) {
            const float x0 = ss[n], x1 = sc[n], x2 = ms[n], x3 = mc[n];
            v[0] += x0 * x2; v[1] += x2 * x2; v[2] += x1 * x3; v[3] += x3 * x3;
            v[4] += x0; v[5] += x1; v[6] += x2; v[7] += x3;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = em_block_sum(v[k], red);
        const float c1 = v[0] / (v[1] + 1e-6f), c2 = v[2] / (v[3] + 1e-6f);
        float P = 0.f, A = 0.f, B = 0.f;
        for (int n = t; n < N; n += EM_THREADS) {
            const float src = ss[n] - c1 * ms[n], tgt = sc[n] - c2 * mc[n], tt = src * src * src;
            P += tgt * tt; A += tgt * tgt; B += tt * tt;
        }
        P = em_block_sum(P, red);
        A = em_block_sum(A, red) + 1e-12f;
        B = em_block_sum(B, red) + 1e-12f;
        const float m0 = v[4] / N, m1 = v[5] / N, m2 = v[6] / N, m3 = v[7] / N;
        if (t == 0) {
            float* r = rec + (long)h * ADAP_PM_REC;
            r[0] = c1; r[1] = c2; r[2] = P; r[3] = A; r[4] = B; r[5] = m1 - m3; r[6] = m0 - m2; r[7] = v[1] + 1e-6f; r[8] = v[3] + 1e-6f;
            r[9] = 1.0f - P / sqrtf(A * B);
            r[10] = fabsf(m1 - m3) + fabsf(m0 - m2);
        }
    }
}

__global__ void pm_attnterms_finish_kernel(const float* __restrict__ rec, int H, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float ld = 0.f, ln = 0.f;
        for (int h = 0; h < H; ++h) { ld += rec[(long)h * ADAP_PM_REC + 9]; ln += rec[(long)h * ADAP_PM_REC + 10]; }
        out[0] = ld / H;
        out[1] = ln / H;
    }
}

__global__ __launch_bounds__(EM_THREADS) void pm_attnterms_bwd_kernel(const float* __restrict__ a, int H, int N,
                                                                      const float* __restrict__ rec, const float* __restrict__ g_delta,
                                                                      const float* __restrict__ g_norm, float gs_mix,
                                                                      float* __restrict__ da) {
    __shared__ float red[4];
    const int t = threadIdx.x;
    const long HN = (long)H * N;
    const float gd = (g_delta ? *g_delta : 0.f) / H, gn = (g_norm ? *g_norm : 0.f) / H;
    {
        const int h = blockIdx.x;                      // one workgroup per head
        const float* ss = a + (long)h * N;
        const float* sc = ss + HN;
        const float* ms = ss + 2 * HN;
        const float* mc = ss + 3 * HN;
        const float* r = rec + (long)h * ADAP_PM_REC;
        const float c1 = r[0], c2 = r[1], P = r[2], A = r[3], B = r[4], bb1 = r[7], bb2 = r[8];
        const float inv = 1.0f / sqrtf(A * B);
        // g_tgt = gd (-inv t + (P/A) inv tgt);  g_src = gd (-inv tgt + (P/B) inv t) 3 src^2
        float q1 = 0.f, q2 = 0.f;                    // <g_src, ms>, <g_tgt, mc>
        for (int n = t; n < N; n += EM_THREADS) {
            const float src = ss[n] - c1 * ms[n], tgt = sc[n] - c2 * mc[n], tt = src * src * src;
            const float gt = gd * (-inv * tt + (P / A) * inv * tgt);
            const float gs = gd * (-inv * tgt + (P / B) * inv * tt) * 3.0f * src * src;
            q1 += gs * ms[n];
            q2 += gt * mc[n];
        }
        q1 = em_block_sum(q1, red);
        q2 = em_block_sum(q2, red);
        const float s1 = q1 / bb1, s2 = q2 / bb2;
        const float sg_c = r[5] > 0.f ? 1.f : (r[5] < 0.f ? -1.f : 0.f), sg_s = r[6] > 0.f ? 1.f : (r[6] < 0.f ? -1.f : 0.f);
        const float nc = gn * sg_c / N, ns = gn * sg_s / N;
        for (int n = t; n < N; n += EM_THREADS) {
            const float x0 = ss[n], x1 = sc[n], x2 = ms[n], x3 = mc[n];
            const float src = x0 - c1 * x2, tgt = x1 - c2 * x3, tt = src * src * src;
            const float gt = gd * (-inv * tt + (P / A) * inv * tgt);
            const float gs = gd * (-inv * tgt + (P / B) * inv * tt) * 3.0f * src * src;
            da[(long)h * N + n] = gs - s1 * x2 + ns;
            da[HN + (long)h * N + n] = gt - s2 * x3 + nc;
            da[2 * HN + (long)h * N + n] = gs_mix * (-c1 * gs - s1 * (x0 - 2.0f * c1 * x2) - ns);
            da[3 * HN + (long)h * N + n] = gs_mix * (-c2 * gt - s2 * (x1 - 2.0f * c2 * x3) - nc);
        }
    }
}

extern "C" int adap_promptmix_attn_terms(const float* a, int H, int N, float gs_mix, float* rec, float* out,
                                         const float* g_delta, const float* g_norm, float* da, void* stream) {
    ADAP_REQUIRE(a && rec && (out || da), ADAP_ERR_SHAPE, "promptmix_attn_terms: null pointer");
    ADAP_REQUIRE(H >= 1 && N >= 1, ADAP_ERR_SHAPE, "promptmix_attn_terms: H %d N %d", H, N);
    if (out) {
        hipLaunchKernelGGL(pm_attnterms_fwd_kernel, dim3(H), dim3(EM_THREADS), 0, (hipStream_t)stream, a, H, N, rec);
        hipLaunchKernelGGL(pm_attnterms_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rec, H, out);
    } else {
        hipLaunchKernelGGL(pm_attnterms_bwd_kernel, dim3(H), dim3(EM_THREADS), 0, (hipStream_t)stream, a, H, N, rec, g_delta, g_norm,
                           gs_mix, da);
    }
    return adap_check_launch("promptmix_attn_terms");
}

// ---------------------------------------------------------------------------------------------------------------------
// convert_attn_to_spatial_weight (ldm/util.py:1718) for one instance whose score map already has the feature map's
// resolution: a_n = mean over heads; w_n = min(exp(-(a_n - mean) / max(std + 0.001, mean / 2)), 1) / mean(w)  (std unbiased;
// `reversed`: small where the subject attends).  Up to two sources; sw = their average (ddpm.py:3869-3872 uses (mix + subj) / 2).
// ---------------------------------------------------------------------------------------------------------------------
#define PM_SW_THREADS 1024
__device__ __forceinline__ float pm_block_sum16(float v, float* red) {          // 16 waves, combined in a fixed order
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PM_SW_THREADS / 64; ++k) s += red[k];
    return s;
}

__global__ __launch_bounds__(PM_SW_THREADS) void pm_spatial_weight_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                                          int H, int N, int reversed, float* __restrict__ sw) {
    __shared__ float red[PM_SW_THREADS / 64];
    const int t = threadIdx.x;
    const int nsrc = a1 ? 2 : 1;
    for (int s = 0; s < nsrc; ++s) {
        const float* a = s ? a1 : a0;
        auto val = [&](int n) {
            float v = 0.f;
            for (int h = 0; h < H; ++h) v += a[(long)h * N + n];
            return v / H;
        };
        float m = 0.f;
        for (int n = t; n < N; n += PM_SW_THREADS) m += val(n);
        m = pm_block_sum16(m, red) / N;
        float q = 0.f;
        for (int n = t; n < N; n += PM_SW_THREADS) { const float d = val(n) - m; q += d * d; }
        q = pm_block_sum16(q, red);
        const float sd = sqrtf(q / (N > 1 ? N - 1 : 1));
        const float den = fmaxf(sd + 0.001f, m * 0.5f);
        const float sgn = reversed ? -1.f : 1.f;
        float ws = 0.f;
        for (int n = t; n < N; n += PM_SW_THREADS) ws += fminf(expf(sgn * (val(n) - m) / den), 1.f);
        ws = pm_block_sum16(ws, red) / N;
        for (int n = t; n < N; n += PM_SW_THREADS) {
            const float w = fminf(expf(sgn * (val(n) - m) / den), 1.f) / ws / nsrc;
            sw[n] = s ? sw[n] + w : w;
        }
    }
}

extern "C" int adap_attn_spatial_weight(const float* a0, const float* a1, int H, int N, int reversed, float* sw, void* stream) {
    ADAP_REQUIRE(a0 && sw && H >= 1 && N >= 1, ADAP_ERR_SHAPE, "attn_spatial_weight: arguments");
    hipLaunchKernelGGL(pm_spatial_weight_kernel, dim3(1), dim3(PM_SW_THREADS), 0, (hipStream_t)stream, a0, a1, H, N, reversed, sw);
    return adap_check_launch("attn_spatial_weight");
}

// ---------------------------------------------------------------------------------------------------------------------
// The two background-suppression terms of calc_comp_fg_bg_preserve_loss (ddpm.py:4520-4545): masked means of the comp
// instances' positive subject scores under the elastic matching's `below` weights,
//   l_s = sum_{h,n} max(sc_a, 0) scb[n] / max(H sum scb, 1e-6),   l_m likewise with (mc_a, mcb), d mc_a scaled by gs_mix (0.02).
// a f32 [4][H][N] (blocks 1 and 3 are used), scb / mcb f32 [N].  col f32 [2][N] = per-token sums over heads (kept for the backward),
// out f32 [4] = (l_s, l_m, H sum scb, H sum mcb).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EM_THREADS) void bgs_fwd_kernel(const float* __restrict__ a, const float* __restrict__ scb,
                                                             const float* __restrict__ mcb, int H, int N, float* __restrict__ col,
                                                             float* __restrict__ out) {
    __shared__ float red[4];
    const int t = threadIdx.x;
    const long HN = (long)H * N;
    float s1 = 0.f, s2 = 0.f, w1 = 0.f, w2 = 0.f;
    for (int n = t; n < N; n += EM_THREADS) {
        float c1 = 0.f, c2 = 0.f;
        for (int h = 0; h < H; ++h) {
            c1 += fmaxf(a[HN + (long)h * N + n], 0.f);
            c2 += fmaxf(a[3 * HN + (long)h * N + n], 0.f);
        }
        col[n] = c1;
        col[N + n] = c2;
        s1 += c1 * scb[n]; s2 += c2 * mcb[n];
        w1 += scb[n]; w2 += mcb[n];
    }
    s1 = em_block_sum(s1, red); s2 = em_block_sum(s2, red);
    w1 = em_block_sum(w1, red) * H; w2 = em_block_sum(w2, red) * H;
    if (t == 0) {
        out[0] = s1 / fmaxf(w1, 1e-6f);
        out[1] = s2 / fmaxf(w2, 1e-6f);
        out[2] = w1;
        out[3] = w2;
    }
}

__global__ __launch_bounds__(EM_THREADS) void bgs_bwd_kernel(const float* __restrict__ a, const float* __restrict__ scb,
                                                             const float* __restrict__ mcb, int H, int N, const float* __restrict__ col,
                                                             const float* __restrict__ out, const float* __restrict__ g_s,
                                                             const float* __restrict__ g_m, float gs_mix, float* __restrict__ da,
                                                             float* __restrict__ dscb, float* __restrict__ dmcb) {
    const int n = blockIdx.x * EM_THREADS + threadIdx.x;
    if (n >= N) return;
    const long HN = (long)H * N;
    const float gs = g_s ? *g_s : 0.f, gm = g_m ? *g_m : 0.f;
    const float w1 = out[2], w2 = out[3];
    const float c1 = fmaxf(w1, 1e-6f), c2 = fmaxf(w2, 1e-6f);
    // d l / d weight[n] = colsum[n] / cnt - l H / cnt (the count's clamp passes the gradient where the raw count >= 1e-6)
    dscb[n] = gs * (col[n] / c1 - (w1 >= 1e-6f ? out[0] * H / c1 : 0.f));
    dmcb[n] = gm * (col[N + n] / c2 - (w2 >= 1e-6f ? out[1] * H / c2 : 0.f));
    const float k1 = gs * scb[n] / c1, k2 = gs_mix * gm * mcb[n] / c2;
    for (int h = 0; h < H; ++h) {
        const long e = (long)h * N + n;
        da[e] = 0.f;
        da[HN + e] = a[HN + e] >= 0.f ? k1 : 0.f;
        da[2 * HN + e] = 0.f;
        da[3 * HN + e] = a[3 * HN + e] >= 0.f ? k2 : 0.f;
    }
}

extern "C" int adap_bg_suppress(const float* a, const float* scb, const float* mcb, int H, int N, float gs_mix, float* col, float* out,
                                const float* g_s, const float* g_m, float* da, float* dscb, float* dmcb, void* stream) {
    ADAP_REQUIRE(a && scb && mcb && col && out, ADAP_ERR_SHAPE, "bg_suppress: null pointer");
    ADAP_REQUIRE(H >= 1 && N >= 1, ADAP_ERR_SHAPE, "bg_suppress: H %d N %d", H, N);
    if (!da) hipLaunchKernelGGL(bgs_fwd_kernel, dim3(1), dim3(EM_THREADS), 0, (hipStream_t)stream, a, scb, mcb, H, N, col, out);
    else {
        ADAP_REQUIRE(dscb && dmcb, ADAP_ERR_SHAPE, "bg_suppress: backward needs dscb and dmcb");
        hipLaunchKernelGGL(bgs_bwd_kernel, dim3((N + EM_THREADS - 1) / EM_THREADS), dim3(EM_THREADS), 0, (hipStream_t)stream, a, scb, mcb,
                           H, N, col, out, g_s, g_m, gs_mix, da, dscb, dmcb);
    }
    return adap_check_launch("bg_suppress");
}

// Fused Prodigy optimiser step + global gradient-norm clip over flat fp32 buffers (gfx950).
//
// Reference: ldm/prodigy.py:97-252 (Prodigy.step) and Lightning's clip_gradients(0.5, "norm") -> torch
// clip_grad_norm_ (ddpm.py:606-633).  The reference walks the parameter list twice and calls .item() twice per
// parameter (prodigy.py:179,189): every parameter costs two device->host syncs.  Here all trainable values live in
// one flat buffer, the D-adaptation state (d, d_max, d_numerator, k, ...) lives in device memory, and one optimiser
// step is five launches with no host sync at all:
//
//   sumsq partials -> clip finish (coef)  -> moments (+ dot / |s| partials) -> finish (new d) -> update
//
// All three streaming kernels are HBM-bound (bytes per element: 4, 36, 16); partial sums are fp64, combined in a
// fixed order by a single block, so a step is bit-reproducible run to run.
#include "common.h"
#include <math.h>

#define OPT_NBLK 1024          // blocks per streaming launch == fp64 partials per reduction slot
#define OPT_THREADS 256

// state[] layout (doubles, device memory; include/adaprompt_hip.h documents it for callers)
enum { ST_D = 0, ST_DMAX, ST_DNUM, ST_DDEN, ST_DHAT, ST_K, ST_CLIP, ST_GNORM, ST_SKIP, ST_DLR, ST_COUNT = 16 };

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// deterministic block sum of one double per thread (256 threads): wave shuffles, then lanes 0..3 of wave 0
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ double prodigy_dlr(const double* st, double lr, double b1, double b2, int use_bc) {
    const double d = st[ST_D], k = st[ST_K];
    double bc = 1.0;
    if (use_bc) bc = sqrt(1.0 - pow(b2, k + 1.0)) / (1.0 - pow(b1, k + 1.0));      // prodigy.py:124-127
    return d * lr * bc;                                                              // prodigy.py:129
}

__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const float4* __restrict__ g, long n4,
                                                                    const float* __restrict__ tail, int ntail,
                                                                    double* __restrict__ partials) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long)OPT_NBLK * OPT_THREADS) {
        float4 v = g[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < ntail) acc += (double)tail[threadIdx.x] * tail[threadIdx.x];
    double tot = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__device__ __forceinline__ double sum_partials(const double* partials, int n, double* sh) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += OPT_THREADS) acc += partials[i];
    return block_sum_f64(acc, sh);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_finish_kernel(const double* __restrict__ partials, double max_norm,
                                                                  double* __restrict__ st) {
    __shared__ double sh[4];
    double tot = sum_partials(partials, OPT_NBLK, sh);
    if (threadIdx.x == 0) {
        // clip_grad_norm_: total_norm and the coefficient are fp32 tensors; coef = clamp(max/(total + 1e-6), max=1)
        float total = (float)sqrt(tot);
        float coef = (float)max_norm / (total + 1e-6f);
        st[ST_GNORM] = total;
        st[ST_CLIP] = coef < 1.0f ? coef : 1.0f;
    }
}

struct ProdigyHyper {
    double lr, b1, b2, b3, d0, decay_coupled;
    int use_bc, safeguard;
};

__device__ __forceinline__ void moments_elem(float p, float p0, float g, float& m, float& v, float& s, float coef,
                                             float decay, float b1, float b2, float b3, float am, float av, float as,
                                             double& num, double& den) {
    float gg = g * coef;
    if (decay != 0.f) gg = fmaf(decay, p, gg);                    // coupled weight decay, prodigy.py:160-161
    num += (double)gg * (double)(p0 - p);                         // prodigy.py:179
    m = fmaf(gg, am, m * b1);                                     // prodigy.py:185
    v = fmaf(gg * gg, av, v * b2);                                // prodigy.py:186
    s = fmaf(gg, as, s * b3);                                     // prodigy.py:188-191
    den += (double)fabsf(s);                                      // prodigy.py:192
}

__global__ __launch_bounds__(OPT_THREADS) void prodigy_moments_kernel(
        const float* __restrict__ p, const float* __restrict__ p0, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, float* __restrict__ s, long n, const double* __restrict__ st, ProdigyHyper hp,
        double* __restrict__ part_num, double* __restrict__ part_den) {
    __shared__ double sh[4];
    const double d = st[ST_D];
    const double dlr = prodigy_dlr(st, hp.lr, hp.b1, hp.b2, hp.use_bc);
    const float coef = (float)st[ST_CLIP];
    const float b1 = (float)hp.b1, b2 = (float)hp.b2, b3 = (float)hp.b3, decay = (float)hp.decay_coupled;
    const float am = (float)(d * (1.0 - hp.b1)), av = (float)(d * d * (1.0 - hp.b2));
    const float as = (float)((d / hp.d0) * (hp.safeguard ? d : dlr));
    double num = 0.0, den = 0.0;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long)OPT_NBLK * OPT_THREADS) {
        float4 P = ((const float4*)p)[i], P0 = ((const float4*)p0)[i], G = ((const float4*)g)[i];
        float4 M = ((float4*)m)[i], V = ((float4*)v)[i], S = ((float4*)s)[i];
        moments_elem(P.x, P0.x, G.x, M.x, V.x, S.x, coef, decay, b1, b2, b3, am, av, as, num, den);
        moments_elem(P.y, P0.y, G.y, M.y, V.y, S.y, coef, decay, b1, b2, b3, am, av, as, num, den);
        moments_elem(P.z, P0.z, G.z, M.z, V.z, S.z, coef, decay, b1, b2, b3, am, av, as, num, den);
        moments_elem(P.w, P0.w, G.w, M.w, V.w, S.w, coef, decay, b1, b2, b3, am, av, as, num, den);
        ((float4*)m)[i] = M;
        ((float4*)v)[i] = V;
        ((float4*)s)[i] = S;
    }
    if (blockIdx.x == 0) {                                         // ragged tail (n % 4 elements)
        long i = (n4 << 2) + threadIdx.x;
        if (i < n) moments_elem(p[i], p0[i], g[i], m[i], v[i], s[i], coef, decay, b1, b2, b3, am, av, as, num, den);
    }
    double tn = block_sum_f64(num, sh);
    double td = block_sum_f64(den, sh);
    if (threadIdx.x == 0) {
        part_num[blockIdx.x] = tn;
        part_den[blockIdx.x] = td;
    }
}

__global__ __launch_bounds__(OPT_THREADS) void prodigy_finish_kernel(const double* __restrict__ partials, int nslots,
                                                                     double* __restrict__ st, ProdigyHyper hp,
                                                                     double d_coef, double growth_rate) {
    __shared__ double sh[4];
    // slot s: numerators at partials[(2s) * NBLK ...], denominators at partials[(2s + 1) * NBLK ...]
    double num = 0.0, den = 0.0;
    for (int sl = 0; sl < nslots; ++sl) {
        num += sum_partials(partials + (size_t)(2 * sl) * OPT_NBLK, OPT_NBLK, sh);
        den += sum_partials(partials + (size_t)(2 * sl + 1) * OPT_NBLK, OPT_NBLK, sh);
    }
    if (threadIdx.x != 0) return;
    double d = st[ST_D];
    const double dlr = prodigy_dlr(st, hp.lr, hp.b1, hp.b2, hp.use_bc);
    if (den == 0.0) {                                   // prodigy.py:200-201: nothing moves, k does not advance
        st[ST_SKIP] = 1.0;
        return;
    }
    const double d_numerator = st[ST_DNUM] * hp.b3 + (d / hp.d0) * dlr * num;     // prodigy.py:135-136, 179
    double d_max = st[ST_DMAX];
    const double d_hat = d_coef * d_numerator / den;                              // prodigy.py:215-219
    if (d == hp.d0) d = fmax(d, d_hat);
    d_max = fmax(d_max, d_hat);
    d = fmin(d_max, d * growth_rate);
    st[ST_D] = d;
    st[ST_DMAX] = d_max;
    st[ST_DNUM] = d_numerator;
    st[ST_DDEN] = den;
    st[ST_DHAT] = d_hat;
    st[ST_K] = st[ST_K] + 1.0;
    st[ST_SKIP] = 0.0;
    st[ST_DLR] = dlr;
}

__global__ __launch_bounds__(OPT_THREADS) void prodigy_update_kernel(float* __restrict__ p, const float* __restrict__ m,
                                                                     const float* __restrict__ v, long n,
                                                                     const double* __restrict__ st, double eps,
                                                                     double decay_decoupled) {
    if (st[ST_SKIP] != 0.0) return;
    const float deps = (float)(st[ST_D] * eps);                    // the NEW d (prodigy.py:239) ...
    const float ndlr = (float)(-st[ST_DLR]);                       // ... with the OLD d*lr (prodigy.py:247)
    const float wd = (float)(-decay_decoupled * st[ST_DLR]);       // prodigy.py:242-243
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * OPT_THREADS) {
        float4 P = ((float4*)p)[i], M = ((const float4*)m)[i], V = ((const float4*)v)[i];
        if (wd != 0.f) { P.x = fmaf(P.x, wd, P.x); P.y = fmaf(P.y, wd, P.y); P.z = fmaf(P.z, wd, P.z); P.w = fmaf(P.w, wd, P.w); }
        P.x = fmaf(ndlr, M.x / (sqrtf(V.x) + deps), P.x);
        P.y = fmaf(ndlr, M.y / (sqrtf(V.y) + deps), P.y);
        P.z = fmaf(ndlr, M.z / (sqrtf(V.z) + deps), P.z);
        P.w = fmaf(ndlr, M.w / (sqrtf(V.w) + deps), P.w);
        ((float4*)p)[i] = P;
    }
    if (blockIdx.x == 0) {
        long i = (n4 << 2) + threadIdx.x;
        if (i < n) {
            float P = p[i];
            if (wd != 0.f) P = fmaf(P, wd, P);
            p[i] = fmaf(ndlr, m[i] / (sqrtf(v[i]) + deps), P);
        }
    }
}

// ---- Adam family (torch.optim.AdamW / torch.optim.NAdam, the reference's other `optimizer_type`s, ddpm.py:5134-5142,
// 5188-5196).  One elementwise pass over a parameter group's range; every step-dependent scalar (bias corrections, NAdam's
// momentum schedule) depends on the step count only, so the host computes it in fp64 and no state crosses the PCIe bus:
//   g' = clip * g + wd_coupled * p;  p *= 1 - decay;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2
//   p -= (cg * g' + cm * m) / (sqrt(v * inv_bc2) + eps)
// AdamW: decay = lr * wd, cg = 0, cm = lr / (1 - b1^t).  NAdam: cg = lr (1 - mu_t) / (1 - prod mu), cm = lr mu_{t+1} / (1 -
// prod mu * mu_{t+1}).  28 bytes per element (read p, g, m, v; write p, m, v): HBM-bound.
struct AdamHyper { float b1, b2, omb1, omb2, eps, decay, wdc, inv_bc2, cg, cm; };   // omb = 1 - beta, rounded from fp64

__device__ __forceinline__ void adam_elem(float& P, float G, float& M, float& V, const AdamHyper& h, float clip) {
    G *= clip;
    if (h.wdc != 0.f) G = fmaf(h.wdc, P, G);
    if (h.decay != 0.f) P = fmaf(-h.decay, P, P);
    M = fmaf(h.b1, M, h.omb1 * G);
    V = fmaf(h.b2, V, h.omb2 * G * G);
    const float den = sqrtf(V * h.inv_bc2) + h.eps;
    P -= fmaf(h.cg, G, h.cm * M) / den;
}

__global__ __launch_bounds__(OPT_THREADS) void adam_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v, long n,
                                                                  const double* __restrict__ st, AdamHyper h) {
    const float clip = (float)st[ST_CLIP];
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = ((float4*)p)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
        const float4 G = ((const float4*)g)[i];
        adam_elem(P.x, G.x, M.x, V.x, h, clip);
        adam_elem(P.y, G.y, M.y, V.y, h, clip);
        adam_elem(P.z, G.z, M.z, V.z, h, clip);
        adam_elem(P.w, G.w, M.w, V.w, h, clip);
        ((float4*)p)[i] = P;
        ((float4*)m)[i] = M;
        ((float4*)v)[i] = V;
    }
    if (blockIdx.x == 0) {
        const long i = (n4 << 2) + threadIdx.x;
        if (i < n) adam_elem(p[i], g[i], m[i], v[i], h, clip);
    }
}

__global__ void optim_state_init_kernel(double* st, double d0) {
    if (threadIdx.x < ST_COUNT) st[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        st[ST_D] = d0;
        st[ST_DMAX] = d0;
        st[ST_DHAT] = d0;
        st[ST_CLIP] = 1.0;
    }
}

// ---------------------------------------------------------------------------------------------- C ABI
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" long adap_optim_workspace_doubles(int nslots) {
    if (nslots < 1) nslots = 1;
    return (long)OPT_NBLK * (1 + 2 * (long)nslots);
}

extern "C" int adap_prodigy_state_init(double* state, double d0, void* stream) {
    ADAP_REQUIRE(state && d0 > 0, ADAP_ERR_SHAPE, "prodigy_state_init: state is null or d0 <= 0");
    hipLaunchKernelGGL(optim_state_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, d0);
    return adap_check_launch("prodigy_state_init");
}

extern "C" int adap_grad_clip_coef(const float* g, long n, double max_norm, double* state, double* workspace,
                                   void* stream) {
    ADAP_REQUIRE(g && state && workspace && n >= 0, ADAP_ERR_SHAPE, "grad_clip_coef: null pointer or n < 0");
    ADAP_REQUIRE(aligned16(g), ADAP_ERR_ALIGN, "grad_clip_coef: g must be 16-byte aligned");
    ADAP_REQUIRE(max_norm > 0, ADAP_ERR_SHAPE, "grad_clip_coef: max_norm must be > 0");
    const long n4 = n >> 2;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(OPT_NBLK), dim3(OPT_THREADS), 0, (hipStream_t)stream,
                       (const float4*)g, n4, g + (n4 << 2), (int)(n & 3), workspace);
    hipLaunchKernelGGL(clip_finish_kernel, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream, workspace, max_norm,
                       state);
    return adap_check_launch("grad_clip_coef");
}

extern "C" int adap_prodigy_moments(const float* p, const float* p0, const float* g, float* m, float* v, float* s,
                                    long n, const double* state, double* workspace, int slot, double lr, double beta1,
                                    double beta2, double beta3, double d0, double weight_decay_coupled,
                                    int use_bias_correction, int safeguard_warmup, void* stream) {
    ADAP_REQUIRE(p && p0 && g && m && v && s && state && workspace, ADAP_ERR_SHAPE, "prodigy_moments: null pointer");
    ADAP_REQUIRE(n >= 0 && slot >= 0, ADAP_ERR_SHAPE, "prodigy_moments: n < 0 or slot < 0");
    ADAP_REQUIRE(aligned16(p) && aligned16(p0) && aligned16(g) && aligned16(m) && aligned16(v) && aligned16(s),
                 ADAP_ERR_ALIGN, "prodigy_moments: buffers must be 16-byte aligned");
    ADAP_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && d0 > 0, ADAP_ERR_SHAPE,
                 "prodigy_moments: betas must be in [0,1) and d0 > 0");
    ProdigyHyper hp{lr, beta1, beta2, beta3, d0, weight_decay_coupled, use_bias_correction, safeguard_warmup};
    double* part = workspace + (size_t)OPT_NBLK * (1 + 2 * (size_t)slot);
    hipLaunchKernelGGL(prodigy_moments_kernel, dim3(OPT_NBLK), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, p0, g, m,
                       v, s, n, state, hp, part, part + OPT_NBLK);
    return adap_check_launch("prodigy_moments");
}

extern "C" int adap_prodigy_finish(double* state, const double* workspace, int nslots, double lr, double beta1,
                                   double beta2, double beta3, double d0, double d_coef, double growth_rate,
                                   int use_bias_correction, void* stream) {
    ADAP_REQUIRE(state && workspace && nslots >= 1, ADAP_ERR_SHAPE, "prodigy_finish: null pointer or nslots < 1");
    ProdigyHyper hp{lr, beta1, beta2, beta3, d0, 0.0, use_bias_correction, 0};
    hipLaunchKernelGGL(prodigy_finish_kernel, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream,
                       workspace + OPT_NBLK, nslots, state, hp, d_coef, growth_rate);
    return adap_check_launch("prodigy_finish");
}

extern "C" int adap_prodigy_update(float* p, const float* m, const float* v, long n, const double* state, double eps,
                                   double weight_decay_decoupled, void* stream) {
    ADAP_REQUIRE(p && m && v && state && n >= 0, ADAP_ERR_SHAPE, "prodigy_update: null pointer or n < 0");
    ADAP_REQUIRE(aligned16(p) && aligned16(m) && aligned16(v), ADAP_ERR_ALIGN,
                 "prodigy_update: buffers must be 16-byte aligned");
    long blocks = ((n >> 2) + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prodigy_update_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, m,
                       v, n, state, eps, weight_decay_decoupled);
    return adap_check_launch("prodigy_update");
}

extern "C" int adap_adam_update(float* p, const float* g, float* m, float* v, long n, const double* state, double beta1,
                                double beta2, double eps, double decay, double weight_decay_coupled,
                                double inv_bias_correction2, double coef_grad, double coef_moment, void* stream) {
    ADAP_REQUIRE(p && g && m && v && state && n >= 0, ADAP_ERR_SHAPE, "adam_update: null pointer or n < 0");
    ADAP_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), ADAP_ERR_ALIGN,
                 "adam_update: buffers must be 16-byte aligned");
    ADAP_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && inv_bias_correction2 > 0,
                 ADAP_ERR_SHAPE, "adam_update: betas must be in [0,1), eps >= 0, inv_bias_correction2 > 0");
    AdamHyper h{(float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)decay, (float)weight_decay_coupled,
                (float)inv_bias_correction2, (float)coef_grad, (float)coef_moment};
    long blocks = ((n >> 2) + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, g, m, v,
                       n, state, h);
    return adap_check_launch("adam_update");
}

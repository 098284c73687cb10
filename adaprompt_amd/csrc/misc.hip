// Small HBM-bound kernels of the SD-1.5 hot path: GEGLU, time-embedding MLP (few-row linear), sinusoidal
// timestep embedding, q_sample, posterior sample, masked weighted MSE (+ gradient), channel concat,
// 2x2 sum-pool (adjoint of the nearest upsample), channel pad/cast, bf16 transpose, and the
// VAE mid-attention softmax with the post-softmax fg/bg hetero-pair zero fill.
#include "common.h"

#define GRID_CAP 4096
static inline int grid_for(long n, int per_block) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > GRID_CAP) g = GRID_CAP;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------
// GEGLU (attention.py:32-40): h = [a | gate] along channels; out = a * gelu(gate), exact (erf) GELU.
// h bf16 [rows][2*I] (output of ff.net.0.proj), out bf16 [rows][I] (operand of ff.net.2).
// ---------------------------------------------------------------------------------------------
// Phi(g) and phi(g) of the exact (erf) GELU.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 the
// results are rounded to) on v_rcp_f32 / v_exp_f32: libm's erff is ~40 VALU instructions with two divergent branches, and
// it -- not HBM -- was what bound the GEGLU kernels (25 us for 21 M elements at 64 x 64); exp(-g^2 / 2) is shared with phi.
__device__ __forceinline__ void gelu_cdf_pdf(float g, float& cdf, float& pdf) {
    const float x = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
    const float e = __expf(-x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f * (1.0f + copysignf(erf_abs, g));
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float g) {
    float c, d;
    gelu_cdf_pdf(g, c, d);
    return g * c;
}

__global__ __launch_bounds__(256) void geglu_fwd_kernel(const uint16_t* __restrict__ h, long ldh, uint16_t* __restrict__ out,
                                                        long ldo, long rows, int I) {
    const int ch_per_row = I / 8;
    const long total = rows * ch_per_row;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx / ch_per_row;
        int ch = (int)(idx - r * ch_per_row);
        uint4 av = *(const uint4*)(h + r * ldh + ch * 8);
        uint4 gv = *(const uint4*)(h + r * ldh + I + ch * 8);
        float a[8], g[8], o[8];
        unpack_bf16x8(av, a);
        unpack_bf16x8(gv, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[e] * gelu_f(g[e]);
        *(uint4*)(out + r * ldo + ch * 8) = pack_bf16x8(o);
    }
}

// ---------------------------------------------------------------------------------------------
// MLP activation of the CLIP vision encoder (HF `ACT2FN`: "quick_gelu" x * sigmoid(1.702 x) for the OpenAI checkpoints,
// "gelu" (erf) for the LAION ones): f32 [rows][cols] (the fc1 accumulator output) -> bf16 operand of fc2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long ldx, uint16_t* __restrict__ out, long ldo,
                                                      long rows, int cols, int kind) {
    const int ch_per_row = cols / 8;
    const long total = rows * ch_per_row;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx / ch_per_row;
        int ch = (int)(idx - r * ch_per_row);
        const float4 a = *(const float4*)(x + r * ldx + ch * 8);
        const float4 b = *(const float4*)(x + r * ldx + ch * 8 + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = kind == 0 ? v[e] / (1.0f + __expf(-1.702f * v[e])) : gelu_f(v[e]);
        *(uint4*)(out + r * ldo + ch * 8) = pack_bf16x8(o);
    }
}

extern "C" int adap_act_fwd(const float* x, long ldx, void* out, long ldo, long rows, int cols, int kind, void* stream) {
    ADAP_REQUIRE(x && out, ADAP_ERR_SHAPE, "act_fwd: null pointer");
    ADAP_REQUIRE(cols % 8 == 0 && ldx % 4 == 0 && ldo % 8 == 0 && ldx >= cols && ldo >= cols, ADAP_ERR_ALIGN, "act_fwd: alignment");
    ADAP_REQUIRE(kind == 0 || kind == 1, ADAP_ERR_UNSUPPORTED, "act_fwd: kind %d (0 = quick_gelu, 1 = gelu)", kind);
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (uint16_t*)out, ldo, rows, cols, kind);
    return adap_check_launch("act_fwd");
}

// ---------------------------------------------------------------------------------------------
// row gather of bf16 rows: dst[b][i] = src[b][idx[b][i]] (16 bytes per thread)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ src, long lds, const int* __restrict__ idx,
                                                          uint16_t* __restrict__ dst, long ldd, int rows_src, int rows_dst,
                                                          int cols, long total) {
    const int cpr = cols / 8;
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long row = t / cpr;
        const int ch = (int)(t - row * cpr);
        const long b = row / rows_dst;
        const int r = idx[row];
        *(uint4*)(dst + row * ldd + ch * 8) = *(const uint4*)(src + (b * rows_src + r) * lds + ch * 8);
    }
}

extern "C" int adap_gather_rows_bf16(const void* src, long lds, const int* idx, void* dst, long ldd, int B, int rows_src,
                                     int rows_dst, int cols, void* stream) {
    ADAP_REQUIRE(src && idx && dst, ADAP_ERR_SHAPE, "gather_rows: null pointer");
    ADAP_REQUIRE(cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds >= cols && ldd >= cols, ADAP_ERR_ALIGN, "gather_rows: alignment");
    ADAP_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, ADAP_ERR_ALIGN, "gather_rows: 16-byte base alignment");
    const long total = (long)B * rows_dst * (cols / 8);
    if (total == 0) return ADAP_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, lds,
                       idx, (uint16_t*)dst, ldd, rows_src, rows_dst, cols, total);
    return adap_check_launch("gather_rows");
}

extern "C" int adap_geglu_fwd(const void* h, long ldh, void* out, long ldo, long rows, int inner, void* stream) {
    ADAP_REQUIRE(h && out, ADAP_ERR_SHAPE, "geglu_fwd: null pointer");
    ADAP_REQUIRE(inner % 8 == 0 && ldh % 8 == 0 && ldo % 8 == 0 && ldh >= 2 * inner, ADAP_ERR_ALIGN, "geglu_fwd: alignment");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(rows * (inner / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)h, ldh, (uint16_t*)out, ldo, rows, inner);
    return adap_check_launch("geglu_fwd");
}

// dh = [dout * gelu(g) | dout * a * gelu'(g)]   (bf16 in, bf16 out: operand of the proj data-gradient)
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint16_t* __restrict__ dout, long lddo,
                                                        const uint16_t* __restrict__ h, long ldh, uint16_t* __restrict__ dh,
                                                        long lddh, long rows, int I) {
    const int ch_per_row = I / 8;
    const long total = rows * ch_per_row;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx / ch_per_row;
        int ch = (int)(idx - r * ch_per_row);
        uint4 dv = *(const uint4*)(dout + r * lddo + ch * 8);
        uint4 av = *(const uint4*)(h + r * ldh + ch * 8);
        uint4 gv = *(const uint4*)(h + r * ldh + I + ch * 8);
        float d[8], a[8], g[8], da[8], dg[8];
        unpack_bf16x8(dv, d);
        unpack_bf16x8(av, a);
        unpack_bf16x8(gv, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float cdf, pdf;
            gelu_cdf_pdf(g[e], cdf, pdf);
            da[e] = d[e] * g[e] * cdf;
            dg[e] = d[e] * a[e] * (cdf + g[e] * pdf);
        }
        *(uint4*)(dh + r * lddh + ch * 8) = pack_bf16x8(da);
        *(uint4*)(dh + r * lddh + I + ch * 8) = pack_bf16x8(dg);
    }
}

extern "C" int adap_geglu_bwd(const void* dout, long lddo, const void* h, long ldh, void* dh, long lddh, long rows,
                              int inner, void* stream) {
    ADAP_REQUIRE(dout && h && dh, ADAP_ERR_SHAPE, "geglu_bwd: null pointer");
    ADAP_REQUIRE(inner % 8 == 0 && ldh % 8 == 0 && lddo % 8 == 0 && lddh % 8 == 0, ADAP_ERR_ALIGN, "geglu_bwd: alignment");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * (inner / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dout, lddo, (const uint16_t*)h, ldh, (uint16_t*)dh, lddh, rows, inner);
    return adap_check_launch("geglu_bwd");
}

// ---------------------------------------------------------------------------------------------
// Few-row linear in exact f32: y[r][n] = bias[n] + sum_k act(x[r][k]) * w[n][k], R <= 8 rows.
// time_embed (openaimodel.py:518-522) and every ResBlock's emb_layers (SiLU -> Linear, :217-223);
// the weights of all 22 emb_layers can be concatenated along n and served by one launch.
// One wave per output column.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                           int R, int K, int N, int pre_silu, int post_silu) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        float4 wv = *(const float4*)(w + (size_t)n * K + k);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R) {
                float4 xv = *(const float4*)(x + (size_t)r * ldx + k);
                if (pre_silu) { xv.x = silu_f(xv.x); xv.y = silu_f(xv.y); xv.z = silu_f(xv.z); xv.w = silu_f(xv.w); }
                acc[r] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r < R) {
            float s = wave_sum(acc[r]);
            if (lane == 0) {
                s += bias ? bias[n] : 0.f;
                if (post_silu) s = silu_f(s);
                y[(size_t)r * ldy + n] = s;
            }
        }
    }
}

extern "C" int adap_linear_small(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                                 int R, int K, int N, int pre_silu, int post_silu, void* stream) {
    ADAP_REQUIRE(x && w && y, ADAP_ERR_SHAPE, "linear_small: null pointer");
    ADAP_REQUIRE(R >= 1 && R <= 8, ADAP_ERR_UNSUPPORTED, "linear_small: R=%d (1..8)", R);
    ADAP_REQUIRE(K % 4 == 0 && ldx % 4 == 0, ADAP_ERR_ALIGN, "linear_small: K/ldx alignment");
    hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias, y, ldy, R, K,
                       N, pre_silu, post_silu);
    return adap_check_launch("linear_small");
}

// ---------------------------------------------------------------------------------------------
// timestep_embedding (util.py:154-174): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half)
// ---------------------------------------------------------------------------------------------
__global__ void timestep_embedding_kernel(const long long* __restrict__ t, float* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * half; idx += gridDim.x * blockDim.x) {
        int b = idx / half, i = idx - b * half;
        float f = expf(-9.210340371976184f * (float)i / (float)half);
        float a = (float)t[b] * f;
        out[(size_t)b * dim + i] = cosf(a);
        out[(size_t)b * dim + half + i] = sinf(a);
        if ((dim & 1) && i == 0) out[(size_t)b * dim + dim - 1] = 0.f;
    }
}

extern "C" int adap_timestep_embedding(const long long* t, float* out, int B, int dim, void* stream) {
    ADAP_REQUIRE(t && out && B > 0 && dim >= 2, ADAP_ERR_SHAPE, "timestep_embedding: bad args");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((long)B * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t,
                       out, B, dim);
    return adap_check_launch("timestep_embedding");
}

// ---------------------------------------------------------------------------------------------
// q_sample (ddpm.py:416-419): out = sqrt_ac[t_b] * x0 + sqrt_1mac[t_b] * noise      (any layout, per-sample scalars)
// ---------------------------------------------------------------------------------------------
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const long long* __restrict__ t,
                                const float* __restrict__ sa, const float* __restrict__ sb, float* __restrict__ out, int B,
                                long per, int T) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < B * per; idx += (long)gridDim.x * blockDim.x) {
        int b = (int)(idx / per);
        // a timestep outside the schedule must not become an out-of-bounds table read (a GPU fault): the sample
        // turns into NaN instead, which no downstream check can miss (the reference raises an index error here)
        const long long tt = t[b];
        const bool ok = tt >= 0 && tt < T;
        const float a = ok ? sa[tt] : __builtin_nanf(""), c = ok ? sb[tt] : __builtin_nanf("");
        out[idx] = a * x0[idx] + c * noise[idx];
    }
}

extern "C" int adap_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ac,
                             const float* sqrt_1mac, float* out, int B, long per_sample, int num_timesteps, void* stream) {
    ADAP_REQUIRE(x0 && noise && t && sqrt_ac && sqrt_1mac && out, ADAP_ERR_SHAPE, "q_sample: null pointer");
    ADAP_REQUIRE(num_timesteps > 0, ADAP_ERR_SHAPE, "q_sample: num_timesteps must be the length of the schedule tables");
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(B * per_sample, 256)), dim3(256), 0, (hipStream_t)stream, x0, noise, t,
                       sqrt_ac, sqrt_1mac, out, B, per_sample, num_timesteps);
    return adap_check_launch("q_sample");
}

// ---------------------------------------------------------------------------------------------
// posterior sample + scale (distributions.py:24-37, ddpm.py:955-962), pixel-major moments [P][2*Z]:
//   z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise)
// ---------------------------------------------------------------------------------------------
__global__ void posterior_sample_kernel(const float* __restrict__ moments, long ldm, const float* __restrict__ noise,
                                        float* __restrict__ z, long P, int Z, float scale) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < P * Z; idx += (long)gridDim.x * blockDim.x) {
        long px = idx / Z;
        int c = (int)(idx - px * Z);
        float mean = moments[px * ldm + c];
        float lv = fminf(fmaxf(moments[px * ldm + Z + c], -30.f), 20.f);
        z[idx] = scale * (mean + expf(0.5f * lv) * noise[idx]);
    }
}

extern "C" int adap_posterior_sample(const float* moments, long ldm, const float* noise, float* z, long pixels, int zch,
                                     float scale, void* stream) {
    ADAP_REQUIRE(moments && noise && z, ADAP_ERR_SHAPE, "posterior_sample: null pointer");
    hipLaunchKernelGGL(posterior_sample_kernel, dim3(grid_for(pixels * zch, 256)), dim3(256), 0, (hipStream_t)stream, moments,
                       ldm, noise, z, pixels, zch, scale);
    return adap_check_launch("posterior_sample");
}

// ---------------------------------------------------------------------------------------------
// calc_recon_loss (ddpm.py:3571-3595), pixel-major [P][C] tensors, masks [P]:
//   pix = ((out - tgt) * im)^2 ; w = fg*im*w_fg + (1-fg)*im*w_bg
//   loss = sum(pix * w) / (C * sum(w) + 1e-6) ;  grad = 2 (out - tgt) im^2 w / den
// One workgroup (the tensors are [4*64*64][4]): deterministic two-phase reduce.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void masked_mse_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                          const float* __restrict__ img_mask, const float* __restrict__ fg_mask,
                                                          float w_fg, float w_bg, long P, int C, float* __restrict__ loss,
                                                          float* __restrict__ grad) {
    __shared__ double snum[16], sden[16];
    __shared__ float sinv;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double num = 0.0, den = 0.0;
    for (long px = tid; px < P; px += 1024) {
        float im = img_mask ? img_mask[px] : 1.f;
        float fg = fg_mask ? fg_mask[px] : 1.f;
        float w = fg * im * w_fg + (1.f - fg) * im * w_bg;
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            float d = (out[px * C + c] - tgt[px * C + c]) * im;
            s += d * d;
        }
        num += (double)s * w;
        den += (double)w * C;
    }
    for (int o = 32; o > 0; o >>= 1) {
        num += __shfl_xor(num, o, 64);
        den += __shfl_xor(den, o, 64);
    }
    if (lane == 0) { snum[wv] = num; sden[wv] = den; }
    __syncthreads();
    if (tid == 0) {
        double n = 0.0, d = 0.0;
        for (int i = 0; i < 16; ++i) { n += snum[i]; d += sden[i]; }
        d += 1e-6;
        loss[0] = (float)(n / d);
        sinv = (float)(1.0 / d);
    }
    __syncthreads();
    if (grad) {
        const float inv = sinv;
        for (long px = tid; px < P; px += 1024) {
            float im = img_mask ? img_mask[px] : 1.f;
            float fg = fg_mask ? fg_mask[px] : 1.f;
            float w = fg * im * w_fg + (1.f - fg) * im * w_bg;
            float k = 2.f * im * im * w * inv;
            for (int c = 0; c < C; ++c) grad[px * C + c] = k * (out[px * C + c] - tgt[px * C + c]);
        }
    }
}

extern "C" int adap_masked_mse(const float* out, const float* tgt, const float* img_mask, const float* fg_mask, float w_fg,
                               float w_bg, long pixels, int C, float* loss, float* grad, void* stream) {
    ADAP_REQUIRE(out && tgt && loss, ADAP_ERR_SHAPE, "masked_mse: null pointer");
    hipLaunchKernelGGL(masked_mse_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, out, tgt, img_mask, fg_mask, w_fg, w_bg,
                       pixels, C, loss, grad);
    return adap_check_launch("masked_mse");
}

// ---------------------------------------------------------------------------------------------
// channel concat of two pixel-major f32 tensors (torch.cat([h, hs.pop()], dim=1), openaimodel.py:1018)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, long lda, int Ca, const float* __restrict__ b,
                                                      long ldb, int Cb, float* __restrict__ out, long ldo, long rows) {
    const int qa = Ca / 4, qb = Cb / 4, qt = qa + qb;
    const long total = rows * qt;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx / qt;
        int q = (int)(idx - r * qt);
        float4 v = (q < qa) ? *(const float4*)(a + r * lda + 4 * q) : *(const float4*)(b + r * ldb + 4 * (q - qa));
        *(float4*)(out + r * ldo + 4 * q) = v;
    }
}

extern "C" int adap_concat2(const float* a, long lda, int Ca, const float* b, long ldb, int Cb, float* out, long ldo,
                            long rows, void* stream) {
    ADAP_REQUIRE(a && b && out, ADAP_ERR_SHAPE, "concat2: null pointer");
    ADAP_REQUIRE(Ca % 4 == 0 && Cb % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0, ADAP_ERR_ALIGN, "concat2: alignment");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(concat2_kernel, dim3(grid_for(rows * ((Ca + Cb) / 4), 256)), dim3(256), 0, (hipStream_t)stream, a, lda, Ca,
                       b, ldb, Cb, out, ldo, rows);
    return adap_check_launch("concat2");
}

// ---------------------------------------------------------------------------------------------
// 2x2 sum pool, pixel-major f32: adjoint of F.interpolate(scale_factor=2, mode='nearest')
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                       int C) {
    const int q = C / 4;
    const long total = (long)B * H * W * q;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int cq = (int)(idx % q);
        long px = idx / q;
        int x = (int)(px % W);
        long t = px / W;
        int y = (int)(t % H);
        int b = (int)(t / H);
        const float* base = in + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + 4 * cq;
        float4 v0 = *(const float4*)(base), v1 = *(const float4*)(base + C);
        float4 v2 = *(const float4*)(base + (size_t)2 * W * C), v3 = *(const float4*)(base + (size_t)2 * W * C + C);
        *(float4*)(out + px * C + 4 * cq) =
            make_float4(v0.x + v1.x + v2.x + v3.x, v0.y + v1.y + v2.y + v3.y, v0.z + v1.z + v2.z + v3.z, v0.w + v1.w + v2.w + v3.w);
    }
}

extern "C" int adap_sumpool2x2(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    ADAP_REQUIRE(in && out && C % 4 == 0, ADAP_ERR_SHAPE, "sumpool2x2: bad args");
    hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_for((long)B * H * W * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                       B, H, W, C);
    return adap_check_launch("sumpool2x2");
}

// ---------------------------------------------------------------------------------------------
// f32 [rows][Cin] -> bf16 [rows][Cout>=Cin] with zero padding (image 3->8, latent 4->8 channels) / plain cast
// ---------------------------------------------------------------------------------------------
__global__ void pad_cast_kernel(const float* __restrict__ in, long ldi, int Cin, uint16_t* __restrict__ out, long ldo, int Cout,
                                long rows) {
    const long total = rows * Cout;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx / Cout;
        int c = (int)(idx - r * Cout);
        out[r * ldo + c] = c < Cin ? f32_to_bf16(in[r * ldi + c]) : (uint16_t)0;
    }
}

extern "C" int adap_pad_cast_bf16(const float* in, long ldi, int Cin, void* out, long ldo, int Cout, long rows, void* stream) {
    ADAP_REQUIRE(in && out && Cout >= Cin, ADAP_ERR_SHAPE, "pad_cast: bad args");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(pad_cast_kernel, dim3(grid_for(rows * Cout, 256)), dim3(256), 0, (hipStream_t)stream, in, ldi, Cin,
                       (uint16_t*)out, ldo, Cout, rows);
    return adap_check_launch("pad_cast");
}

// ---------------------------------------------------------------------------------------------
// nearest-neighbour x2 upsample with the cast to bf16: f32 [B][H][W][C] -> bf16 [B][2H][2W][C] (Upsample, openaimodel.py:95-123:
// F.interpolate(scale_factor=2, mode="nearest") in front of its conv3x3).  The conv then reads a plain bf16 image and takes
// the stencil-window kernel instead of the register-staged gather variant.  One thread = 8 channels of one SOURCE pixel: two
// 16-byte loads, four 16-byte stores.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2x_bf16_kernel(const float* __restrict__ in, long ldi, uint16_t* __restrict__ out, int B,
                                                              int H, int W, int C) {
    const int octs = C >> 3;
    const long total = (long)B * H * W * octs;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int o = (int)(idx % octs);
        long pxl = idx / octs;
        const int xx = (int)(pxl % W);
        pxl /= W;
        const int yy = (int)(pxl % H);
        const int b = (int)(pxl / H);
        const float4* src = (const float4*)(in + ((size_t)(b * H + yy) * W + xx) * ldi + 8 * o);
        const float4 a = src[0], c = src[1];
        float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        const uint4 q = pack_bf16x8(v);
        uint16_t* dst = out + (((size_t)b * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + 8 * o;
        *(uint4*)dst = q;
        *(uint4*)(dst + C) = q;
        *(uint4*)(dst + (size_t)2 * W * C) = q;
        *(uint4*)(dst + (size_t)2 * W * C + C) = q;
    }
}

extern "C" int adap_upsample2x_bf16(const float* in, long ldi, void* out, int B, int H, int W, int C, void* stream) {
    ADAP_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0, ADAP_ERR_SHAPE, "upsample2x_bf16: bad args");
    ADAP_REQUIRE(C % 8 == 0 && ldi % 4 == 0 && ldi >= C && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, ADAP_ERR_ALIGN,
                 "upsample2x_bf16: C=%d ldi=%ld", C, ldi);
    hipLaunchKernelGGL(upsample2x_bf16_kernel, dim3(grid_for((long)B * H * W * (C >> 3), 256)), dim3(256), 0, (hipStream_t)stream, in,
                       ldi, (uint16_t*)out, B, H, W, C);
    return adap_check_launch("upsample2x_bf16");
}

// ---------------------------------------------------------------------------------------------
// batched bf16 transpose [R][C] -> [C][R] through a 64x64 LDS tile (V^T for the VAE mid attention)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R,
                                                             int C) {
    __shared__ uint16_t tile[64][66];
    const size_t boff = (size_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < R && c0 + c < C) ? in[boff + (size_t)(r0 + r) * C + c0 + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int c = i >> 6, r = i & 63;
        if (r0 + r < R && c0 + c < C) out[boff + (size_t)(c0 + c) * R + r0 + r] = tile[r][c];
    }
}

extern "C" int adap_transpose_bf16(const void* in, void* out, int batch, int R, int C, void* stream) {
    ADAP_REQUIRE(in && out && batch > 0 && R > 0 && C > 0, ADAP_ERR_SHAPE, "transpose: bad args");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)in, (uint16_t*)out, R, C);
    return adap_check_launch("transpose_bf16");
}

// ---------------------------------------------------------------------------------------------
// VAE mid AttnBlock softmax (model.py:190-232): P = softmax_j(S[i][j] * scale); when per-pixel classes
// are given (0 = outside aug mask, 1 = fg, 2 = bg) every pair with class_i != class_j or class 0 is zeroed
// AFTER the softmax (no renormalisation).  S f32 [rows][N] -> P bf16.  One workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_softmax_kernel(const float* __restrict__ S, long lds_, uint16_t* __restrict__ P,
                                                          long ldp, const uint8_t* __restrict__ cls, int N, int rows_per_batch,
                                                          float scale) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const int b = (int)(row / rows_per_batch);
    const int i = (int)(row - (long)b * rows_per_batch);
    const float* s = S + row * lds_;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mx = -INFINITY;
    for (int j = tid * 4; j < N; j += 1024) {
        float4 v = *(const float4*)(s + j);
        mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    __syncthreads();
    float sum = 0.f;
    for (int j = tid * 4; j < N; j += 1024) {
        float4 v = *(const float4*)(s + j);
        sum += __expf(v.x * scale - mx) + __expf(v.y * scale - mx) + __expf(v.z * scale - mx) + __expf(v.w * scale - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    const uint8_t* cb = cls ? cls + (size_t)b * N : nullptr;
    const int ci = cb ? cb[i] : 1;
    for (int j = tid * 4; j < N; j += 1024) {
        float4 v = *(const float4*)(s + j);
        float o[4] = {__expf(v.x * scale - mx) * inv, __expf(v.y * scale - mx) * inv, __expf(v.z * scale - mx) * inv,
                      __expf(v.w * scale - mx) * inv};
        if (cb) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ci == 0 || cb[j + e] != ci) o[e] = 0.f;
        }
        uint2 pk;
        pk.x = pack_bf16x2(o[0], o[1]);
        pk.y = pack_bf16x2(o[2], o[3]);
        *(uint2*)(P + row * ldp + j) = pk;
    }
}

extern "C" int adap_vae_softmax(const float* S, long lds_, void* P, long ldp, const uint8_t* pixel_class, long rows, int N,
                                int rows_per_batch, float scale, void* stream) {
    ADAP_REQUIRE(S && P && N % 4 == 0 && lds_ % 4 == 0 && ldp % 4 == 0, ADAP_ERR_ALIGN, "vae_softmax: bad args");
    ADAP_REQUIRE(scale > 0.f, ADAP_ERR_UNSUPPORTED, "vae_softmax: scale must be positive");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(vae_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, lds_, (uint16_t*)P, ldp,
                       pixel_class, N, rows_per_batch, scale);
    return adap_check_launch("vae_softmax");
}

// ---------------------------------------------------------------------------------------------
// y += x (f32), used for gradient accumulation where two branches meet
// ---------------------------------------------------------------------------------------------
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float a, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 xv = ((const float4*)x)[i], yv = ((float4*)y)[i];
        ((float4*)y)[i] = make_float4(yv.x + a * xv.x, yv.y + a * xv.y, yv.z + a * xv.z, yv.w + a * xv.w);
    }
}

extern "C" int adap_axpy(const float* x, float* y, float a, long n, void* stream) {
    ADAP_REQUIRE(x && y && n % 4 == 0, ADAP_ERR_ALIGN, "axpy: n must be a multiple of 4");
    if (n == 0) return ADAP_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, a, n / 4);
    return adap_check_launch("axpy");
}


// ---------------------------------------------------------------------------------------------
// y = a + b over [rows][C] with independent leading dimensions, written as f32 (packed) and, optionally, as the
// bf16 operand copy: where the gradients of the two consumers of a skip connection meet (openaimodel.py:1018 keeps
// every encoder activation for the decoder), replacing autograd's own add so that the sum arrives with its bf16
// copy for the next data-gradient contraction.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                   long ldb, float* __restrict__ y32, uint16_t* __restrict__ y16,
                                                   long rows, int C4) {
    const long total = rows * C4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C4;
        const int c = (int)(i - r * C4) * 4;
        float4 x = *(const float4*)(a + r * lda + c), z = *(const float4*)(b + r * ldb + c);
        float4 o = make_float4(x.x + z.x, x.y + z.y, x.z + z.z, x.w + z.w);
        *(float4*)(y32 + r * (long)C4 * 4 + c) = o;
        if (y16) {
            uint2 w;
            w.x = pack_bf16x2(o.x, o.y);
            w.y = pack_bf16x2(o.z, o.w);
            *(uint2*)(y16 + r * (long)C4 * 4 + c) = w;
        }
    }
}

extern "C" int adap_add2(const float* a, long lda, const float* b, long ldb, float* y32, void* y16, long rows, int C,
                         void* stream) {
    ADAP_REQUIRE(a && b && y32 && rows >= 0 && C > 0, ADAP_ERR_SHAPE, "add2: null pointer or empty");
    ADAP_REQUIRE(C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 &&
                     ((uintptr_t)y32 % 16) == 0 && (!y16 || ((uintptr_t)y16 % 8) == 0),
                 ADAP_ERR_ALIGN, "add2: C and leading dims must be multiples of 4, pointers 16-byte aligned");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(add2_kernel, dim3(grid_for(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                       y32, (uint16_t*)y16, rows, C / 4);
    return adap_check_launch("add2");
}

// ---------------------------------------------------------------------------------------------
// Row-wise demeaned cosine loss with a sign-preserving squared reference, forward and analytic backward
// (ldm/util.py:437-535 calc_ref_cosine_loss with exponent 2: the cross-layer attention consistency loss and the
// prompt-delta loss of the recon iteration).  Per row of x, r [R][D]:
//   xt = x - mean(x), rt = r - mean(r) (if demean);  t = rt * |rt|;  cos = <xt,t> / sqrt((|xt|^2 + 1e-12)(|t|^2 + 1e-12))
//   loss = 1 - cos (align) or max(0, cos) (repel)      -- F.cosine_embedding_loss with margin 0
// One workgroup per row; torch spends ~25 element-wise launches per call on this chain (and again in backward).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ r,
                                                          long ldr, const float* __restrict__ gl, float* __restrict__ loss,
                                                          float* __restrict__ dx, long lddx, float* __restrict__ dr, long lddr,
                                                          int D, int demean, int align, float ref_grad_scale, int expo) {
    __shared__ float red[4];
    // the reference's sign-preserving power r |r|^(e-1) and its derivative e |r|^(e-1), e in {1, 2, 3}
    auto spow = [expo](float v) { return expo == 1 ? v : (expo == 2 ? v * fabsf(v) : v * v * v); };
    auto dspow = [expo](float v) { return expo == 1 ? 1.0f : (expo == 2 ? 2.0f * fabsf(v) : 3.0f * v * v); };
    const long row = blockIdx.x;
    const float* xr = x + row * ldx;
    const float* rr = r + row * ldr;
    const int t = threadIdx.x;
    float sx = 0.f, sr = 0.f;
    if (demean) {
        for (int i = t; i < D; i += 256) { sx += xr[i]; sr += rr[i]; }
        sx = block_sum_256(sx, red) / D;
        sr = block_sum_256(sr, red) / D;
    }
    float P = 0.f, A = 0.f, Bq = 0.f;
    for (int i = t; i < D; i += 256) {
        const float xt = xr[i] - sx, rt = rr[i] - sr, tt = spow(rt);
        P += xt * tt; A += xt * xt; Bq += tt * tt;
    }
    P = block_sum_256(P, red);
    A = block_sum_256(A, red) + 1e-12f;
    Bq = block_sum_256(Bq, red) + 1e-12f;
    const float inv = 1.0f / sqrtf(A * Bq);
    const float c = P * inv;
    if (loss && t == 0) loss[row] = align ? 1.0f - c : fmaxf(c, 0.f);
    if (!dx && !dr) return;
    // d loss / d cos, times the incoming gradient of this row's loss
    float gc = gl[row] * (align ? -1.0f : (c > 0.f ? 1.0f : 0.f));
    // dcos/dxt = (t - (P/A) xt) * inv ;  dcos/dt = (xt - (P/B) t) * inv ;  dt/drt = 2 |rt|
    float mx = 0.f, mr = 0.f;
    if (demean) {
        for (int i = t; i < D; i += 256) {
            const float xt = xr[i] - sx, rt = rr[i] - sr, tt = spow(rt);
            mx += (tt - (P / A) * xt) * inv;
            mr += (xt - (P / Bq) * tt) * inv * dspow(rt);
        }
        mx = block_sum_256(mx, red) / D;
        mr = block_sum_256(mr, red) / D;
    }
    for (int i = t; i < D; i += 256) {
        const float xt = xr[i] - sx, rt = rr[i] - sr, tt = spow(rt);
        if (dx) dx[row * lddx + i] = gc * ((tt - (P / A) * xt) * inv - mx);
        if (dr) dr[row * lddr + i] = gc * ref_grad_scale * ((xt - (P / Bq) * tt) * inv * dspow(rt) - mr);
    }
}

// loss != NULL: forward (gl, dx, dr NULL).  dx / dr != NULL: backward for the rows' loss gradients gl [R].
extern "C" int adap_cosine_rows(const float* x, long ldx, const float* r, long ldr, const float* gl, float* loss, float* dx,
                                long lddx, float* dr, long lddr, long R, int D, int demean, int align, float ref_grad_scale,
                                int exponent, void* stream) {
    ADAP_REQUIRE(x && r && (loss || ((dx || dr) && gl)), ADAP_ERR_SHAPE, "cosine_rows: null pointer");
    ADAP_REQUIRE(exponent >= 1 && exponent <= 3, ADAP_ERR_UNSUPPORTED, "cosine_rows: exponent %d (1, 2 or 3)", exponent);
    ADAP_REQUIRE(R >= 0 && D >= 1 && ldx >= D && ldr >= D, ADAP_ERR_SHAPE, "cosine_rows: shape");
    if (R == 0) return ADAP_OK;
    hipLaunchKernelGGL(cosine_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, ldx, r, ldr, gl, loss, dx,
                       lddx, dr, lddr, D, demean, align, ref_grad_scale, exponent);
    return adap_check_launch("cosine_rows");
}

// ---------------------------------------------------------------------------------------------
// ortho_subtract (ldm/util.py:280): per row of a, b [R][D]  out = a - c b,  c = <a,b> / (<b,b> + 1e-6)  -- the component of a
// orthogonal to b.  Stage 2's prompt-mix and elastic-matching losses call it ~60 times per micro-batch; as torch expressions it
// is 8 element-wise / reduction launches forward and ~15 in autograd's backward.  One workgroup per row, one launch each way.
//   backward, g = d L / d out:   da = g - s b,   db = -c g - s (a - 2 c b),   s = <g,b> / (<b,b> + 1e-6)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ortho_rows_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                         long ldb, const float* __restrict__ g, long ldg, float* __restrict__ out,
                                                         long ldo, float* __restrict__ da, long ldda, float* __restrict__ db,
                                                         long lddb, int D) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const float* ar = a + row * lda;
    const float* br = b + row * ldb;
    const int t = threadIdx.x;
    float ab = 0.f, bb = 0.f, gb = 0.f;
    if (g) {
        const float* gr = g + row * ldg;
        for (int i = t; i < D; i += 256) { const float bv = br[i]; ab += ar[i] * bv; bb += bv * bv; gb += gr[i] * bv; }
        gb = block_sum_256(gb, red);
    } else {
        for (int i = t; i < D; i += 256) { const float bv = br[i]; ab += ar[i] * bv; bb += bv * bv; }
    }
    ab = block_sum_256(ab, red);
    bb = block_sum_256(bb, red) + 1e-6f;
    const float c = ab / bb;
    if (!g) {
        for (int i = t; i < D; i += 256) out[row * ldo + i] = ar[i] - c * br[i];
        return;
    }
    const float s = gb / bb;
    const float* gr = g + row * ldg;
    for (int i = t; i < D; i += 256) {
        const float bv = br[i], gv = gr[i];
        if (da) da[row * ldda + i] = gv - s * bv;
        if (db) db[row * lddb + i] = -c * gv - s * (ar[i] - 2.0f * c * bv);
    }
}

// g == NULL: forward, out <- a - c b.  g != NULL: backward for g = d L / d out: da and / or db.
extern "C" int adap_ortho_rows(const float* a, long lda, const float* b, long ldb, const float* g, long ldg, float* out, long ldo,
                               float* da, long ldda, float* db, long lddb, long R, int D, void* stream) {
    ADAP_REQUIRE(a && b && ((!g && out) || (g && (da || db))), ADAP_ERR_SHAPE, "ortho_rows: null pointer");
    ADAP_REQUIRE(R >= 0 && D >= 1 && lda >= D && ldb >= D, ADAP_ERR_SHAPE, "ortho_rows: shape");
    if (R == 0) return ADAP_OK;
    hipLaunchKernelGGL(ortho_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, g, ldg, out, ldo, da,
                       ldda, db, lddb, D);
    return adap_check_launch("ortho_rows");
}

// Few, very long rows (Stage 2's pooled feature maps: 4 rows of 72 000): one workgroup per row is 4 workgroups walking 72 000
// elements twice (118 us per call, ~76 calls per compositional micro-batch).  Two launches instead: every row is cut into S
// slices; stage 1 leaves (<a,b>, <b,b>, <g,b>) per slice, stage 2 sums a row's slices IN ORDER (every workgroup of the row the same
// way: bit-reproducible) and applies the result to its own slice.
#define ORTHO_SLICE 2048
__global__ __launch_bounds__(256) void ortho_rows_partial_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                                 long ldb, const float* __restrict__ g, long ldg,
                                                                 float* __restrict__ part, int D, int S) {
    __shared__ float red[4];
    const long row = blockIdx.y;
    const int sl = blockIdx.x, t = threadIdx.x;
    const int i0 = sl * ORTHO_SLICE, i1 = min(D, i0 + ORTHO_SLICE);
    const float* ar = a + row * lda;
    const float* br = b + row * ldb;
    const float* gr = g ? g + row * ldg : nullptr;
    float ab = 0.f, bb = 0.f, gb = 0.f;
    for (int i = i0 + t; i < i1; i += 256) {
        const float bv = br[i];
        ab += ar[i] * bv; bb += bv * bv;
        if (gr) gb += gr[i] * bv;
    }
    ab = block_sum_256(ab, red);
    bb = block_sum_256(bb, red);
    gb = block_sum_256(gb, red);
    if (t == 0) {
        float* p = part + (row * S + sl) * 3;
        p[0] = ab; p[1] = bb; p[2] = gb;
    }
}

__global__ __launch_bounds__(256) void ortho_rows_apply_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                               long ldb, const float* __restrict__ g, long ldg, float* __restrict__ out,
                                                               long ldo, float* __restrict__ da, long ldda, float* __restrict__ db,
                                                               long lddb, const float* __restrict__ part, int D, int S) {
    const long row = blockIdx.y;
    const int sl = blockIdx.x, t = threadIdx.x;
    float ab = 0.f, bb = 0.f, gb = 0.f;
    for (int k = 0; k < S; ++k) {                      // the same order in every workgroup of the row
        const float* p = part + (row * S + k) * 3;
        ab += p[0]; bb += p[1]; gb += p[2];
    }
    bb += 1e-6f;
    const float c = ab / bb, s = gb / bb;
    const int i0 = sl * ORTHO_SLICE, i1 = min(D, i0 + ORTHO_SLICE);
    const float* ar = a + row * lda;
    const float* br = b + row * ldb;
    if (!g) {
        for (int i = i0 + t; i < i1; i += 256) out[row * ldo + i] = ar[i] - c * br[i];
        return;
    }
    const float* gr = g + row * ldg;
    for (int i = i0 + t; i < i1; i += 256) {
        const float bv = br[i], gv = gr[i];
        if (da) da[row * ldda + i] = gv - s * bv;
        if (db) db[row * lddb + i] = -c * gv - s * (ar[i] - 2.0f * c * bv);
    }
}

extern "C" long adap_ortho_rows_workspace_floats(long R, int D) {
    return D >= 4 * ORTHO_SLICE ? R * ((D + ORTHO_SLICE - 1) / ORTHO_SLICE) * 3 : 0;
}

// adap_ortho_rows with a workspace of adap_ortho_rows_workspace_floats(R, D) floats: rows of >= 8192 elements are cut into
// 2048-element slices over two launches (NULL or a short row: the one-workgroup-per-row kernel)
extern "C" int adap_ortho_rows_ws(const float* a, long lda, const float* b, long ldb, const float* g, long ldg, float* out, long ldo,
                                  float* da, long ldda, float* db, long lddb, long R, int D, float* workspace, void* stream) {
    if (!workspace || D < 4 * ORTHO_SLICE || R > 65535)
        return adap_ortho_rows(a, lda, b, ldb, g, ldg, out, ldo, da, ldda, db, lddb, R, D, stream);
    ADAP_REQUIRE(a && b && ((!g && out) || (g && (da || db))), ADAP_ERR_SHAPE, "ortho_rows: null pointer");
    ADAP_REQUIRE(R >= 0 && lda >= D && ldb >= D, ADAP_ERR_SHAPE, "ortho_rows: shape");
    if (R == 0) return ADAP_OK;
    const int S = (D + ORTHO_SLICE - 1) / ORTHO_SLICE;
    hipLaunchKernelGGL(ortho_rows_partial_kernel, dim3(S, (unsigned)R), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, g, ldg,
                       workspace, D, S);
    hipLaunchKernelGGL(ortho_rows_apply_kernel, dim3(S, (unsigned)R), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, g, ldg, out,
                       ldo, da, ldda, db, lddb, workspace, D, S);
    return adap_check_launch("ortho_rows");
}

// ---------------------------------------------------------------------------------------------
// The four mask hinge terms of calc_fg_bg_complementary_loss (ddpm.py:4143-4238) for a stack of same-resolution
// layers, forward and analytic backward.  S = subject score map, G = background-token score map, [L][B][H][N]
// (read with an element stride: the two are columns of the token-map tensor), f = foreground mask [B][N] in {0,1}.
//   aS[l,b] = sum_{h,n} S f / max(H sum_n f, 1e-6)             (the forward value is unaffected by the 0.5 ScaleGrad)
//   aG[l,b] = sum_{h,n} G (1-f) / max(H sum_n (1-f), 1e-6)
//   x1 = S(1-f) + m - aS   x2 = G f + m - aG   x3 = G f + m3 - aS   x4 = S(1-f) + m - aG
//   out[j][l] = sum_{b,h,n} x_j [x_j > 0] iw_b / max(#{x_j > 0}, 1e-6)
// Kernel 1 (grid L*B): aS, aG.  Kernel 2 (grid L*B): per (l,b) the four weighted sums and positive counts.
// Finish (tiny): out[j][l] and the total counts.  Backward (grid L*B): recomputes x_j and applies
//   dS = (1-f) (c1 p1 + c4 p4) iw - 0.5 f / dS_den * iw (c1 P1 + c3 P3),   c_j = gout[j][l] / cnt_j[l],  P_j = #{x_j > 0} in (l,b)
//   dG = f (c2 p2 + c3 p3) iw - (1-f) / dG_den * iw (c2 P2 + c4 P4)
// ---------------------------------------------------------------------------------------------
struct HingeParams {
    const float* S; const float* G; long estride;      // element (l,b,h,n) at S[((l*B+b)*H+h)*N*estride + n*estride]
    const float* f;                                     // [B][N]
    const float* iw;                                    // [B] or NULL
    int L, B, H, N;
    float m, m3;
    float* avg;                                         // [L*B][2]   aS, aG
    float* part;                                        // [L*B][8]   sum_j (4), cnt_j (4)
};

__global__ __launch_bounds__(256) void hinge_avg_kernel(HingeParams p) {
    __shared__ float red[4];
    const int lb = blockIdx.x, b = lb % p.B, t = threadIdx.x;
    const long base = (long)lb * p.H * p.N * p.estride;
    float sS = 0.f, sG = 0.f, nf = 0.f;
    for (int i = t; i < p.H * p.N; i += 256) {
        const int n = i % p.N;
        const float f = p.f[(long)b * p.N + n];
        sS += p.S[base + (long)i * p.estride] * f;
        if (p.G) sG += p.G[base + (long)i * p.estride] * (1.f - f);
        nf += f;
    }
    sS = block_sum_256(sS, red);
    sG = block_sum_256(sG, red);
    nf = block_sum_256(nf, red);
    if (t == 0) {
        p.avg[2 * lb] = sS / fmaxf(nf, 1e-6f);
        p.avg[2 * lb + 1] = sG / fmaxf((float)p.H * p.N - nf, 1e-6f);
    }
}

__global__ __launch_bounds__(256) void hinge_sum_kernel(HingeParams p) {
    __shared__ float red[4];
    const int lb = blockIdx.x, b = lb % p.B, t = threadIdx.x;
    const long base = (long)lb * p.H * p.N * p.estride;
    const float aS = p.avg[2 * lb], aG = p.avg[2 * lb + 1];
    float s[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < p.H * p.N; i += 256) {
        const int n = i % p.N;
        const float f = p.f[(long)b * p.N + n];
        const float S = p.S[base + (long)i * p.estride], G = p.G ? p.G[base + (long)i * p.estride] : 0.f;
        float x[4] = {S * (1.f - f) + p.m - aS, G * f + p.m - aG, G * f + p.m3 - aS, S * (1.f - f) + p.m - aG};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x[j] > 0.f) { s[j] += x[j]; c[j] += 1.f; }
    }
    const float w = p.iw ? p.iw[b] : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sj = block_sum_256(s[j], red), cj = block_sum_256(c[j], red);
        if (t == 0) { p.part[8 * lb + j] = sj * w; p.part[8 * lb + 4 + j] = cj; }
    }
}

// out[j][l] = sum_b part_sum / max(sum_b part_cnt, 1e-6); cnt[j][l] = that denominator
__global__ void hinge_finish_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ cnt, int L, int B,
                                    int has_bg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * L) return;
    const int j = i / L, l = i - j * L;
    float s = 0.f, c = 0.f;
    for (int b = 0; b < B; ++b) { s += part[8 * (l * B + b) + j]; c += part[8 * (l * B + b) + 4 + j]; }
    c = fmaxf(c, 1e-6f);
    const bool live = has_bg || j == 0;
    out[i] = live ? s / c : 0.f;
    cnt[i] = c;
}

__global__ __launch_bounds__(256) void hinge_bwd_kernel(HingeParams p, const float* __restrict__ gout, const float* __restrict__ cnt,
                                                        float* __restrict__ dS, float* __restrict__ dG, long dstride) {
    const int lb = blockIdx.x, l = lb / p.B, b = lb % p.B, t = threadIdx.x;
    const long base = (long)lb * p.H * p.N * p.estride, dbase = (long)lb * p.H * p.N * dstride;
    const float aS = p.avg[2 * lb], aG = p.avg[2 * lb + 1];
    const float w = p.iw ? p.iw[b] : 1.f;
    float c[4], P[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = (p.G || j == 0) ? gout[j * p.L + l] / cnt[j * p.L + l] * w : 0.f;
        P[j] = p.part[8 * lb + 4 + j];
    }
    // denominators of the two masked means of this instance
    __shared__ float red[4];
    float nf = 0.f;
    for (int n = t; n < p.N; n += 256) nf += p.f[(long)b * p.N + n];
    nf = block_sum_256(nf, red) * p.H;
    const float denS = fmaxf(nf, 1e-6f), denG = fmaxf((float)p.H * p.N - nf, 1e-6f);
    const float viaS = 0.5f * (c[0] * P[0] + c[2] * P[2]) / denS;       // ScaleGrad(0.5) on the foreground part of S
    const float viaG = (c[1] * P[1] + c[3] * P[3]) / denG;
    for (int i = t; i < p.H * p.N; i += 256) {
        const int n = i % p.N;
        const float f = p.f[(long)b * p.N + n];
        const float S = p.S[base + (long)i * p.estride], G = p.G ? p.G[base + (long)i * p.estride] : 0.f;
        const float x0 = S * (1.f - f) + p.m - aS, x1 = G * f + p.m - aG, x2 = G * f + p.m3 - aS, x3 = S * (1.f - f) + p.m - aG;
        const float p0 = x0 > 0.f, p1 = x1 > 0.f, p2 = x2 > 0.f, p3 = x3 > 0.f;
        dS[dbase + (long)i * dstride] = (1.f - f) * (c[0] * p0 + c[3] * p3) - f * viaS;
        if (dG) dG[dbase + (long)i * dstride] = f * (c[1] * p1 + c[2] * p2) - (1.f - f) * viaG;
    }
}

// workspace floats: 2*L*B (avg) + 8*L*B (partials) + 4*L (counts)
extern "C" long adap_mask_hinges_workspace_floats(int L, int B) { return 10L * L * B + 4L * L; }

// forward: out f32 [4][L].  S / G: element stride `estride` floats (G may be NULL: only term 0).  workspace is read again
// by the backward (keep it): adap_mask_hinges_bwd(gout [4][L]) -> dS, dG with element stride `dstride`.
extern "C" int adap_mask_hinges_fwd(const float* S, const float* G, long estride, const float* fmask, const float* iw, float* out,
                                    float* workspace, int L, int B, int H, int N, float margin, float margin_bg_at_mf,
                                    void* stream) {
    ADAP_REQUIRE(S && fmask && out && workspace && L > 0 && B > 0 && H > 0 && N > 0 && estride >= 1, ADAP_ERR_SHAPE,
                 "mask_hinges_fwd: arguments");
    HingeParams p = {S, G, estride, fmask, iw, L, B, H, N, margin, margin_bg_at_mf, workspace, workspace + 2L * L * B};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hinge_avg_kernel, dim3(L * B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(hinge_sum_kernel, dim3(L * B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(hinge_finish_kernel, dim3((4 * L + 63) / 64), dim3(64), 0, s, p.part, out, workspace + 10L * L * B, L, B,
                       G != nullptr);
    return adap_check_launch("mask_hinges_fwd");
}

extern "C" int adap_mask_hinges_bwd(const float* S, const float* G, long estride, const float* fmask, const float* iw,
                                    const float* gout, const float* workspace, float* dS, float* dG, long dstride, int L, int B,
                                    int H, int N, float margin, float margin_bg_at_mf, void* stream) {
    ADAP_REQUIRE(S && fmask && gout && workspace && dS && (!G || dG) && dstride >= 1, ADAP_ERR_SHAPE, "mask_hinges_bwd: arguments");
    HingeParams p = {S, G, estride, fmask, iw, L, B, H, N, margin, margin_bg_at_mf, (float*)workspace,
                     (float*)workspace + 2L * L * B};
    hipLaunchKernelGGL(hinge_bwd_kernel, dim3(L * B), dim3(256), 0, (hipStream_t)stream, p, gout, workspace + 10L * L * B, dS, dG,
                       dstride);
    return adap_check_launch("mask_hinges_bwd");
}

// ---------------------------------------------------------------------------------------------
// Key masks of the UNet's levels in ONE launch (attention.py:223-232, :332: the image mask [B,1,h,w], nearest-resized to the
// level, != 0) and, for the levels whose self-attention runs on the kept keys alone, the stable partition behind it: perm
// lists a sample's kept keys first (in order), then the masked ones; inv undoes it; count = kept keys (N when none is kept:
// that sample attends to every key, functional.KeyCompaction).  One workgroup per (level, sample); the partition walks the
// row in chunks of 256 with a ballot prefix, so the order inside a class is the key order (= torch.argsort(stable=True)).
// The nearest source index is aten's: min(int(floorf(dst * (float)in / out)), in - 1).
// ---------------------------------------------------------------------------------------------
#define KEYMASK_MAX_LEVELS 8
struct KeyMaskLevels {
    const float* img;
    uint8_t* u8;
    int* i32;
    int B, h, w, nlev;
    int H[KEYMASK_MAX_LEVELS], W[KEYMASK_MAX_LEVELS];
    long mask_off[KEYMASK_MAX_LEVELS], perm_off[KEYMASK_MAX_LEVELS], inv_off[KEYMASK_MAX_LEVELS], count_off[KEYMASK_MAX_LEVELS];
};

__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

__global__ __launch_bounds__(256) void key_masks_kernel(KeyMaskLevels p) {
    __shared__ int wave_kept[4];
    __shared__ int red[4];
    const int lev = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = p.H[lev], W = p.W[lev], N = H * W;
    const float sh = (float)p.h / (float)H, sw = (float)p.w / (float)W;
    const float* src = p.img + (long)b * p.h * p.w;
    uint8_t* m = p.u8 + p.mask_off[lev] + (long)b * N;
    int cnt = 0;
    for (int i = tid; i < N; i += 256) {
        const int y = i / W, x = i - y * W;
        const uint8_t v = src[(long)nearest_src(y, sh, p.h) * p.w + nearest_src(x, sw, p.w)] != 0.f;
        m[i] = v;
        cnt += v;
    }
    if (p.perm_off[lev] < 0) return;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) red[wv] = cnt;
    __syncthreads();
    const int kept = red[0] + red[1] + red[2] + red[3];
    if (tid == 0) p.i32[p.count_off[lev] + b] = kept == 0 ? N : kept;
    int* perm = p.i32 + p.perm_off[lev] + (long)b * N;
    int* inv = p.i32 + p.inv_off[lev] + (long)b * N;
    int base_k = 0, base_m = kept;
    for (int c0 = 0; c0 < N; c0 += 256) {
        const int i = c0 + tid;
        const bool valid = i < N;
        const bool v = valid && m[i] != 0;          // (this thread's own store above: i = tid mod 256 in both loops)
        const unsigned long long bal = __ballot(v);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();                            // (the previous chunk's readers of wave_kept are done)
        if (lane == 0) wave_kept[wv] = __popcll(bal);
        __syncthreads();
        int kept_before = before, chunk_kept = 0;
        for (int q = 0; q < 4; ++q) {
            if (q < wv) kept_before += wave_kept[q];
            chunk_kept += wave_kept[q];
        }
        if (valid) {
            const int pos = v ? base_k + kept_before : base_m + (tid - kept_before);
            perm[pos] = i;
            inv[i] = pos;
        }
        const int chunk_n = N - c0 < 256 ? N - c0 : 256;
        base_k += chunk_kept;
        base_m += chunk_n - chunk_kept;
    }
}

// img f32 [B][h][w] (the mask's single channel); dims int [2 * nlev] = (H, W) per level; offs long [4 * nlev] = per level the
// byte offset of its mask [B][H*W] in out_u8 and the int32 offsets of perm [B][H*W], inv [B][H*W], count [B] in out_i32
// (perm offset < 0: the level gets its mask only).  dims / offs are host arrays.
extern "C" int adap_key_masks(const float* img, int B, int h, int w, int nlev, const int* dims, const long* offs, void* out_u8,
                              int* out_i32, void* stream) {
    ADAP_REQUIRE(img && dims && offs && out_u8 && B > 0 && h > 0 && w > 0, ADAP_ERR_SHAPE, "key_masks: arguments");
    ADAP_REQUIRE(nlev >= 1 && nlev <= KEYMASK_MAX_LEVELS, ADAP_ERR_UNSUPPORTED, "key_masks: %d levels (1 .. %d)", nlev, KEYMASK_MAX_LEVELS);
    KeyMaskLevels p = {};
    p.img = img; p.u8 = (uint8_t*)out_u8; p.i32 = out_i32; p.B = B; p.h = h; p.w = w; p.nlev = nlev;
    for (int l = 0; l < nlev; ++l) {
        p.H[l] = dims[2 * l]; p.W[l] = dims[2 * l + 1];
        p.mask_off[l] = offs[4 * l]; p.perm_off[l] = offs[4 * l + 1]; p.inv_off[l] = offs[4 * l + 2]; p.count_off[l] = offs[4 * l + 3];
        ADAP_REQUIRE(p.H[l] > 0 && p.W[l] > 0 && (long)p.H[l] * p.W[l] < (1L << 30) && p.mask_off[l] >= 0, ADAP_ERR_SHAPE,
                     "key_masks: level %d is %d x %d", l, p.H[l], p.W[l]);
        ADAP_REQUIRE(p.perm_off[l] < 0 || (out_i32 && p.inv_off[l] >= 0 && p.count_off[l] >= 0), ADAP_ERR_SHAPE,
                     "key_masks: level %d wants its compaction but has no int32 output", l);
    }
    hipLaunchKernelGGL(key_masks_kernel, dim3(nlev, B), dim3(256), 0, (hipStream_t)stream, p);
    return adap_check_launch("key_masks");
}

// ---------------------------------------------------------------------------------------------
// Pixel classes of the VAE's masked mid-block attention (model.py:196-232): fg / aug masks [B,1,h,w] nearest-resized to H x W;
// class 1 where fg * aug != 0, 2 where (1 - fg) * aug != 0, else 0 (aug NULL = all ones).  out u8 [B][H*W].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pixel_classes_kernel(const float* __restrict__ fg, const float* __restrict__ aug,
                                                            uint8_t* __restrict__ out, int B, int h, int w, int ha, int wa,
                                                            int H, int W) {
    const long total = (long)B * H * W;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W, sha = (float)ha / (float)H, swa = (float)wa / (float)W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const long b = i / ((long)W * H);
        const float f = fg[b * h * w + (long)nearest_src(y, sh, h) * w + nearest_src(x, sw, w)];
        const float a = aug ? aug[b * ha * wa + (long)nearest_src(y, sha, ha) * wa + nearest_src(x, swa, wa)] : 1.f;
        out[i] = (f * a != 0.f) ? 1 : (((1.f - f) * a != 0.f) ? 2 : 0);
    }
}

extern "C" int adap_pixel_classes(const float* fg, int h, int w, const float* aug, int ha, int wa, void* out, int B, int H, int W,
                                  void* stream) {
    ADAP_REQUIRE(fg && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && (!aug || (ha > 0 && wa > 0)), ADAP_ERR_SHAPE,
                 "pixel_classes: arguments");
    hipLaunchKernelGGL(pixel_classes_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, fg, aug,
                       (uint8_t*)out, B, h, w, ha, wa, H, W);
    return adap_check_launch("pixel_classes");
}

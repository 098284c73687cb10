// GroupNorm(32)(+SiLU) and LayerNorm, forward and data-gradient, for pixel-major (NHWC) tensors.
//
// HBM-bound kernels.  GroupNorm32 is computed in fp32 (util.py:217-219) over [HW x C/32] per
// (sample, group); eps 1e-5 in the UNet ResBlocks / out, 1e-6 in SpatialTransformer.norm and in
// all VAE norms (attention.py:71-72, model.py:39-40) -- eps is an argument.  In the pixel-major
// layout a group is a run of C/32 consecutive channels of every pixel, so a block streams a slab
// of whole pixel rows with fully coalesced accesses; every thread owns fixed 8-channel "octs"
// (32 B of f32 or 16 B of bf16 per access), group sums are combined through LDS in a fixed
// order, per-slab partials go to a small workspace ([B][nchunks][32][2] floats) and the apply
// pass finishes the reduction in fp64 -- no float atomics anywhere, results are bit-reproducible.
// The input may be f32 (the residual stream) or bf16 (block-internal tensors such as the output
// of a ResBlock's first conv, which only this norm and its backward ever read); the output is
// bf16 (operand of the following contraction) and/or f32.
//
// Algorithmic bytes per element: forward = read x once + write y
// (f32 -> bf16: 6 B, bf16 -> bf16: 4 B); the kernels read x twice (statistics, then apply), the
// second read is served by the 256 MB Infinity Cache for the UNet's tensors (<= 84 MB at bs=4).
#include "common.h"

#define GN_G 32
#define GN_MAXSLOT 2     // 8-channel octs per thread: C <= 4096
#define GN_UNR 4         // pixel rows in flight per thread (forward)
#define GN_UNRB 2        // ... in the backward (two input streams)

struct GnGeom {
    int Q;        // octs per pixel row (C/8)
    int TPR;      // threads per row
    int rpp;      // rows per pass
    int nslots;   // octs per thread
};

__device__ __forceinline__ GnGeom gn_geom(int C) {
    GnGeom g;
    g.Q = C >> 3;
    g.TPR = g.Q < 256 ? g.Q : 256;
    g.rpp = 256 / g.TPR;
    g.nslots = (g.Q + g.TPR - 1) / g.TPR;
    return g;
}

template <bool BF16>
__device__ __forceinline__ void gn_load8(const void* base, size_t off, float* o) {
    if (BF16) {
        uint4 v = *(const uint4*)((const uint16_t*)base + off);
        unpack_bf16x8(v, o);
    } else {
        const float4* p = (const float4*)((const float*)base + off);
        float4 a = p[0], b = p[1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
        o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
}

__device__ __forceinline__ void gn_store8_f32(float* p, const float* o) {
    *(float4*)p = make_float4(o[0], o[1], o[2], o[3]);
    *(float4*)(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

// Deterministic block reduction of per-thread channel sums into the 32 group sums, two levels, fixed order:
// every thread parks its (slot, e) partials in LDS [GN_MAXSLOT*8][256]; then each thread folds the rows-per-pass
// partials of one channel (all 256 threads busy), and thread g < 32 adds group g's channel sums.
__device__ __forceinline__ void gn_fold_channels(const float* sm, float* chs, const GnGeom& g, int C, int tid) {
    for (int c = tid; c < C; c += 256) {
        int q = c >> 3, e = c & 7;
        int slot = q / g.TPR, lir = q - slot * g.TPR;
        const float* col = sm + (slot * 8 + e) * 256 + lir;
        float acc = 0.f;
        for (int r0 = 0; r0 < g.rpp; ++r0) acc += col[r0 * g.TPR];
        chs[c] = acc;
    }
}

__device__ __forceinline__ float gn_group_sum(const float* chs, int cpg, int grp) {
    float acc = 0.f;
    for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) acc += chs[c];
    return acc;
}

// park -> fold -> group sum, with the barriers; returns group tid's sum in threads tid < 32 (others 0)
__device__ __forceinline__ float gn_block_reduce(float* sm, float* ch, const float (*v)[8], const GnGeom& g, int C, int cpg,
                                                 int tid);

// sum of the per-slab partials of (b, group) in fp64, 8 threads per group (tid>>3 = group), fixed order
__device__ __forceinline__ void gn_finish_partials(const float* partial, int b, int nchunks, int tid, double* su, double* sq) {
    const int grp = tid >> 3, part = tid & 7;
    double a = 0.0, c2 = 0.0;
    const float* pp = partial + ((size_t)b * nchunks * GN_G + grp) * 2;
    for (int c = part; c < nchunks; c += 8) {
        float2 v = *(const float2*)(pp + (size_t)c * GN_G * 2);
        a += v.x; c2 += v.y;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 8);
        c2 += __shfl_down(c2, o, 8);
    }
    *su = a;
    *sq = c2;
}

__device__ __forceinline__ void gn_park(float* sm, const float (*v)[8], int tid) {
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[(k * 8 + e) * 256 + tid] = v[k][e];
}

__device__ __forceinline__ float gn_block_reduce(float* sm, float* ch, const float (*v)[8], const GnGeom& g, int C, int cpg,
                                                 int tid) {
    gn_park(sm, v, tid);
    __syncthreads();
    gn_fold_channels(sm, ch, g, C, tid);
    __syncthreads();
    float r = tid < GN_G ? gn_group_sum(ch, cpg, tid) : 0.f;
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
// forward pass 1: per-slab partial sums.  grid (nchunks, B), block 256.
// ---------------------------------------------------------------------------------------------
template <bool XB>
__global__ __launch_bounds__(256) void gn_stats_kernel(const void* __restrict__ x, long ldx, int HW, int C,
                                                       int rows_per_chunk, float* __restrict__ partial) {
    __shared__ float smA[GN_MAXSLOT * 8 * 256], chA[GN_MAXSLOT * 8 * 256];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const GnGeom g = gn_geom(C);
    const int cpg = C / GN_G;
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    const int row_begin = chunk * rows_per_chunk;
    const int row_end = min(HW, row_begin + rows_per_chunk);
    float s[GN_MAXSLOT][8], ss[GN_MAXSLOT][8];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[k][e] = 0.f; ss[k][e] = 0.f; }
    if (r0 < g.rpp) {
        const size_t boff = (size_t)b * HW;
        // GN_UNR rows in flight per thread: a pure HBM stream, memory-level parallelism is the lever
        for (int r = row_begin + r0; r < row_end; r += g.rpp * GN_UNR) {
            float v[GN_UNR][GN_MAXSLOT][8];
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u) {
                const int rr = r + u * g.rpp;
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k) {
                    int q = lir + k * g.TPR;
                    if (rr < row_end && k < g.nslots && q < g.Q) gn_load8<XB>(x, (boff + rr) * ldx + 8 * q, v[u][k]);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[u][k][e] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u)
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s[k][e] += v[u][k][e];
                        ss[k][e] += v[u][k][e] * v[u][k][e];
                    }
        }
    }
    const float ra = gn_block_reduce(smA, chA, s, g, C, cpg, tid);
    const float rb = gn_block_reduce(smA, chA, ss, g, C, cpg, tid);
    if (tid < GN_G) *(float2*)(partial + (((size_t)b * nchunks + chunk) * GN_G + tid) * 2) = make_float2(ra, rb);
}

// ---------------------------------------------------------------------------------------------
// forward pass 2: finish the reduction, normalise, affine, optional SiLU, write bf16 / f32.
// ---------------------------------------------------------------------------------------------
template <bool XB>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ x, long ldx, int HW, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ partial, int stats_chunks, float eps, int act,
                                                       int rows_per_chunk, float* __restrict__ y32, long ldy32,
                                                       uint16_t* __restrict__ y16, long ldy16,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ float lmean[GN_G], lrstd[GN_G];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cpg = C / GN_G;
    if (stats_chunks == 0) {                    // statistics already finished (gn_finish_kernel): mean_out / rstd_out are inputs
        if (tid < GN_G) {
            lmean[tid] = mean_out[b * GN_G + tid];
            lrstd[tid] = rstd_out[b * GN_G + tid];
        }
    } else {
        double su, sq;
        gn_finish_partials(partial, b, stats_chunks, tid, &su, &sq);
        if ((tid & 7) == 0) {
            const int grp = tid >> 3;
            double n = (double)cpg * HW;
            double mean = su / n;
            double var = sq / n - mean * mean;
            if (var < 0.0) var = 0.0;
            float rstd = (float)(1.0 / sqrt(var + (double)eps));
            lmean[grp] = (float)mean;
            lrstd[grp] = rstd;
            if (chunk == 0) {
                mean_out[b * GN_G + grp] = (float)mean;
                rstd_out[b * GN_G + grp] = rstd;
            }
        }
    }
    __syncthreads();
    const GnGeom g = gn_geom(C);
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    if (r0 >= g.rpp) return;
    float sc[GN_MAXSLOT][8], sh[GN_MAXSLOT][8];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k) {
        int q = lir + k * g.TPR;
        if (k < g.nslots && q < g.Q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int c = 8 * q + e, grp = c / cpg;
                float a = lrstd[grp] * gamma[c];
                sc[k][e] = a;
                sh[k][e] = beta[c] - lmean[grp] * a;
            }
        }
    }
    const int row_begin = chunk * rows_per_chunk;
    const int row_end = min(HW, row_begin + rows_per_chunk);
    const size_t boff = (size_t)b * HW;
    for (int r = row_begin + r0; r < row_end; r += g.rpp * GN_UNR) {
        float v[GN_UNR][GN_MAXSLOT][8];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) gn_load8<XB>(x, (boff + rr) * ldx + 8 * q, v[u][k]);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o[e] = v[u][k][e] * sc[k][e] + sh[k][e];
                        if (act) o[e] = silu_f(o[e]);
                    }
                    if (y32) gn_store8_f32(y32 + (boff + rr) * ldy32 + 8 * q, o);
                    if (y16) *(uint4*)(y16 + (boff + rr) * ldy16 + 8 * q) = pack_bf16x8(o);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Statistics that came out of the producing contraction's epilogue (adap_conv2d_next_gn_partial): up to 1024 tile records
// per sample.  Finished ONCE here -- 16 threads per (group, statistic) walk the records in a fixed order in fp64 -- instead
// of by every apply workgroup (256 of them would each re-read 256 KB).  grid B, block 1024.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gn_finish_kernel(const float* __restrict__ partial, int nchunks, double count, float eps,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double acc[16][64];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int v = tid & 63, part = tid >> 6;                        // v = group * 2 + statistic
    const float* pp = partial + (size_t)b * nchunks * 64 + v;
    // sixteen loads in flight per thread (a fixed order all the same): with four, the 4096 records per image of the 512 x 512
    // level were 64 dependent rounds of memory latency for the 4 workgroups of this launch
    double a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = 0.0;
    int c = part;
    for (; c + 240 < nchunks; c += 256) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = pp[(size_t)(c + 16 * u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] += (double)x[u];
    }
    for (; c < nchunks; c += 16) a[0] += (double)pp[(size_t)c * 64];
    acc[part][v] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
                   (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
    __syncthreads();
    if (tid < GN_G) {
        double su = 0.0, sq = 0.0;
        for (int k = 0; k < 16; ++k) { su += acc[k][2 * tid]; sq += acc[k][2 * tid + 1]; }
        const double mean = su / count;
        double var = sq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_out[b * GN_G + tid] = (float)mean;
        rstd_out[b * GN_G + tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// single-launch path (defined below the two-pass kernels): returns 1 when it took the call
static int gn_fused_fwd_try(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, float* y32, long ldy32,
                            void* y16, long ldy16, float* mean, float* rstd, float* workspace, int* sync, int B, int HW, int C,
                            float eps, int act, hipStream_t s);
static int gn_fused_bwd_try(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx, const float* gamma,
                            const float* beta, const float* mean, const float* rstd, float* dx32, long lddx32, int accumulate,
                            void* dx16, long lddx16, const float* add_src, long ldadd, float* workspace, int* sync, int B, int HW,
                            int C, int act, hipStream_t s);
static void gn_note_two_pass();

static void gn_chunks(int HW, int C, int* nchunks, int* rows_per_chunk) {
    long elems = (long)HW * C;
    int n = (int)(elems / 8192);
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    if (n > HW) n = HW;
    int rpc = (HW + n - 1) / n;
    n = (HW + rpc - 1) / rpc;
    *nchunks = n;
    *rows_per_chunk = rpc;
}

extern "C" long adap_groupnorm_workspace_floats(int B, int HW, int C) {
    int n, rpc;
    gn_chunks(HW, C, &n, &rpc);
    long two_pass = (long)B * n * GN_G * 2;
    long fused = 512L * 64;                 // single-launch path: one 64-float record per workgroup, <= 512 workgroups
    return two_pass > fused ? two_pass : fused;
}

extern "C" int adap_groupnorm_fwd(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta,
                                  float* y32, long ldy32, void* y16, long ldy16,
                                  float* mean, float* rstd, float* workspace, void* sync,
                                  int B, int HW, int C, float eps, int act, void* stream) {
    ADAP_REQUIRE(x && gamma && beta && mean && rstd && workspace && (y32 || y16), ADAP_ERR_SHAPE, "groupnorm_fwd: null pointer");
    ADAP_REQUIRE(x_dtype == 0 || x_dtype == 1, ADAP_ERR_UNSUPPORTED, "groupnorm_fwd: x_dtype %d", x_dtype);
    ADAP_REQUIRE(C % GN_G == 0 && C % 8 == 0 && C <= 256 * 8 * GN_MAXSLOT, ADAP_ERR_SHAPE, "groupnorm_fwd: C=%d", C);
    ADAP_REQUIRE(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, ADAP_ERR_ALIGN, "groupnorm_fwd: x alignment (ld %% 8, 16-byte base)");
    ADAP_REQUIRE(!y32 || (ldy32 % 4 == 0 && ((uintptr_t)y32 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_fwd: y32 alignment");
    ADAP_REQUIRE(!y16 || (ldy16 % 8 == 0 && ((uintptr_t)y16 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_fwd: y16 alignment");
    ADAP_REQUIRE(B > 0 && HW > 0, ADAP_ERR_SHAPE, "groupnorm_fwd: empty");
    int n, rpc;
    gn_chunks(HW, C, &n, &rpc);
    hipStream_t s = (hipStream_t)stream;
    if (sync && gn_fused_fwd_try(x, x_dtype, ldx, gamma, beta, y32, ldy32, y16, ldy16, mean, rstd, workspace, (int*)sync, B, HW, C,
                                 eps, act, s))
        return adap_check_launch("groupnorm_fwd (single launch)");
    gn_note_two_pass();
    if (x_dtype == 1) {
        hipLaunchKernelGGL(gn_stats_kernel<true>, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, rpc, workspace);
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, gamma, beta, workspace, n, eps, act,
                           rpc, y32, ldy32, (uint16_t*)y16, ldy16, mean, rstd);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<false>, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, rpc, workspace);
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, gamma, beta, workspace, n, eps, act,
                           rpc, y32, ldy32, (uint16_t*)y16, ldy16, mean, rstd);
    }
    return adap_check_launch("groupnorm_fwd");
}

extern "C" int adap_groupnorm_fwd_stats(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta,
                                        float* y32, long ldy32, void* y16, long ldy16, float* mean, float* rstd,
                                        const float* partial, int stats_chunks, int B, int HW, int C, float eps, int act,
                                        void* stream) {
    ADAP_REQUIRE(x && gamma && beta && mean && rstd && partial && (y32 || y16), ADAP_ERR_SHAPE, "groupnorm_fwd_stats: null pointer");
    ADAP_REQUIRE(x_dtype == 0 || x_dtype == 1, ADAP_ERR_UNSUPPORTED, "groupnorm_fwd_stats: x_dtype %d", x_dtype);
    ADAP_REQUIRE(C % GN_G == 0 && C % 8 == 0 && C <= 256 * 8 * GN_MAXSLOT, ADAP_ERR_SHAPE, "groupnorm_fwd_stats: C=%d", C);
    ADAP_REQUIRE(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, ADAP_ERR_ALIGN, "groupnorm_fwd_stats: x alignment");
    ADAP_REQUIRE(!y32 || (ldy32 % 4 == 0 && ((uintptr_t)y32 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_fwd_stats: y32 alignment");
    ADAP_REQUIRE(!y16 || (ldy16 % 8 == 0 && ((uintptr_t)y16 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_fwd_stats: y16 alignment");
    ADAP_REQUIRE(B > 0 && HW > 0 && stats_chunks > 0, ADAP_ERR_SHAPE, "groupnorm_fwd_stats: empty");
    ADAP_REQUIRE(HW % 64 == 0 && stats_chunks == HW / 64, ADAP_ERR_SHAPE,
                 "groupnorm_fwd_stats: %d records per image for HW=%d (one per 64 pixels expected)", stats_chunks, HW);
    int n, rpc;
    gn_chunks(HW, C, &n, &rpc);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_finish_kernel, dim3(B), dim3(1024), 0, s, partial, stats_chunks, (double)(C / GN_G) * HW, eps, mean, rstd);
    if (x_dtype == 1)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, gamma, beta, partial, 0, eps, act, rpc,
                           y32, ldy32, (uint16_t*)y16, ldy16, mean, rstd);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, gamma, beta, partial, 0, eps, act, rpc,
                           y32, ldy32, (uint16_t*)y16, ldy16, mean, rstd);
    return adap_check_launch("groupnorm_fwd_stats");
}

// ---------------------------------------------------------------------------------------------
// backward (data gradient).  y = act(xhat*gamma + beta), xhat = (x - mean)*rstd
//   dz  = dy * act'(z)            dyh = dz * gamma
//   dx  = rstd * (dyh - mean_g(dyh) - xhat * mean_g(dyh * xhat))
// pass 1 accumulates per (b, g) the two sums; pass 2 applies.  x and dy may each be f32 or bf16.
// ---------------------------------------------------------------------------------------------
template <bool XB, bool DYB>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const void* __restrict__ dy, long lddy, const void* __restrict__ x,
                                                           long ldx, int HW, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int act, int rows_per_chunk,
                                                           float* __restrict__ partial) {
    __shared__ float smA[GN_MAXSLOT * 8 * 256], chA[GN_MAXSLOT * 8 * 256];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const GnGeom g = gn_geom(C);
    const int cpg = C / GN_G;
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    float sA[GN_MAXSLOT][8], sB[GN_MAXSLOT][8];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { sA[k][e] = 0.f; sB[k][e] = 0.f; }
    if (r0 < g.rpp) {
        float sc[GN_MAXSLOT][8], sh[GN_MAXSLOT][8], ga[GN_MAXSLOT][8], be[GN_MAXSLOT][8];
#pragma unroll
        for (int k = 0; k < GN_MAXSLOT; ++k) {
            int q = lir + k * g.TPR;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (k < g.nslots && q < g.Q) {
                    int c = 8 * q + e, grp = c / cpg;
                    float rs = rstd[b * GN_G + grp];
                    sc[k][e] = rs;
                    sh[k][e] = -mean[b * GN_G + grp] * rs;
                    ga[k][e] = gamma[c];
                    be[k][e] = beta[c];
                }
            }
        }
        const int row_begin = chunk * rows_per_chunk;
        const int row_end = min(HW, row_begin + rows_per_chunk);
        const size_t boff = (size_t)b * HW;
        for (int r = row_begin + r0; r < row_end; r += g.rpp * GN_UNRB) {
            float xv[GN_UNRB][GN_MAXSLOT][8], d[GN_UNRB][GN_MAXSLOT][8];
#pragma unroll
            for (int u = 0; u < GN_UNRB; ++u) {
                const int rr = r + u * g.rpp;
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k) {
                    int q = lir + k * g.TPR;
                    if (rr < row_end && k < g.nslots && q < g.Q) {
                        gn_load8<XB>(x, (boff + rr) * ldx + 8 * q, xv[u][k]);
                        gn_load8<DYB>(dy, (boff + rr) * lddy + 8 * q, d[u][k]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GN_UNRB; ++u) {
                const int rr = r + u * g.rpp;
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k) {
                    int q = lir + k * g.TPR;
                    if (rr < row_end && k < g.nslots && q < g.Q) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float xh = xv[u][k][e] * sc[k][e] + sh[k][e];
                            float dz = d[u][k][e];
                            if (act) dz *= dsilu_f(xh * ga[k][e] + be[k][e]);
                            float dyh = dz * ga[k][e];
                            sA[k][e] += dyh;
                            sB[k][e] += dyh * xh;
                        }
                    }
                }
            }
        }
    }
    const float ra = gn_block_reduce(smA, chA, sA, g, C, cpg, tid);
    const float rb = gn_block_reduce(smA, chA, sB, g, C, cpg, tid);
    if (tid < GN_G) *(float2*)(partial + (((size_t)b * nchunks + chunk) * GN_G + tid) * 2) = make_float2(ra, rb);
}

template <bool XB, bool DYB>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const void* __restrict__ dy, long lddy, const void* __restrict__ x,
                                                           long ldx, int HW, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int act,
                                                           const float* __restrict__ partial, int stats_chunks,
                                                           int rows_per_chunk, float* __restrict__ dx32, long lddx32,
                                                           int accumulate, uint16_t* __restrict__ dx16, long lddx16,
                                                           const float* __restrict__ add_src, long ldadd) {
    __shared__ float lA[GN_G], lB[GN_G];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cpg = C / GN_G;
    {
        double a, bb;
        gn_finish_partials(partial, b, stats_chunks, tid, &a, &bb);
        if ((tid & 7) == 0) {
            double n = (double)cpg * HW;
            lA[tid >> 3] = (float)(a / n);
            lB[tid >> 3] = (float)(bb / n);
        }
    }
    __syncthreads();
    const GnGeom g = gn_geom(C);
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    if (r0 >= g.rpp) return;
    float sc[GN_MAXSLOT][8], sh[GN_MAXSLOT][8], ga[GN_MAXSLOT][8], be[GN_MAXSLOT][8], mA[GN_MAXSLOT][8], mB[GN_MAXSLOT][8];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k) {
        int q = lir + k * g.TPR;
        if (k < g.nslots && q < g.Q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int c = 8 * q + e, grp = c / cpg;
                float rs = rstd[b * GN_G + grp];
                sc[k][e] = rs;
                sh[k][e] = -mean[b * GN_G + grp] * rs;
                ga[k][e] = gamma[c];
                be[k][e] = beta[c];
                mA[k][e] = lA[grp];
                mB[k][e] = lB[grp];
            }
        }
    }
    const int row_begin = chunk * rows_per_chunk;
    const int row_end = min(HW, row_begin + rows_per_chunk);
    const size_t boff = (size_t)b * HW;
    for (int r = row_begin + r0; r < row_end; r += g.rpp * GN_UNRB) {
        float xv[GN_UNRB][GN_MAXSLOT][8], d[GN_UNRB][GN_MAXSLOT][8], ac[GN_UNRB][GN_MAXSLOT][8];
#pragma unroll
        for (int u = 0; u < GN_UNRB; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) {
                    gn_load8<XB>(x, (boff + rr) * ldx + 8 * q, xv[u][k]);
                    gn_load8<DYB>(dy, (boff + rr) * lddy + 8 * q, d[u][k]);
                    if (dx32 && accumulate) gn_load8<false>(add_src, (boff + rr) * ldadd + 8 * q, ac[u][k]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GN_UNRB; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float xh = xv[u][k][e] * sc[k][e] + sh[k][e];
                        float dz = d[u][k][e];
                        if (act) dz *= dsilu_f(xh * ga[k][e] + be[k][e]);
                        float dyh = dz * ga[k][e];
                        o[e] = sc[k][e] * (dyh - mA[k][e] - xh * mB[k][e]);
                        if (dx32 && accumulate) o[e] += ac[u][k][e];
                    }
                    if (dx32) gn_store8_f32(dx32 + (boff + rr) * lddx32 + 8 * q, o);
                    if (dx16) *(uint4*)(dx16 + (boff + rr) * lddx16 + 8 * q) = pack_bf16x8(o);
                }
            }
        }
    }
}

template <bool XB, bool DYB>
static void gn_bwd_launch(const void* dy, long lddy, const void* x, long ldx, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, float* dx32, long lddx32, int accumulate, void* dx16,
                          long lddx16, const float* add_src, long ldadd, float* workspace, int B, int HW, int C, int act, int n,
                          int rpc, hipStream_t s) {
    hipLaunchKernelGGL((gn_bwd_stats_kernel<XB, DYB>), dim3(n, B), dim3(256), 0, s, dy, lddy, x, ldx, HW, C, gamma, beta, mean,
                       rstd, act, rpc, workspace);
    hipLaunchKernelGGL((gn_bwd_apply_kernel<XB, DYB>), dim3(n, B), dim3(256), 0, s, dy, lddy, x, ldx, HW, C, gamma, beta, mean,
                       rstd, act, workspace, n, rpc, dx32, lddx32, accumulate, (uint16_t*)dx16, lddx16, add_src, ldadd);
}

extern "C" int adap_groupnorm_bwd(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  float* dx32, long lddx32, int accumulate, void* dx16, long lddx16,
                                  const float* add_src, long ldadd,
                                  float* workspace, void* sync, int B, int HW, int C, int act, void* stream) {
    ADAP_REQUIRE(dy && x && gamma && beta && mean && rstd && workspace && (dx32 || dx16), ADAP_ERR_SHAPE, "groupnorm_bwd: null pointer");
    ADAP_REQUIRE(C % GN_G == 0 && C % 8 == 0 && C <= 256 * 8 * GN_MAXSLOT, ADAP_ERR_SHAPE, "groupnorm_bwd: C=%d", C);
    ADAP_REQUIRE((dy_dtype == 0 || dy_dtype == 1) && (x_dtype == 0 || x_dtype == 1), ADAP_ERR_UNSUPPORTED, "groupnorm_bwd: dtype");
    ADAP_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, ADAP_ERR_ALIGN,
                 "groupnorm_bwd: x / dy alignment");
    ADAP_REQUIRE(!dx32 || (lddx32 % 4 == 0 && ((uintptr_t)dx32 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_bwd: dx32 alignment");
    ADAP_REQUIRE(!dx16 || (lddx16 % 8 == 0 && ((uintptr_t)dx16 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_bwd: dx16 alignment");
    if (accumulate && !add_src) {           // in-place accumulation: the addend is dx32 itself
        add_src = dx32;
        ldadd = lddx32;
    }
    ADAP_REQUIRE(!accumulate || (dx32 && add_src && ldadd % 4 == 0 && ((uintptr_t)add_src % 16) == 0), ADAP_ERR_ALIGN,
                 "groupnorm_bwd: accumulate needs dx32 and a 16-byte aligned addend");
    int n, rpc;
    gn_chunks(HW, C, &n, &rpc);
    hipStream_t s = (hipStream_t)stream;
    if (sync && gn_fused_bwd_try(dy, dy_dtype, lddy, x, x_dtype, ldx, gamma, beta, mean, rstd, dx32, lddx32, accumulate, dx16,
                                 lddx16, add_src, ldadd, workspace, (int*)sync, B, HW, C, act, s))
        return adap_check_launch("groupnorm_bwd (single launch)");
    gn_note_two_pass();
#define GN_BWD(XB, DYB) gn_bwd_launch<XB, DYB>(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx32, lddx32, accumulate, dx16, lddx16, \
                                               add_src, ldadd, \
                                              workspace, B, HW, C, act, n, rpc, s)
    if (x_dtype == 1 && dy_dtype == 1) GN_BWD(true, true);
    else if (x_dtype == 1) GN_BWD(true, false);
    else if (dy_dtype == 1) GN_BWD(false, true);
    else GN_BWD(false, false);
#undef GN_BWD
    return adap_check_launch("groupnorm_bwd");
}

// =============================================================================================
// Single-launch GroupNorm (forward and backward) for tensors whose slab per workgroup fits the register
// file: the UNet's shapes at the training batch sizes (<= 64x64x960 at bs 4).
//
// The two-pass kernels above read x twice and pay two launches; on a 21 MB tensor each launch is
// 6-17 us of mostly latency.  Here ONE grid of <= 256 workgroups (<= one per CU,
// so all are resident) of 512 threads walks the tensor ONCE: a workgroup loads its slab of whole
// pixel rows into registers (every load issued before the first use: ~10 x 16 B in flight per thread),
// reduces it to 32 x 2 group partials through LDS in a fixed order, publishes the 256-byte record
// write-through, meets the other workgroups of its SAMPLE at an arrival counter, reads the sample's
// records back (fp64 finish, fixed order: bit-reproducible), and normalises / differentiates the
// slab it still holds.  HBM traffic = the algorithmic bytes: x once, y once.
//
// Hand-off (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"; cdna_hip_programming.md
// Guideline 16, form R2 "the data IS the flag"): a workgroup's record is 64 granules of 8 bytes {value, tag = the launch's
// epoch}, each written by ONE relaxed agent-scope atomic store (global_store_dwordx2 sc1, write-through) of wave 0;
// consumers re-read the granules they need with 8-byte sc1 loads until every tag is this launch's -- no drain, no flag,
// no arrival counter on the critical path (the first version had all three: ~4 us more per launch).  Records and epoch
// live in a buffer the caller zero-initialises ONCE per stream: tags only ever grow, so a stale record can never match.
// The epoch word is advanced by the last workgroup of the launch to finish its sweep (a departure counter, off the
// critical path); every workgroup has read the epoch by then, because a sweep completes only after all workgroups of
// its sample have published.  A sweep that exceeds its bound sets the poison word and carries on with what it read: the
// grid always drains.
// =============================================================================================
#define GN_FT 512              // threads per workgroup
// the caller's persistent int32 buffer: [0] launch epoch, [1] poison, [2] departures, [64 ...) the records:
// GN_MAX_WGS x 64 granules of 8 bytes {value bits, tag}
#define GN_SYNC_EPOCH 0
#define GN_SYNC_POISON 1
#define GN_SYNC_DEPART 2
#define GN_SYNC_HDR 64
#define GN_MAX_WGS 512
#define GN_SYNC_STAMPS_ON 3      // diagnostic: non-zero -> workgroup w stores 6 shader-clock stamps (s_memtime) at GN_SYNC_STAMPS + 16 w
#define GN_SYNC_STAMPS (GN_SYNC_HDR + GN_MAX_WGS * 64 * 2)
#define GN_SYNC_INTS (GN_SYNC_STAMPS + GN_MAX_WGS * 16)
#define GN_SPIN_LIMIT (1u << 20)

typedef unsigned long long gn_u64;

// tools/gn_probe.py stamps: where a launch spends its time (loads landed / published / sweep done / stores issued).  The
// stamp values go only to the stamp area of the sync buffer, which nothing else reads; off (one uniform branch) by default.
__device__ __forceinline__ void gn_stamp(int* sync, bool on, int wg, int slot) {
    if (on && threadIdx.x == 0) {
        const gn_u64 t = __builtin_amdgcn_s_memtime();
        ((gn_u64*)(sync + GN_SYNC_STAMPS))[(size_t)wg * 8 + slot] = t;
    }
}

// publish this workgroup's 64 group partials: entry t (thread t < 64, one wave, one 512-byte store instruction) as the
// 8-byte granule {tag = epoch, value}: the data IS the flag -- no drain, no flag store, no arrival counter
__device__ __forceinline__ void gn_publish(int* sync, int wg, int tid, float v, unsigned tag) {
    if (tid < 64) {
        gn_u64* g = (gn_u64*)(sync + GN_SYNC_HDR) + (size_t)wg * 64 + tid;
        __hip_atomic_store(g, ((gn_u64)tag << 32) | (gn_u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Sum over the sample's slabs of every record entry, in fp64, fixed order.  Thread t sweeps entry (t & 63) of the records
// t >> 6, (t >> 6) + 8, ...: all of a thread's 8-byte sc1 loads are issued together and re-issued until every tag is this
// launch's (bounded: a give-up sets the poison word and carries on, so the grid always drains); wave w folds its records
// in order, then thread t < 64 folds the 8 waves in order.  Returns entry t's total in threads t < 64.
__device__ __forceinline__ double gn_sweep(int* sync, int first_wg, int nslab, int tid, unsigned tag, double* lds8x64) {
    const int e = tid & 63, w = tid >> 6;
    const gn_u64* rec = (const gn_u64*)(sync + GN_SYNC_HDR) + (size_t)first_wg * 64 + e;
    double acc = 0.0;
#pragma unroll 1
    for (int c0 = w; c0 < nslab; c0 += 64) {
        float v[8];
        unsigned spins = 0;
        bool gave_up = false;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + 8 * j;
                if (c < nslab) {
                    const gn_u64 x = __hip_atomic_load(rec + (size_t)c * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[j] = __builtin_bit_cast(float, (unsigned)x);
                    ok &= (unsigned)(x >> 32) == tag;
                } else {
                    v[j] = 0.f;
                }
            }
            if (__all(ok)) break;
            if (++spins > GN_SPIN_LIMIT) {
                if (e == 0) __hip_atomic_store(sync + GN_SYNC_POISON, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (double)v[j];
        // a timed-out exchange must never look like a result: the statistics become NaN, and with them this workgroup's
        // outputs, mean / rstd, the loss and every gradient downstream (the host also polls the poison word, ops.py)
        if (gave_up) acc = __builtin_nan("");
    }
    lds8x64[w * 64 + e] = acc;
    __syncthreads();
    double tot = 0.0;
    if (tid < 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += lds8x64[k * 64 + tid];
    }
    return tot;
}

// every workgroup, once its sweep is done (issued early, the returned count is only looked at when the kernel ends):
// the last one of the LAUNCH to leave resets the counter and opens the next epoch.  All workgroups have read the epoch
// by then: a sweep only completes after every workgroup of its sample has published, and the count after every sample.
__device__ __forceinline__ int gn_depart(int* sync) {
    return __hip_atomic_fetch_add(sync + GN_SYNC_DEPART, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gn_close(int* sync, int departed_before, int total_wgs) {
    if (departed_before == total_wgs - 1) {
        __hip_atomic_store(sync + GN_SYNC_DEPART, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sync + GN_SYNC_EPOCH, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// per-thread (group g0, group g0+1) partials of two statistics -> the 32 x 2 group sums of the workgroup, returned in
// thread t < 64 (g = t & 31, stat = t >> 5).  red: float4[GN_FT] in LDS.  8 threads per output (o = tid >> 3): thread p of
// them adds the contributions p, p + 8, ... (rows outer, octs inner) and a fixed 8-lane tree folds them: same order on
// every launch (bit-reproducible), ~5 LDS reads deep instead of ~40.
__device__ __forceinline__ float gn_fused_block_reduce(float4* red, float* out64, float a0, float b0, float a1, float b1, int Q,
                                                       int rpp, int cpg, int tid) {
    red[tid] = make_float4(a0, b0, a1, b1);
    __syncthreads();
    {
        const int o = tid >> 3, p = tid & 7;
        const int g = o & 31, stat = o >> 5;
        const int q_lo = (g * cpg) >> 3, q_hi = ((g + 1) * cpg - 1) >> 3;
        const int nq = q_hi - q_lo + 1, n = nq * rpp;
        float acc = 0.f;
        for (int i = p; i < n; i += 8) {
            const int r0 = i / nq, q = q_lo + (i - r0 * nq);
            const int g0 = (8 * q) / cpg;          // first group the oct touches
            const float4 v = red[r0 * Q + q];
            const int comp = (g == g0 ? 0 : 2) + stat;
            acc += comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        }
#pragma unroll
        for (int w = 4; w > 0; w >>= 1) acc += __shfl_down(acc, w, 8);
        if (p == 0) out64[o] = acc;
    }
    __syncthreads();
    return tid < 64 ? out64[tid] : 0.f;
}

// ---- buffer addressing of a slab: one descriptor per tensor (base = the slab's first row: wave-uniform, SGPRs), ONE
// per-thread byte offset (row r0, oct c0) and a SCALAR row step per pass -- the 64-bit address arithmetic per row and
// tensor that plain pointers need would otherwise sit in registers next to the slab itself.  A pass whose row lies beyond
// the slab (or a thread beyond the last whole row of a pass) gets an out-of-range offset: the load returns 0, the
// store is dropped (hardware range check), no branch.
typedef unsigned int gn_u32x4 __attribute__((ext_vector_type(4)));
#define GN_OOB 0xC0000000u

struct GnSlab {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;     // bytes, this thread's oct in pass 0 (GN_OOB for idle threads)
    unsigned step;     // bytes per pass
};

__device__ __forceinline__ GnSlab gn_slab(const void* base, long ld, int esz, size_t first_row, int r0, int c0, int rpp,
                                          bool live) {
    GnSlab t;
    t.rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + first_row * (size_t)ld * esz), 0, 0x80000000u, 0x00020000);
    t.voff = live ? (unsigned)((r0 * ld + c0) * esz) : GN_OOB;
    t.step = (unsigned)(rpp * ld * esz);
    return t;
}

template <bool BF> struct GnOct;
template <> struct GnOct<false> {
    gn_u32x4 a, b;
    __device__ __forceinline__ void load(const GnSlab& t, unsigned voff, int k) {
        a = __builtin_amdgcn_raw_buffer_load_b128(t.rs, voff, k * t.step, 0);
        b = __builtin_amdgcn_raw_buffer_load_b128(t.rs, voff + 16, k * t.step, 0);
    }
    // the compiler must not carry values DERIVED from these registers across the hand-off (it would keep xhat and dyh of
    // every oct alive through the spin: twice the registers, spills): make them opaque at that point
    __device__ __forceinline__ void opaque() {
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    }
    // (whole-vector bit casts: per-element `bit_cast(float, a[e])` makes hipcc 7.2 narrow the b128 load to ONE dword)
    __device__ __forceinline__ void get(float* o) const {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        o[0] = fa.x; o[1] = fa.y; o[2] = fa.z; o[3] = fa.w;
        o[4] = fb.x; o[5] = fb.y; o[6] = fb.z; o[7] = fb.w;
    }
};
template <> struct GnOct<true> {
    gn_u32x4 a;
    __device__ __forceinline__ void load(const GnSlab& t, unsigned voff, int k) {
        a = __builtin_amdgcn_raw_buffer_load_b128(t.rs, voff, k * t.step, 0);
    }
    __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
    __device__ __forceinline__ void get(float* o) const {
        uint4 v = make_uint4(a.x, a.y, a.z, a.w);
        unpack_bf16x8(v, o);
    }
};

// Stores go through plain global stores (uniform base + 32-bit offset, exec-masked): a 16-byte raw_buffer_store with an SGPR
// soffset is NOT safe on gfx950 with hipcc 7.2 -- the compiler reuses the data registers two instructions after the store
// (it knows no hazard for that form) and under back-pressure the store then ships the NEW contents of its second dword
// (measured: 40-150 corrupted octs per 5.2 M elements, always elements 2-3 of the packed bf16 oct).
struct GnOut {
    char* base;        // the slab's first row (wave-uniform)
    unsigned voff;     // bytes, this thread's oct in pass 0
    unsigned step;     // bytes per pass
    bool on;
};

__device__ __forceinline__ GnOut gn_out(void* p, long ld, int esz, size_t first_row, int r0, int c0, int rpp, bool on) {
    GnOut t;
    t.base = (char*)p + first_row * (size_t)ld * esz;
    t.voff = (unsigned)((r0 * ld + c0) * esz);
    t.step = (unsigned)(rpp * ld * esz);
    t.on = on;
    return t;
}

__device__ __forceinline__ void gn_store_f32(const GnOut& t, bool ok, int k, const float* o) {
    if (t.on && ok) {
        float4* p = (float4*)(t.base + (size_t)(t.voff + (unsigned)k * t.step));
        p[0] = make_float4(o[0], o[1], o[2], o[3]);
        p[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

__device__ __forceinline__ void gn_store_bf16(const GnOut& t, bool ok, int k, const float* o) {
    if (t.on && ok) *(uint4*)(t.base + (size_t)(t.voff + (unsigned)k * t.step)) = pack_bf16x8(o);
}

__device__ __forceinline__ void gn_load_f32x8(const float* p, float* o) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

template <bool XB, int NO, bool ACT>
__global__ __launch_bounds__(GN_FT) void gn_fused_fwd_kernel(const void* __restrict__ x, long ldx, int HW, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int rows_per_slab, double inv_n,
                                                             float* __restrict__ y32, long ldy32, uint16_t* __restrict__ y16,
                                                             long ldy16, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, float* partial, int* sync) {
    __shared__ float4 red[GN_FT];
    __shared__ double fin[GN_FT];
    __shared__ float out64[64], lmean[GN_G], lrstd[GN_G];
    const int tid = threadIdx.x;
    const int slab = blockIdx.x, nslab = gridDim.x, b = blockIdx.y;
    const int wg = b * nslab + slab;
    const gn_u64 t_start = __builtin_amdgcn_s_memtime();
    const int Q = C >> 3, cpg = C / GN_G;
    const int rpp = GN_FT / Q;
    const int lir = tid % Q, r0 = tid / Q;
    const bool live = r0 < rpp;
    const int row_begin = slab * rows_per_slab;
    const int nrows = min(HW, row_begin + rows_per_slab) - row_begin;
    const size_t first = (size_t)b * HW + row_begin;
    const int c0 = 8 * lir, g0 = min(c0 / cpg, GN_G - 1);
    const int n0 = min(8, (g0 + 1) * cpg - c0);          // channels of this oct that belong to g0; the rest to g0 + 1
    const GnSlab tx = gn_slab(x, ldx, XB ? 2 : 4, first, r0, c0, rpp, live);

    GnOct<XB> xr[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) xr[k].load(tx, (r0 + k * rpp < nrows) ? tx.voff : GN_OOB, k);
    // the epoch (an sc1 load, served by the memory side) is issued BEHIND the slab's loads: vmcnt retires in order, so in
    // front of them it would hold up the first use of x
    const unsigned tag = (unsigned)__hip_atomic_load(sync + GN_SYNC_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const bool stamps = sync[GN_SYNC_STAMPS_ON] != 0;
    if (stamps && tid == 0) ((gn_u64*)(sync + GN_SYNC_STAMPS))[(size_t)wg * 8 + 0] = t_start;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        float v[8];
        xr[k].get(v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] += v[e];
            ss[e] += v[e] * v[e];
        }
    }
    float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e < n0) { a0 += s[e]; b0 += ss[e]; }
        else { a1 += s[e]; b1 += ss[e]; }
    }
    const float mine = gn_fused_block_reduce(red, out64, a0, b0, a1, b1, Q, rpp, cpg, tid);
    gn_stamp(sync, stamps, wg, 1);                           // slab loaded and reduced
    gn_publish(sync, b * nslab + slab, tid, mine, tag);
    float ga[8], be[8];                                      // fetched under the hand-off's latency
    gn_load_f32x8(gamma + (live ? c0 : 0), ga);              // (idle threads: every store is masked)
    gn_load_f32x8(beta + (live ? c0 : 0), be);
    int departed = 0;
    {
        const double tot = gn_sweep(sync, b * nslab, nslab, tid, tag, fin);      // entry t: sum (t < 32) / sum of squares
        gn_stamp(sync, stamps, wg, 2);                       // every record of the sample read
        if (tid == 0) departed = gn_depart(sync);
        if (tid < 64) fin[tid] = tot;
        __syncthreads();
        if (tid < GN_G) {
            // (fp64 only where the cancellation is: an fp64 divide / sqrt is a ~1 us software sequence on the critical path)
            // inv_n = 1 / (channels per group * pixels), from the host
            const double mean = fin[tid] * inv_n;
            double var = fin[32 + tid] * inv_n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float vf = (float)(var + (double)eps);
            float rstd = __builtin_amdgcn_rsqf(vf);
            rstd = rstd * (1.5f - 0.5f * vf * rstd * rstd);       // one Newton step: <= 1 ulp of the correctly rounded value
            lmean[tid] = (float)mean;
            lrstd[tid] = rstd;
            if (slab == 0) {
                mean_out[b * GN_G + tid] = (float)mean;
                rstd_out[b * GN_G + tid] = rstd;
            }
        }
    }
    __syncthreads();
    gn_stamp(sync, stamps, wg, 5);                           // statistics finished
    float sc[8], sh[8];
    {
        const int g1 = min(g0 + 1, GN_G - 1);
        const float rs0 = lrstd[g0], rs1 = lrstd[g1], mu0 = lmean[g0], mu1 = lmean[g1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = (e < n0 ? rs0 : rs1) * ga[e];
            sc[e] = a;
            sh[e] = be[e] - (e < n0 ? mu0 : mu1) * a;
        }
    }
    const GnOut t32 = gn_out(y32, ldy32, 4, first, r0, c0, rpp, live && y32 != nullptr);
    const GnOut t16 = gn_out(y16, ldy16, 2, first, r0, c0, rpp, live && y16 != nullptr);
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const bool ok = r0 + k * rpp < nrows;
        float v[8], o[8];
        xr[k].get(v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = v[e] * sc[e] + sh[e];
            if (ACT) o[e] = silu_f(o[e]);
        }
        gn_store_f32(t32, ok, k, o);
        gn_store_bf16(t16, ok, k, o);
    }
    gn_stamp(sync, stamps, wg, 3);                           // stores issued
    if (tid == 0) gn_close(sync, departed, nslab * gridDim.y);
    gn_stamp(sync, stamps, wg, 4);
}

template <bool XB, bool DYB, int NO, bool ACT>
__global__ __launch_bounds__(GN_FT) void gn_fused_bwd_kernel(const void* __restrict__ dy, long lddy, const void* __restrict__ x,
                                                             long ldx, int HW, int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int rows_per_slab, double inv_n,
                                                             float* __restrict__ dx32, long lddx32, int accumulate,
                                                             uint16_t* __restrict__ dx16, long lddx16,
                                                             const float* __restrict__ add_src, long ldadd, float* partial,
                                                             int* sync) {
    __shared__ float4 red[GN_FT];
    __shared__ double fin[GN_FT];
    __shared__ float out64[64], lA[GN_G], lB[GN_G];
    const int tid = threadIdx.x;
    const int slab = blockIdx.x, nslab = gridDim.x, b = blockIdx.y;
    const int Q = C >> 3, cpg = C / GN_G;
    const int rpp = GN_FT / Q;
    const int lir = tid % Q, r0 = tid / Q;
    const bool live = r0 < rpp;
    const int row_begin = slab * rows_per_slab;
    const int nrows = min(HW, row_begin + rows_per_slab) - row_begin;
    const size_t first = (size_t)b * HW + row_begin;
    const int c0 = 8 * lir, g0 = min(c0 / cpg, GN_G - 1), g1 = min(g0 + 1, GN_G - 1);
    const int n0 = min(8, (g0 + 1) * cpg - c0);
    const GnSlab tx = gn_slab(x, ldx, XB ? 2 : 4, first, r0, c0, rpp, live);
    const GnSlab td = gn_slab(dy, lddy, DYB ? 2 : 4, first, r0, c0, rpp, live);

    GnOct<XB> xr[NO];
    GnOct<DYB> dr[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const bool ok = r0 + k * rpp < nrows;
        xr[k].load(tx, ok ? tx.voff : GN_OOB, k);
        dr[k].load(td, ok ? td.voff : GN_OOB, k);
    }
    const unsigned tag = (unsigned)__hip_atomic_load(sync + GN_SYNC_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // per-channel constants: xhat = x * sc + sh, z = xhat * ga + be
    float sc[8], sh[8], ga[8], be[8];
    {
        const int cc = live ? c0 : 0;
        gn_load_f32x8(gamma + cc, ga);
        gn_load_f32x8(beta + cc, be);
        const float rs0 = rstd[b * GN_G + g0], rs1 = rstd[b * GN_G + g1];
        const float mu0 = mean[b * GN_G + g0], mu1 = mean[b * GN_G + g1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = e < n0 ? rs0 : rs1;
            sh[e] = -(e < n0 ? mu0 : mu1) * sc[e];
        }
    }
    float sA[8], sB[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sA[e] = 0.f; sB[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        float xv[8], d[8];
        xr[k].get(xv);
        dr[k].get(d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = xv[e] * sc[e] + sh[e];
            float dz = d[e];                                     // zero for the padding rows: they add nothing
            if (ACT) dz *= dsilu_f(xh * ga[e] + be[e]);
            const float dyh = dz * ga[e];
            sA[e] += dyh;
            sB[e] += dyh * xh;
        }
        __builtin_amdgcn_sched_barrier(0);       // one oct at a time: the unrolled chains would otherwise all be live at once
    }
    float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e < n0) { a0 += sA[e]; b0 += sB[e]; }
        else { a1 += sA[e]; b1 += sB[e]; }
    }
    const float mine = gn_fused_block_reduce(red, out64, a0, b0, a1, b1, Q, rpp, cpg, tid);
    gn_publish(sync, b * nslab + slab, tid, mine, tag);
    int departed = 0;
    {
        const double tot = gn_sweep(sync, b * nslab, nslab, tid, tag, fin);
        if (tid == 0) departed = gn_depart(sync);
        if (tid < 64) {
            if (tid < 32) lA[tid] = (float)(tot * inv_n);
            else lB[tid - 32] = (float)(tot * inv_n);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        xr[k].opaque();
        dr[k].opaque();
    }
    const float mA0 = lA[g0], mA1 = lA[g1], mB0 = lB[g0], mB1 = lB[g1];
    const GnOut t32 = gn_out(dx32, lddx32, 4, first, r0, c0, rpp, live && dx32 != nullptr);
    const GnOut t16 = gn_out(dx16, lddx16, 2, first, r0, c0, rpp, live && dx16 != nullptr);
    const GnSlab tad = gn_slab(add_src, ldadd, 4, first, r0, c0, rpp, live && dx32 != nullptr && accumulate);
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const bool ok = r0 + k * rpp < nrows;
        float xv[8], d[8], o[8], ac[8];
        GnOct<false> acr;
        acr.load(tad, ok ? tad.voff : GN_OOB, k);                // zeros when there is nothing to add
        xr[k].get(xv);
        dr[k].get(d);
        acr.get(ac);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = xv[e] * sc[e] + sh[e];
            float dz = d[e];
            if (ACT) dz *= dsilu_f(xh * ga[e] + be[e]);
            const float dyh = dz * ga[e];
            o[e] = sc[e] * (dyh - (e < n0 ? mA0 : mA1) - xh * (e < n0 ? mB0 : mB1)) + ac[e];
        }
        gn_store_f32(t32, ok, k, o);
        gn_store_bf16(t16, ok, k, o);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) gn_close(sync, departed, nslab * gridDim.y);
}

// geometry of the single-launch path: slabs per sample so that B * nslab <= the CU count; 0 octs = not eligible
static int g_gn_cus = 0;
static int g_gn_last_variant = 0;          // 0 = two-pass, N > 0 = single launch with N octs per thread
static int gn_fused_geom(int B, int HW, int C, int max_octs, int* nslab, int* rows_per_slab) {
    static const bool env_two_pass = getenv("ADAP_GN_TWO_PASS") != nullptr;       // (read once; the host can also pass sync = NULL)
    if (env_two_pass) return 0;
    if (g_gn_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 0;
        g_gn_cus = n > 0 ? n : -1;
    }
    const int Q = C >> 3;
    if (g_gn_cus <= 0 || Q > GN_FT || B > g_gn_cus || g_gn_cus > GN_MAX_WGS) return 0;
    if (C / GN_G < 8) return 0;      // the kernels assume an 8-channel oct touches at most two groups
    const int rpp = GN_FT / Q;
    int ns = g_gn_cus / B;
    if (ns > 128) ns = 128;          // every workgroup sweeps all of its sample's records: past ~128 the sweep costs more than
    if (ns > HW) ns = HW;            // the extra CUs bring (one instance: 5 MB tensors)
    int rps = (HW + ns - 1) / ns;
    rps = (rps + rpp - 1) / rpp * rpp;                 // whole passes: no half-empty last pass inside a slab
    ns = (HW + rps - 1) / rps;
    const int octs = rps / rpp;
    if (octs > max_octs) return 0;
    if ((long)rps * C * 4 >= (1L << 30)) return 0;     // 32-bit byte offsets inside a slab
    *nslab = ns;
    *rows_per_slab = rps;
    return octs;
}

extern "C" int adap_groupnorm_last_variant(void) { return g_gn_last_variant; }
extern "C" long adap_groupnorm_sync_ints(void) { return GN_SYNC_INTS; }

static void gn_note_two_pass() { g_gn_last_variant = 0; }

static int gn_fused_fwd_try(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, float* y32, long ldy32,
                            void* y16, long ldy16, float* mean, float* rstd, float* workspace, int* sync, int B, int HW, int C,
                            float eps, int act, hipStream_t s) {
    int nslab, rps;
    const int octs = gn_fused_geom(B, HW, C, 16, &nslab, &rps);
    if (!octs) return 0;
#define GN_FF(XB, NO, ACT) hipLaunchKernelGGL((gn_fused_fwd_kernel<XB, NO, ACT>), dim3(nslab, B), dim3(GN_FT), 0, s, x, ldx, HW, C, \
                                              gamma, beta, eps, rps, 1.0 / ((double)(C / GN_G) * HW), y32, ldy32, (uint16_t*)y16,    \
                                              ldy16, mean, rstd, workspace, sync)
#define GN_FF2(XB, ACT) do { if (octs <= 6) GN_FF(XB, 6, ACT); else if (octs <= 12) GN_FF(XB, 12, ACT); else GN_FF(XB, 16, ACT); } while (0)
    if (x_dtype == 1) { if (act) GN_FF2(true, true); else GN_FF2(true, false); }
    else { if (act) GN_FF2(false, true); else GN_FF2(false, false); }
#undef GN_FF2
#undef GN_FF
    g_gn_last_variant = octs;
    return 1;
}

static int gn_fused_bwd_try(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx, const float* gamma,
                            const float* beta, const float* mean, const float* rstd, float* dx32, long lddx32, int accumulate,
                            void* dx16, long lddx16, const float* add_src, long ldadd, float* workspace, int* sync, int B, int HW,
                            int C, int act, hipStream_t s) {
    int nslab, rps;
    const int octs = gn_fused_geom(B, HW, C, 8, &nslab, &rps);
    if (!octs) return 0;
#define GN_FB(XB, DYB, NO, ACT) hipLaunchKernelGGL((gn_fused_bwd_kernel<XB, DYB, NO, ACT>), dim3(nslab, B), dim3(GN_FT), 0, s, dy, \
                                                   lddy, x, ldx, HW, C, gamma, beta, mean, rstd, rps,                           \
                                                   1.0 / ((double)(C / GN_G) * HW), dx32, lddx32, accumulate,                  \
                                                   (uint16_t*)dx16, lddx16, add_src, ldadd, workspace, sync)
// (12 rows of f32 x + dy per thread do not fit 256 registers next to the per-channel constants: such shapes -- 64x64x640 at
// bs 4 -- take the two-pass path in the backward)
#define GN_FB2(XB, DYB) do { if (act) { if (octs <= 4) GN_FB(XB, DYB, 4, true); else GN_FB(XB, DYB, 8, true); }      \
                             else { if (octs <= 4) GN_FB(XB, DYB, 4, false); else GN_FB(XB, DYB, 8, false); } } while (0)
    if (x_dtype == 1 && dy_dtype == 1) GN_FB2(true, true);
    else if (x_dtype == 1) GN_FB2(true, false);
    else if (dy_dtype == 1) GN_FB2(false, true);
    else GN_FB2(false, false);
#undef GN_FB2
#undef GN_FB
    g_gn_last_variant = octs;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the channel dim of [rows][D] (attention.py:267-269: eps 1e-5), one wave per row.
// forward writes bf16 (operand of to_q/to_k/to_v/GEGLU) and the per-row mean / rstd.
// ---------------------------------------------------------------------------------------------
#define LN_MAXV 5    // float4 per lane: D <= 1280

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, uint16_t* __restrict__ y16, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows, int D,
                                                     float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Q = D >> 2;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int q = lane + 64 * k;
        if (q < Q) {
            v[k] = *(const float4*)(x + row * ldx + 4 * q);
            s += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    const float mu = wave_sum(s) / D;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int q = lane + 64 * k;
        if (q < Q) {
            float a = v[k].x - mu, b = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
            ss += a * a + b * b + c * c + d * d;
        }
    }
    const float rs = rsqrtf(wave_sum(ss) / D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int q = lane + 64 * k;
        if (q < Q) {
            float4 g = *(const float4*)(gamma + 4 * q), b = *(const float4*)(beta + 4 * q);
            uint2 w;
            w.x = pack_bf16x2((v[k].x - mu) * rs * g.x + b.x, (v[k].y - mu) * rs * g.y + b.y);
            w.y = pack_bf16x2((v[k].z - mu) * rs * g.z + b.z, (v[k].w - mu) * rs * g.w + b.w);
            *(uint2*)(y16 + row * ldy + 4 * q) = w;
        }
    }
}

extern "C" int adap_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y16, long ldy,
                                  float* mean, float* rstd, long rows, int D, float eps, void* stream) {
    ADAP_REQUIRE(x && gamma && beta && y16 && mean && rstd, ADAP_ERR_SHAPE, "layernorm_fwd: null pointer");
    ADAP_REQUIRE(D % 4 == 0 && D <= 256 * LN_MAXV, ADAP_ERR_SHAPE, "layernorm_fwd: D=%d", D);
    ADAP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, ADAP_ERR_ALIGN, "layernorm_fwd: ld alignment");
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                       (uint16_t*)y16, ldy, mean, rstd, rows, D, eps);
    return adap_check_launch("layernorm_fwd");
}

// backward: dx (+= into the f32 residual-stream gradient) = rstd*(dyh - mean(dyh) - xhat*mean(dyh*xhat)), dyh = dy*gamma
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                     long ldx, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx, long lddx,
                                                     int accumulate, uint16_t* __restrict__ dx16, long lddx16, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Q = D >> 2;
    const float mu = mean[row], rs = rstd[row];
    float dyh[LN_MAXV][4], xh[LN_MAXV][4];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int q = lane + 64 * k;
        if (q < Q) {
            float4 xv = *(const float4*)(x + row * ldx + 4 * q);
            float4 dv = *(const float4*)(dy + row * lddy + 4 * q);
            float4 g = *(const float4*)(gamma + 4 * q);
            float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w}, gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[k][e] = (xs[e] - mu) * rs;
                dyh[k][e] = ds[e] * gs[e];
                sa += dyh[k][e];
                sb += dyh[k][e] * xh[k][e];
            }
        }
    }
    sa = wave_sum(sa) / D;
    sb = wave_sum(sb) / D;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int q = lane + 64 * k;
        if (q < Q) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (dyh[k][e] - sa - xh[k][e] * sb);
            float* dst = dx + row * lddx + 4 * q;
            if (accumulate) {
                float4 t = *(const float4*)dst;
                o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
            }
            *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
            if (dx16) {      // bf16 copy of the (accumulated) residual-stream gradient: operand of the next data-gradient GEMM
                uint2 w;
                w.x = pack_bf16x2(o[0], o[1]);
                w.y = pack_bf16x2(o[2], o[3]);
                *(uint2*)(dx16 + row * lddx16 + 4 * q) = w;
            }
        }
    }
}

extern "C" int adap_layernorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                                  const float* mean, const float* rstd, float* dx, long lddx, int accumulate,
                                  void* dx16, long lddx16, long rows, int D, void* stream) {
    ADAP_REQUIRE(dy && x && gamma && mean && rstd && dx, ADAP_ERR_SHAPE, "layernorm_bwd: null pointer");
    ADAP_REQUIRE(D % 4 == 0 && D <= 256 * LN_MAXV, ADAP_ERR_SHAPE, "layernorm_bwd: D=%d", D);
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx,
                       gamma, mean, rstd, dx, lddx, accumulate, (uint16_t*)dx16, lddx16, rows, D);
    return adap_check_launch("layernorm_bwd");
}

// GroupNorm(32)(+SiLU) and LayerNorm, forward and data-gradient, for pixel-major (NHWC) tensors.
//
// HBM-bound kernels.  GroupNorm32 is computed in fp32 (util.py:217-219) over [HW x C/32] per
// (sample, group); eps 1e-5 in the UNet ResBlocks / out, 1e-6 in SpatialTransformer.norm and in
// all VAE norms (attention.py:71-72, model.py:39-40) -- eps is an argument.  In the pixel-major
// layout a group is a run of C/32 consecutive channels of every pixel, so a block streams a slab
// of whole pixel rows with fully coalesced 16-byte loads and every thread owns fixed channel
// quads; group sums are combined through LDS, per-slab partials go to a small workspace
// ([B][nchunks][32][2] floats) and the apply pass finishes the reduction deterministically (no
// float atomics in HBM).  Output is bf16 (operand of the following contraction) and/or f32.
//
// Algorithmic bytes per element: forward 4 (stats read) + 4 (apply read) + 2 (bf16
// write); the second read hits the Infinity Cache for the UNet's tensors (<= 84 MB at bs=4).
#include "common.h"

#define GN_G 32
#define GN_MAXSLOT 3     // C <= 3072
#define GN_UNR 4         // pixel rows in flight per thread
#define GN_UNRB 2        // ... in the backward (two input streams)

struct GnGeom {
    int Q;        // float4 quads per pixel row (C/4)
    int TPR;      // threads per row
    int rpp;      // rows per pass
    int nslots;   // quads per thread
};

__device__ __forceinline__ GnGeom gn_geom(int C) {
    GnGeom g;
    g.Q = C >> 2;
    g.TPR = g.Q < 256 ? g.Q : 256;
    g.rpp = 256 / g.TPR;
    g.nslots = (g.Q + g.TPR - 1) / g.TPR;
    return g;
}

// Deterministic block reduction of per-thread channel sums into the 32 group sums: every thread parks its
// (slot, e) partials in LDS, then thread g < 32 adds the contributions of group g's channels in a fixed order
// (no float atomics, so results are bit-reproducible run to run).  sm: [GN_MAXSLOT*4][256] floats.
__device__ __forceinline__ float gn_group_sum(const float* sm, const GnGeom& g, int cpg, int grp) {
    float acc = 0.f;
    for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
        int q = c >> 2, e = c & 3;
        int slot = q / g.TPR, lir = q - slot * g.TPR;
        const float* col = sm + (slot * 4 + e) * 256 + lir;
        for (int r0 = 0; r0 < g.rpp; ++r0) acc += col[r0 * g.TPR];
    }
    return acc;
}

__device__ __forceinline__ void gn_park(float* sm, const float (*v)[4], int tid) {
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) sm[(k * 4 + e) * 256 + tid] = v[k][e];
}

// ---------------------------------------------------------------------------------------------
// forward pass 1: per-slab partial sums.  grid (nchunks, B), block 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, long ldx, int HW, int C,
                                                       int rows_per_chunk, float* __restrict__ partial) {
    __shared__ float smA[GN_MAXSLOT * 4 * 256], smB[GN_MAXSLOT * 4 * 256];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const GnGeom g = gn_geom(C);
    const int cpg = C / GN_G;
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    const int row_begin = chunk * rows_per_chunk;
    const int row_end = min(HW, row_begin + rows_per_chunk);
    float s[GN_MAXSLOT][4], ss[GN_MAXSLOT][4];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[k][e] = 0.f; ss[k][e] = 0.f; }
    if (r0 < g.rpp) {
        const float* xb = x + (size_t)b * HW * ldx;
        // GN_UNR rows in flight per thread: the kernel is a pure HBM stream, memory-level parallelism is the lever
        for (int r = row_begin + r0; r < row_end; r += g.rpp * GN_UNR) {
            float4 v[GN_UNR][GN_MAXSLOT];
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u) {
                const int rr = r + u * g.rpp;
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k) {
                    int q = lir + k * g.TPR;
                    v[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rr < row_end && k < g.nslots && q < g.Q) v[u][k] = *(const float4*)(xb + (size_t)rr * ldx + 4 * q);
                }
            }
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u)
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k) {
                    s[k][0] += v[u][k].x; ss[k][0] += v[u][k].x * v[u][k].x;
                    s[k][1] += v[u][k].y; ss[k][1] += v[u][k].y * v[u][k].y;
                    s[k][2] += v[u][k].z; ss[k][2] += v[u][k].z * v[u][k].z;
                    s[k][3] += v[u][k].w; ss[k][3] += v[u][k].w * v[u][k].w;
                }
        }
    }
    gn_park(smA, s, tid);
    gn_park(smB, ss, tid);
    __syncthreads();
    if (tid < GN_G) {
        float2 o = make_float2(gn_group_sum(smA, g, cpg, tid), gn_group_sum(smB, g, cpg, tid));
        *(float2*)(partial + (((size_t)b * nchunks + chunk) * GN_G + tid) * 2) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// forward pass 2: finish the reduction, normalise, affine, optional SiLU, write bf16 / f32.
// grid (nchunks_apply, B).  stats_chunks = grid.x of the stats launch.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, long ldx, int HW, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ partial, int stats_chunks, float eps, int act,
                                                       int rows_per_chunk, float* __restrict__ y32, long ldy32,
                                                       uint16_t* __restrict__ y16, long ldy16,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ float lmean[GN_G], lrstd[GN_G];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cpg = C / GN_G;
    if (tid < GN_G) {
        double su = 0.0, sq = 0.0;
        const float* pp = partial + ((size_t)b * stats_chunks * GN_G + tid) * 2;
        for (int c = 0; c < stats_chunks; ++c) {
            float2 v = *(const float2*)(pp + (size_t)c * GN_G * 2);
            su += v.x; sq += v.y;
        }
        double n = (double)cpg * HW;
        double mean = su / n;
        double var = sq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        lmean[tid] = (float)mean;
        lrstd[tid] = rstd;
        if (chunk == 0) {
            mean_out[b * GN_G + tid] = (float)mean;
            rstd_out[b * GN_G + tid] = rstd;
        }
    }
    __syncthreads();
    const GnGeom g = gn_geom(C);
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    if (r0 >= g.rpp) return;
    float sc[GN_MAXSLOT][4], sh[GN_MAXSLOT][4];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k) {
        int q = lir + k * g.TPR;
        if (k < g.nslots && q < g.Q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int c = 4 * q + e, grp = c / cpg;
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
        float4 v[GN_UNR][GN_MAXSLOT];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) v[u][k] = *(const float4*)(x + (boff + rr) * ldx + 4 * q);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) {
                    float o[4] = {v[u][k].x * sc[k][0] + sh[k][0], v[u][k].y * sc[k][1] + sh[k][1],
                                  v[u][k].z * sc[k][2] + sh[k][2], v[u][k].w * sc[k][3] + sh[k][3]};
                    if (act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = silu_f(o[e]);
                    }
                    if (y32) *(float4*)(y32 + (boff + rr) * ldy32 + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
                    if (y16) {
                        uint2 w;
                        w.x = pack_bf16x2(o[0], o[1]);
                        w.y = pack_bf16x2(o[2], o[3]);
                        *(uint2*)(y16 + (boff + rr) * ldy16 + 4 * q) = w;
                    }
                }
            }
        }
    }
}

static void gn_chunks(int HW, int C, int* nchunks, int* rows_per_chunk) {
    long elems = (long)HW * C;
    int n = (int)(elems / 16384);
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
    return (long)B * n * GN_G * 2;
}

extern "C" int adap_groupnorm_fwd(const float* x, long ldx, const float* gamma, const float* beta,
                                  float* y32, long ldy32, void* y16, long ldy16,
                                  float* mean, float* rstd, float* workspace,
                                  int B, int HW, int C, float eps, int act, void* stream) {
    ADAP_REQUIRE(x && gamma && beta && mean && rstd && workspace && (y32 || y16), ADAP_ERR_SHAPE, "groupnorm_fwd: null pointer");
    ADAP_REQUIRE(C % GN_G == 0 && C % 4 == 0 && C <= 256 * 4 * GN_MAXSLOT, ADAP_ERR_SHAPE, "groupnorm_fwd: C=%d", C);
    ADAP_REQUIRE(ldx % 4 == 0 && ((uintptr_t)x % 16) == 0, ADAP_ERR_ALIGN, "groupnorm_fwd: x alignment");
    ADAP_REQUIRE(!y32 || (ldy32 % 4 == 0 && ((uintptr_t)y32 % 16) == 0), ADAP_ERR_ALIGN, "groupnorm_fwd: y32 alignment");
    ADAP_REQUIRE(!y16 || (ldy16 % 4 == 0 && ((uintptr_t)y16 % 8) == 0), ADAP_ERR_ALIGN, "groupnorm_fwd: y16 alignment");
    ADAP_REQUIRE(B > 0 && HW > 0, ADAP_ERR_SHAPE, "groupnorm_fwd: empty");
    int n, rpc;
    gn_chunks(HW, C, &n, &rpc);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, rpc, workspace);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(n, B), dim3(256), 0, s, x, ldx, HW, C, gamma, beta, workspace, n, eps, act,
                       rpc, y32, ldy32, (uint16_t*)y16, ldy16, mean, rstd);
    return adap_check_launch("groupnorm_fwd");
}

// ---------------------------------------------------------------------------------------------
// backward (data gradient).  y = act(xhat*gamma + beta), xhat = (x - mean)*rstd
//   dz  = dy * act'(z)            dyh = dz * gamma
//   dx  = rstd * (dyh - mean_g(dyh) - xhat * mean_g(dyh * xhat))
// pass 1 accumulates per (b, g) the two sums; pass 2 applies.  dy may be f32 or bf16.
// ---------------------------------------------------------------------------------------------
template <bool DY_BF16>
__device__ __forceinline__ void load_dy4(const void* dy, size_t off, float* o) {
    if (DY_BF16) {
        uint2 v = *(const uint2*)((const uint16_t*)dy + off);
        o[0] = __builtin_bit_cast(float, v.x << 16);
        o[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
        o[2] = __builtin_bit_cast(float, v.y << 16);
        o[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
    } else {
        float4 v = *(const float4*)((const float*)dy + off);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
}

template <bool DY_BF16>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const void* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                           long ldx, int HW, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int act, int rows_per_chunk,
                                                           float* __restrict__ partial) {
    __shared__ float smA[GN_MAXSLOT * 4 * 256], smB[GN_MAXSLOT * 4 * 256];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const GnGeom g = gn_geom(C);
    const int cpg = C / GN_G;
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    float sA[GN_MAXSLOT][4], sB[GN_MAXSLOT][4];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sA[k][e] = 0.f; sB[k][e] = 0.f; }
    if (r0 < g.rpp) {
        float sc[GN_MAXSLOT][4], sh[GN_MAXSLOT][4], ga[GN_MAXSLOT][4], be[GN_MAXSLOT][4];
#pragma unroll
        for (int k = 0; k < GN_MAXSLOT; ++k) {
            int q = lir + k * g.TPR;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k < g.nslots && q < g.Q) {
                    int c = 4 * q + e, grp = c / cpg;
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
            float4 xv[GN_UNRB][GN_MAXSLOT];
            float d[GN_UNRB][GN_MAXSLOT][4];
#pragma unroll
            for (int u = 0; u < GN_UNRB; ++u) {
                const int rr = r + u * g.rpp;
#pragma unroll
                for (int k = 0; k < GN_MAXSLOT; ++k) {
                    int q = lir + k * g.TPR;
                    if (rr < row_end && k < g.nslots && q < g.Q) {
                        xv[u][k] = *(const float4*)(x + (boff + rr) * ldx + 4 * q);
                        load_dy4<DY_BF16>(dy, (boff + rr) * lddy + 4 * q, d[u][k]);
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
                        float xs[4] = {xv[u][k].x, xv[u][k].y, xv[u][k].z, xv[u][k].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float xh = xs[e] * sc[k][e] + sh[k][e];
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
    gn_park(smA, sA, tid);
    gn_park(smB, sB, tid);
    __syncthreads();
    if (tid < GN_G)
        *(float2*)(partial + (((size_t)b * nchunks + chunk) * GN_G + tid) * 2) = make_float2(gn_group_sum(smA, g, cpg, tid), gn_group_sum(smB, g, cpg, tid));
}

template <bool DY_BF16>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const void* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                           long ldx, int HW, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int act,
                                                           const float* __restrict__ partial, int stats_chunks,
                                                           int rows_per_chunk, float* __restrict__ dx32, long lddx32,
                                                           int accumulate, uint16_t* __restrict__ dx16, long lddx16) {
    __shared__ float lA[GN_G], lB[GN_G];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cpg = C / GN_G;
    if (tid < GN_G) {
        double a = 0.0, bb = 0.0;
        const float* pp = partial + ((size_t)b * stats_chunks * GN_G + tid) * 2;
        for (int c = 0; c < stats_chunks; ++c) {
            float2 v = *(const float2*)(pp + (size_t)c * GN_G * 2);
            a += v.x; bb += v.y;
        }
        double n = (double)cpg * HW;
        lA[tid] = (float)(a / n);
        lB[tid] = (float)(bb / n);
    }
    __syncthreads();
    const GnGeom g = gn_geom(C);
    const int lir = tid % g.TPR, r0 = tid / g.TPR;
    if (r0 >= g.rpp) return;
    float sc[GN_MAXSLOT][4], sh[GN_MAXSLOT][4], ga[GN_MAXSLOT][4], be[GN_MAXSLOT][4], mA[GN_MAXSLOT][4], mB[GN_MAXSLOT][4];
#pragma unroll
    for (int k = 0; k < GN_MAXSLOT; ++k) {
        int q = lir + k * g.TPR;
        if (k < g.nslots && q < g.Q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int c = 4 * q + e, grp = c / cpg;
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
        float4 xv[GN_UNRB][GN_MAXSLOT], acc4[GN_UNRB][GN_MAXSLOT];
        float d[GN_UNRB][GN_MAXSLOT][4];
#pragma unroll
        for (int u = 0; u < GN_UNRB; ++u) {
            const int rr = r + u * g.rpp;
#pragma unroll
            for (int k = 0; k < GN_MAXSLOT; ++k) {
                int q = lir + k * g.TPR;
                if (rr < row_end && k < g.nslots && q < g.Q) {
                    xv[u][k] = *(const float4*)(x + (boff + rr) * ldx + 4 * q);
                    load_dy4<DY_BF16>(dy, (boff + rr) * lddy + 4 * q, d[u][k]);
                    if (dx32 && accumulate) acc4[u][k] = *(const float4*)(dx32 + (boff + rr) * lddx32 + 4 * q);
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
                    float xs[4] = {xv[u][k].x, xv[u][k].y, xv[u][k].z, xv[u][k].w};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xh = xs[e] * sc[k][e] + sh[k][e];
                        float dz = d[u][k][e];
                        if (act) dz *= dsilu_f(xh * ga[k][e] + be[k][e]);
                        float dyh = dz * ga[k][e];
                        o[e] = sc[k][e] * (dyh - mA[k][e] - xh * mB[k][e]);
                    }
                    if (dx32) {
                        float* dst = dx32 + (boff + rr) * lddx32 + 4 * q;
                        if (accumulate) {
                            o[0] += acc4[u][k].x; o[1] += acc4[u][k].y; o[2] += acc4[u][k].z; o[3] += acc4[u][k].w;
                        }
                        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
                    }
                    if (dx16) {
                        uint2 w;
                        w.x = pack_bf16x2(o[0], o[1]);
                        w.y = pack_bf16x2(o[2], o[3]);
                        *(uint2*)(dx16 + (boff + rr) * lddx16 + 4 * q) = w;
                    }
                }
            }
        }
    }
}

extern "C" int adap_groupnorm_bwd(const void* dy, int dy_dtype, long lddy, const float* x, long ldx,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  float* dx32, long lddx32, int accumulate, void* dx16, long lddx16,
                                  float* workspace, int B, int HW, int C, int act, void* stream) {
    ADAP_REQUIRE(dy && x && gamma && beta && mean && rstd && workspace && (dx32 || dx16), ADAP_ERR_SHAPE, "groupnorm_bwd: null pointer");
    ADAP_REQUIRE(C % GN_G == 0 && C % 4 == 0 && C <= 256 * 4 * GN_MAXSLOT, ADAP_ERR_SHAPE, "groupnorm_bwd: C=%d", C);
    ADAP_REQUIRE(dy_dtype == 0 || dy_dtype == 1, ADAP_ERR_UNSUPPORTED, "groupnorm_bwd: dy_dtype");
    ADAP_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0, ADAP_ERR_ALIGN, "groupnorm_bwd: ld alignment");
    int n, rpc;
    gn_chunks(HW, C, &n, &rpc);
    hipStream_t s = (hipStream_t)stream;
    if (dy_dtype == 1) {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<true>, dim3(n, B), dim3(256), 0, s, dy, lddy, x, ldx, HW, C, gamma, beta, mean,
                           rstd, act, rpc, workspace);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(n, B), dim3(256), 0, s, dy, lddy, x, ldx, HW, C, gamma, beta, mean,
                           rstd, act, workspace, n, rpc, dx32, lddx32, accumulate, (uint16_t*)dx16, lddx16);
    } else {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<false>, dim3(n, B), dim3(256), 0, s, dy, lddy, x, ldx, HW, C, gamma, beta, mean,
                           rstd, act, rpc, workspace);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(n, B), dim3(256), 0, s, dy, lddy, x, ldx, HW, C, gamma, beta, mean,
                           rstd, act, workspace, n, rpc, dx32, lddx32, accumulate, (uint16_t*)dx16, lddx16);
    }
    return adap_check_launch("groupnorm_bwd");
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
                                                     int accumulate, long rows, int D) {
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
        }
    }
}

extern "C" int adap_layernorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                                  const float* mean, const float* rstd, float* dx, long lddx, int accumulate,
                                  long rows, int D, void* stream) {
    ADAP_REQUIRE(dy && x && gamma && mean && rstd && dx, ADAP_ERR_SHAPE, "layernorm_bwd: null pointer");
    ADAP_REQUIRE(D % 4 == 0 && D <= 256 * LN_MAXV, ADAP_ERR_SHAPE, "layernorm_bwd: D=%d", D);
    if (rows == 0) return ADAP_OK;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx,
                       gamma, mean, rstd, dx, lddx, accumulate, rows, D);
    return adap_check_launch("layernorm_bwd");
}

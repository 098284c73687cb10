// Weight gradients (`unfreeze_model: True`, ddpm.py:775-786; SURVEY 8b "conv3x3_bwd_weight", "linear_bwd"):
//   dW[co][ci][ky][kx] = sum over output pixels m of dY[m][co] * X[pixel(m) shifted by (ky, kx)][ci]
// Both operands are pixel-major (K = pixels is the SLOW dimension of both), the opposite of what a matrix-core
// fragment wants (8 consecutive K values per lane).  Rather than a third contraction kernel family, the gradient is
// computed as   transpose -> the existing implicit-GEMM kernels -> scatter:
//   1. dYT [Cout][Mp]            bf16 = dY transposed                      (Mp = M rounded up to 8, zero tail)
//   2. XT  [taps*Cin][Mp]        bf16 = im2col(X) transposed: row (tap, ci), column m holds the input pixel that tap
//      sees from output pixel m (zero outside the image; stride 2 and the fused nearest upsample are just index maps)
//   3. T   [Cout][taps*Cin]      f32  = dYT . XT^T  through adap_conv2d_nhwc's 1x1 path (LDS-DMA ring kernels, split-K
//      over the pixels -- K is 256 ... 16384 here, the output only 0.1 - 30 M elements)
//   4. dW (+)= T permuted to [Cout][Cin][taps]       (1x1 / Linear: step 3 writes dW directly, step 4 is skipped)
// The transposes cost 2 B/element of dY and 2*taps B/element of X in HBM writes (<= 283 MB for the largest UNet layer,
// 960 -> 320 @64^2), against a contraction of 2*M*Cout*Cin*taps flops; they are HBM-bound kernels.
// Bias gradients, GroupNorm / LayerNorm affine gradients and the per-image channel sums the time-embedding
// projection needs are column reductions over pixel rows: two-stage, fixed order, fp64 finish (bit-reproducible).
#include "common.h"

extern "C" int adap_conv2d_nhwc(const void* x, int x_dtype, long ldx, const void* w_packed, const float* bias,
                                const float* chan_add, long ld_ca, const float* residual, long ldr, float* y32, long ldy32,
                                void* y16, long ldy16, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH,
                                int KW, int stride, int pad, int up, float alpha, int ksplit, float* splitk_ws, int nbatch,
                                long bs_x, long bs_w, long bs_y32, long bs_y16, void* stream);
extern "C" long adap_conv2d_workspace_floats(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW);

static inline long align256(long b) { return (b + 255) & ~255L; }

// ---------------------------------------------------------------------------------------------
// gather + transpose: dst[(tap*C + c)][m] = src[pixel seen by tap from output pixel m][c]  (bf16)
// grid (ceil(Mp/64), ceil(C/64), taps), block 256; 64 x 64 tile through LDS
// ---------------------------------------------------------------------------------------------
struct TrParams {
    const void* src;
    long ld;
    int B, Hin, Win, C, Hout, Wout, KW, stride, pad, up, vec;
    long M, Mp;
    uint16_t* dst;
};

template <bool F32>
__global__ __launch_bounds__(256) void wg_transpose_kernel(TrParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];
    const int t = threadIdx.x;
    const long m0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tap = blockIdx.z;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int HWo = p.Hout * p.Wout;
    const int Hs = p.up ? 2 * p.Hin : p.Hin, Ws = p.up ? 2 * p.Win : p.Win;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int idx = t + 256 * pass;
        const int row = idx >> 3, ch = idx & 7;
        const long m = m0 + row;
        const int c = c0 + ch * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (m < p.M && c < p.C) {
            const int b = (int)(m / HWo);
            const int rem = (int)(m - (long)b * HWo);
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const int uy = oy * p.stride + ky - p.pad, ux = ox * p.stride + kx - p.pad;
            if (uy >= 0 && uy < Hs && ux >= 0 && ux < Ws) {
                const int iy = p.up ? uy >> 1 : uy, ix = p.up ? ux >> 1 : ux;
                const long off = (((long)b * p.Hin + iy) * p.Win + ix) * p.ld + c;
                if (p.vec && c + 8 <= p.C) {
                    if (F32) {
                        const float4 a = *(const float4*)((const float*)p.src + off);
                        const float4 d = *(const float4*)((const float*)p.src + off + 4);
                        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
                    } else {
                        unpack_bf16x8(*(const uint4*)((const uint16_t*)p.src + off), v);
                    }
                } else {
                    for (int e = 0; e < 8 && c + e < p.C; ++e)
                        v[e] = F32 ? ((const float*)p.src)[off + e] : bf16_to_f32(((const uint16_t*)p.src)[off + e]);
                }
            }
        }
        *(uint4*)&tile[row][ch * 8] = pack_bf16x8(v);
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int idx = t + 256 * pass;
        const int c = idx >> 3, mc = idx & 7;
        if (c0 + c < p.C && m0 + mc * 8 < p.Mp) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = (uint32_t)tile[mc * 8 + 2 * j][c] | ((uint32_t)tile[mc * 8 + 2 * j + 1][c] << 16);
            *(uint4*)(p.dst + ((long)tap * p.C + c0 + c) * p.Mp + m0 + mc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

static int launch_transpose(const void* src, int dtype, long ld, int B, int Hin, int Win, int C, int Hout, int Wout, int KH,
                            int KW, int stride, int pad, int up, long M, long Mp, uint16_t* dst, hipStream_t s) {
    TrParams p;
    p.src = src; p.ld = ld; p.B = B; p.Hin = Hin; p.Win = Win; p.C = C; p.Hout = Hout; p.Wout = Wout; p.KW = KW;
    p.stride = stride; p.pad = pad; p.up = up; p.M = M; p.Mp = Mp; p.dst = dst;
    const int al = dtype == 0 ? 4 : 8;          // elements per 16 bytes
    p.vec = (C % 8 == 0) && (ld % al == 0) && (((uintptr_t)src % 16) == 0);
    dim3 grid((unsigned)((Mp + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)(KH * KW));
    if (dtype == 0) hipLaunchKernelGGL(wg_transpose_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(wg_transpose_kernel<false>, grid, dim3(256), 0, s, p);
    return adap_check_launch("wgrad transpose");
}

// dW[co][ci][tap] (+)= T[co][tap*Cin + ci]
__global__ __launch_bounds__(256) void wg_scatter_kernel(const float* __restrict__ T, float* __restrict__ dw, int Cout, int Cin,
                                                         int taps, int accumulate) {
    const long total = (long)Cout * Cin * taps;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int tap = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % Cin);
        const long co = r / Cin;
        const float v = T[co * (long)taps * Cin + (long)tap * Cin + ci];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// dW[co][ci][tap] (+)= Tt[tap*Cin + ci][co]: the contraction ran with the (tap, ci) index as its ROW dimension (see
// adap_conv2d_bwd_weight).  32 co x 32 ci x taps block through LDS: reads are runs of 32 co, writes runs of 32*taps.
__global__ __launch_bounds__(256) void wg_scatter_t_kernel(const float* __restrict__ Tt, float* __restrict__ dw, int Cout, int Cin,
                                                           int taps, int accumulate) {
    extern __shared__ float sc_tile[];              // [32 co][32 ci * taps + 1]
    const int pitch = 32 * taps + 1;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int t = threadIdx.x;
    for (int idx = t; idx < taps * 32 * 32; idx += 256) {
        const int cc = idx & 31;
        const int ii = (idx >> 5) & 31;
        const int tp = idx >> 10;
        float v = 0.f;
        if (co0 + cc < Cout && ci0 + ii < Cin) v = Tt[((long)tp * Cin + ci0 + ii) * Cout + co0 + cc];
        sc_tile[cc * pitch + ii * taps + tp] = v;
    }
    __syncthreads();
    const int run = 32 * taps;
    for (int idx = t; idx < 32 * run; idx += 256) {
        const int cc = idx / run, e = idx - cc * run;
        const int ii = e / taps;
        if (co0 + cc < Cout && ci0 + ii < Cin) {
            const long o = ((long)(co0 + cc) * Cin + ci0) * taps + e;
            const float v = sc_tile[cc * pitch + e];
            dw[o] = accumulate ? dw[o] + v : v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// column reductions over pixel rows.  rows are grouped in `nseg` segments of `seg_rows` (one segment = everything, or
// one image); per segment and channel:  s1 = sum dz,  s2 = sum dz * xhat   (kind < 0: plain column sum of dy, s1 only)
//   kind 0: GroupNorm(32) statistics, index (row / HW) * 32 + c / (C/32);  kind 1: LayerNorm statistics, index = row
//   act 1: dy is the gradient after SiLU, z = xhat*gamma + beta, dz = dy * silu'(z)
// stage 1: grid (ceil(C/64), nsplit, nseg), block 256 = 4 row lanes x 64 channels, partials [seg][split][2][C]
// stage 2: fp64 sum over the splits in fixed order
// ---------------------------------------------------------------------------------------------
struct RedParams {
    const void* dy; long lddy; int dyb;
    const void* x; long ldx; int xb;
    const float *gamma, *beta, *mean, *rstd;
    int kind, act, C, HW, nsplit;
    long seg_rows;
    float* part;
};

__device__ __forceinline__ float ld_elem(const void* p, int is_bf16, long off) {
    return is_bf16 ? bf16_to_f32(((const uint16_t*)p)[off]) : ((const float*)p)[off];
}

__global__ __launch_bounds__(256) void wg_colred_kernel(RedParams p) {
    __shared__ float red[2][4][64];
    const int t = threadIdx.x, cl = t & 63, rl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int split = blockIdx.y, seg = blockIdx.z;
    const long chunk = (p.seg_rows + p.nsplit - 1) / p.nsplit;
    const long r0 = (long)split * chunk, r1 = min(p.seg_rows, r0 + chunk);
    float s1 = 0.f, s2 = 0.f;
    if (c < p.C) {
        const int cpg = p.kind == 0 ? p.C / 32 : 1;
        const float ga = p.kind >= 0 ? p.gamma[c] : 0.f, be = (p.kind >= 0 && p.beta) ? p.beta[c] : 0.f;
        for (long r = r0 + rl; r < r1; r += 4) {
            const long row = (long)seg * p.seg_rows + r;
            float dz = ld_elem(p.dy, p.dyb, row * p.lddy + c);
            if (p.kind >= 0) {
                const long si = p.kind == 0 ? (row / p.HW) * 32 + c / cpg : row;
                const float xh = (ld_elem(p.x, p.xb, row * p.ldx + c) - p.mean[si]) * p.rstd[si];
                if (p.act) {
                    const float z = xh * ga + be;
                    const float sg = 1.0f / (1.0f + __expf(-z));
                    dz *= sg * (1.0f + z * (1.0f - sg));
                }
                s2 += dz * xh;
            }
            s1 += dz;
        }
    }
    red[0][rl][cl] = s1;
    red[1][rl][cl] = s2;
    __syncthreads();
    if (t < 128) {
        const int which = t >> 6;
        const float v = ((red[which][0][cl] + red[which][1][cl]) + red[which][2][cl]) + red[which][3][cl];
        if (c < p.C) p.part[(((long)seg * p.nsplit + split) * 2 + which) * p.C + c] = v;
    }
}

// The same reduction for 16-byte-aligned rows of C % 8 == 0 channels (every layer of the UNet but the 4-channel
// latent): a workgroup owns up to 512 channels x one row chunk, a thread 8 consecutive channels (one 16-B bf16 or
// two 16-B f32 loads per row) of every 4th row -- whole 1 KB row segments per wave instead of 4-byte elements.
template <bool DYB, bool XB>
__global__ __launch_bounds__(256) void wg_colred8_kernel(RedParams p) {
    __shared__ float red[4][64][17];
    const int t = threadIdx.x, oct = t & 63, rl = t >> 6;
    const int c = (blockIdx.x * 64 + oct) * 8;
    const int split = blockIdx.y, seg = blockIdx.z;
    const long chunk = (p.seg_rows + p.nsplit - 1) / p.nsplit;
    const long r0 = (long)split * chunk, r1 = min(p.seg_rows, r0 + chunk);
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (c < p.C) {
        float ga[8], be[8], mu[8], rs[8];
        const int cpg = p.kind == 0 ? p.C / 32 : 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ga[e] = p.kind >= 0 ? p.gamma[c + e] : 0.f;
            be[e] = (p.kind >= 0 && p.beta) ? p.beta[c + e] : 0.f;
            mu[e] = 0.f;
            rs[e] = 0.f;
        }
        long last_b = -1;
#pragma unroll 2
        for (long r = r0 + rl; r < r1; r += 4) {
            const long row = (long)seg * p.seg_rows + r;
            float dz[8];
            if (DYB) {
                unpack_bf16x8(*(const uint4*)((const uint16_t*)p.dy + row * p.lddy + c), dz);
            } else {
                const float4 a = *(const float4*)((const float*)p.dy + row * p.lddy + c);
                const float4 d = *(const float4*)((const float*)p.dy + row * p.lddy + c + 4);
                dz[0] = a.x; dz[1] = a.y; dz[2] = a.z; dz[3] = a.w; dz[4] = d.x; dz[5] = d.y; dz[6] = d.z; dz[7] = d.w;
            }
            if (p.kind >= 0) {
                float xv[8];
                if (XB) {
                    unpack_bf16x8(*(const uint4*)((const uint16_t*)p.x + row * p.ldx + c), xv);
                } else {
                    const float4 a = *(const float4*)((const float*)p.x + row * p.ldx + c);
                    const float4 d = *(const float4*)((const float*)p.x + row * p.ldx + c + 4);
                    xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = d.x; xv[5] = d.y; xv[6] = d.z; xv[7] = d.w;
                }
                if (p.kind == 0) {
                    const long b = row / p.HW;
                    if (b != last_b) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const long si = b * 32 + (c + e) / cpg;
                            mu[e] = p.mean[si];
                            rs[e] = p.rstd[si];
                        }
                        last_b = b;
                    }
                } else {
                    const float m = p.mean[row], q = p.rstd[row];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { mu[e] = m; rs[e] = q; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[e] - mu[e]) * rs[e];
                    if (p.act) {
                        const float z = xh * ga[e] + be[e];
                        const float sg = 1.0f / (1.0f + __expf(-z));
                        dz[e] *= sg * (1.0f + z * (1.0f - sg));
                    }
                    s2[e] += dz[e] * xh;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[e] += dz[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[rl][oct][e] = s1[e];
        red[rl][oct][8 + e] = s2[e];
    }
    __syncthreads();
    for (int idx = t; idx < 64 * 16; idx += 256) {
        const int o = idx >> 4, k = idx & 15;
        const int cc = (blockIdx.x * 64 + o) * 8 + (k & 7);
        if (cc < p.C) {
            const float v = ((red[0][o][k] + red[1][o][k]) + red[2][o][k]) + red[3][o][k];
            p.part[(((long)seg * p.nsplit + split) * 2 + (k >> 3)) * p.C + cc] = v;
        }
    }
}

// out1[seg][c] (+)= sum_split part[seg][split][0][c]; out2 likewise from [1] (either may be NULL).
// grid (ceil(C/64), nseg), block 256 = 4 split lanes x 64 channels: lane l sums splits l, l+4, ... in fp64, the four
// lane sums are added in fixed order.
__global__ __launch_bounds__(256) void wg_colred_finish_kernel(const float* __restrict__ part, float* __restrict__ out1,
                                                               float* __restrict__ out2, int C, int nsplit, int nseg,
                                                               int accumulate) {
    __shared__ double red[2][4][64];
    const int t = threadIdx.x, cl = t & 63, sl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long seg = blockIdx.y;
    double a = 0.0, b = 0.0;
    if (c < C) {
        for (int s = sl; s < nsplit; s += 4) {
            a += (double)part[((seg * nsplit + s) * 2 + 0) * C + c];
            b += (double)part[((seg * nsplit + s) * 2 + 1) * C + c];
        }
    }
    red[0][sl][cl] = a;
    red[1][sl][cl] = b;
    __syncthreads();
    if (t < 128 && c < C) {
        const int which = t >> 6;
        const double v = ((red[which][0][cl] + red[which][1][cl]) + red[which][2][cl]) + red[which][3][cl];
        float* out = which ? out2 : out1;
        const long i = seg * C + c;
        if (out) out[i] = accumulate ? out[i] + (float)v : (float)v;
    }
}

static int colred_nsplit(long seg_rows) {
    const long per = seg_rows <= 8192 ? 64 : 128;      // rows per workgroup (16 / 32 per row lane)
    long n = (seg_rows + per - 1) / per;
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    return (int)n;
}

extern "C" long adap_colsum_workspace_floats(long rows, long seg_rows, int C) {
    if (seg_rows <= 0 || rows <= 0) return 0;
    return (rows / seg_rows) * colred_nsplit(seg_rows) * 2L * C;
}

static int launch_colred(RedParams p, long rows, long seg_rows, float* out1, float* out2, int accumulate, float* workspace,
                         hipStream_t s) {
    const int nseg = (int)(rows / seg_rows);
    p.seg_rows = seg_rows;
    p.nsplit = colred_nsplit(seg_rows);
    p.part = workspace;
    const int ady = p.dyb ? 8 : 4, ax = p.xb ? 8 : 4;
    const bool vec = p.C % 8 == 0 && p.lddy % ady == 0 && ((uintptr_t)p.dy % 16) == 0 &&
                     (p.kind < 0 || (p.ldx % ax == 0 && ((uintptr_t)p.x % 16) == 0));
    if (vec) {
        dim3 grid((p.C / 8 + 63) / 64, p.nsplit, nseg);
        if (p.dyb && p.xb) hipLaunchKernelGGL((wg_colred8_kernel<true, true>), grid, dim3(256), 0, s, p);
        else if (p.dyb) hipLaunchKernelGGL((wg_colred8_kernel<true, false>), grid, dim3(256), 0, s, p);
        else if (p.xb) hipLaunchKernelGGL((wg_colred8_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wg_colred8_kernel<false, false>), grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(wg_colred_kernel, dim3((p.C + 63) / 64, p.nsplit, nseg), dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(wg_colred_finish_kernel, dim3((p.C + 63) / 64, nseg), dim3(256), 0, s, workspace, out1, out2, p.C,
                       p.nsplit, nseg, accumulate);
    return adap_check_launch("wgrad column reduction");
}

// out[seg][c] (+)= sum over the rows of segment seg of dy[row][c]: bias gradients (seg_rows = rows) and the per-image
// channel sums that are the gradient of a ResBlock's time-embedding projection output (seg_rows = H*W)
extern "C" int adap_colsum(const void* dy, int dy_dtype, long lddy, long rows, long seg_rows, int C, float* out,
                           int accumulate, float* workspace, void* stream) {
    ADAP_REQUIRE(dy && out && workspace, ADAP_ERR_SHAPE, "colsum: null pointer");
    ADAP_REQUIRE(dy_dtype == 0 || dy_dtype == 1, ADAP_ERR_UNSUPPORTED, "colsum: dtype %d", dy_dtype);
    ADAP_REQUIRE(rows > 0 && seg_rows > 0 && rows % seg_rows == 0 && C > 0 && lddy >= C, ADAP_ERR_SHAPE,
                 "colsum: rows=%ld seg_rows=%ld C=%d ld=%ld", rows, seg_rows, C, lddy);
    ADAP_REQUIRE(rows / seg_rows <= 65535, ADAP_ERR_SHAPE, "colsum: too many segments");
    RedParams p = {};
    p.dy = dy; p.lddy = lddy; p.dyb = dy_dtype; p.kind = -1; p.C = C; p.HW = 1;
    return launch_colred(p, rows, seg_rows, out, nullptr, accumulate, workspace, (hipStream_t)stream);
}

// dgamma[c] (+)= sum dz * xhat, dbeta[c] (+)= sum dz over all rows; kind 0 GroupNorm32 (mean/rstd [B][32], rows = B*HW),
// kind 1 LayerNorm (mean/rstd [rows]).  workspace: adap_colsum_workspace_floats(rows, rows, C) floats.
extern "C" int adap_norm_affine_bwd(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                                    const float* gamma, const float* beta, const float* mean, const float* rstd, int kind,
                                    int act, float* dgamma, float* dbeta, int accumulate, float* workspace, long rows, int HW,
                                    int C, void* stream) {
    ADAP_REQUIRE(dy && x && gamma && mean && rstd && workspace && (dgamma || dbeta), ADAP_ERR_SHAPE,
                 "norm_affine_bwd: null pointer");
    ADAP_REQUIRE((dy_dtype == 0 || dy_dtype == 1) && (x_dtype == 0 || x_dtype == 1), ADAP_ERR_UNSUPPORTED,
                 "norm_affine_bwd: dtype");
    ADAP_REQUIRE(kind == 0 || kind == 1, ADAP_ERR_UNSUPPORTED, "norm_affine_bwd: kind %d", kind);
    ADAP_REQUIRE(rows > 0 && C > 0 && lddy >= C && ldx >= C && HW > 0, ADAP_ERR_SHAPE, "norm_affine_bwd: shape");
    ADAP_REQUIRE(kind != 0 || (C % 32 == 0 && rows % HW == 0), ADAP_ERR_SHAPE, "norm_affine_bwd: GroupNorm32 needs C %% 32 == 0");
    ADAP_REQUIRE(!act || beta, ADAP_ERR_SHAPE, "norm_affine_bwd: act needs beta");
    RedParams p = {};
    p.dy = dy; p.lddy = lddy; p.dyb = dy_dtype; p.x = x; p.ldx = ldx; p.xb = x_dtype;
    p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd; p.kind = kind; p.act = act; p.C = C; p.HW = HW;
    return launch_colred(p, rows, rows, dbeta, dgamma, accumulate, workspace, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// conv / linear weight gradient
// ---------------------------------------------------------------------------------------------
struct WgPlan {
    long M, Mp, off_dyt, off_xt, off_t, off_sk, off_cs, total, skf;
    int taps, N, swap;
};

static WgPlan wg_plan(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW, int want_bias) {
    WgPlan q;
    q.taps = KH * KW;
    q.M = (long)B * Hout * Wout;
    q.Mp = (q.M + 7) & ~7L;
    q.N = q.taps * Cin;
    // Which operand supplies the contraction's ROWS: the implicit-GEMM family has its 256-row LDS-DMA ring kernel for
    // >= 4096 rows.  Cout is at most 1280 (10240 for the GEGLU projection), taps*Cin reaches 23040 -- so when that side
    // is the long one it becomes the row dimension and the result comes out transposed ([tap*Cin + ci][co]).
    q.swap = (q.N >= 4096 && q.N > Cout && Cout % 4 == 0) ? 1 : 0;
    long o = 0;
    q.off_dyt = o; o += align256(2L * Cout * q.Mp);
    q.off_xt = o; o += align256(2L * q.N * q.Mp);
    q.off_t = o; o += (q.taps > 1 || q.swap) ? align256(4L * Cout * q.N) : 0;
    q.skf = q.swap ? adap_conv2d_workspace_floats(1, q.N, 1, (int)q.Mp, Cout, 1, 1)
                   : adap_conv2d_workspace_floats(1, Cout, 1, (int)q.Mp, q.N, 1, 1);
    q.off_sk = o; o += align256(4L * q.skf);
    q.off_cs = o; o += want_bias ? align256(4L * adap_colsum_workspace_floats(q.M, q.M, Cout)) : 0;
    q.total = o;
    return q;
}

extern "C" long adap_conv2d_bwd_weight_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW) {
    if ((long)B * Hout * Wout >= (1L << 31) - 8) return -1;
    return wg_plan(B, Hout, Wout, Cin, Cout, KH, KW, 1).total;
}

// dw f32 [Cout][Cin][KH][KW] (nn.Conv2d OIHW; KH = KW = 1 with Hout = rows, Wout = 1: nn.Linear [out][in]),
// dbias f32 [Cout] or NULL.  x [B][Hin][Win][Cin] (row pitch ldx), dy [B][Hout][Wout][Cout] (row pitch lddy), each f32
// or bf16 (the contraction runs on bf16 operands with f32 accumulation, like the forward).  stride / pad / up as in
// adap_conv2d_nhwc (up = 1: nearest x2 fused in front of the conv).  accumulate: add to dw / dbias instead of overwriting.
extern "C" int adap_conv2d_bwd_weight(const void* x, int x_dtype, long ldx, const void* dy, int dy_dtype, long lddy,
                                      float* dw, float* dbias, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                                      int KH, int KW, int stride, int pad, int up, int accumulate, void* workspace,
                                      long workspace_bytes, void* stream) {
    ADAP_REQUIRE(x && dy && (dw || dbias) && workspace, ADAP_ERR_SHAPE, "conv2d_bwd_weight: null pointer");
    ADAP_REQUIRE((x_dtype == 0 || x_dtype == 1) && (dy_dtype == 0 || dy_dtype == 1), ADAP_ERR_UNSUPPORTED,
                 "conv2d_bwd_weight: dtype");
    ADAP_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Cin > 0 && Cout > 0, ADAP_ERR_SHAPE,
                 "conv2d_bwd_weight: non-positive dims");
    ADAP_REQUIRE((KH == 1 && KW == 1) || (KH == 3 && KW == 3), ADAP_ERR_UNSUPPORTED, "conv2d_bwd_weight: kernel %dx%d", KH, KW);
    ADAP_REQUIRE(stride == 1 || stride == 2, ADAP_ERR_UNSUPPORTED, "conv2d_bwd_weight: stride %d", stride);
    ADAP_REQUIRE(up == 0 || up == 1, ADAP_ERR_UNSUPPORTED, "conv2d_bwd_weight: up %d", up);
    ADAP_REQUIRE(ldx >= Cin && lddy >= Cout, ADAP_ERR_SHAPE, "conv2d_bwd_weight: row pitch");
    ADAP_REQUIRE((KH * KW * Cin) % 4 == 0, ADAP_ERR_ALIGN, "conv2d_bwd_weight: taps*Cin=%d must be a multiple of 4", KH * KW * Cin);
    ADAP_REQUIRE(((uintptr_t)workspace % 256) == 0 && (!dw || ((uintptr_t)dw % 16) == 0), ADAP_ERR_ALIGN,
                 "conv2d_bwd_weight: workspace must be 256-byte, dw 16-byte aligned");
    ADAP_REQUIRE((long)B * Hout * Wout < (1L << 31) - 8, ADAP_ERR_SHAPE, "conv2d_bwd_weight: too many pixels");
    {   // the output grid must be the one the forward conv produces
        const int Hs = up ? 2 * Hin : Hin, Ws = up ? 2 * Win : Win;
        ADAP_REQUIRE((Hout - 1) * stride + KH - pad <= Hs + pad + 1 && (Wout - 1) * stride + KW - pad <= Ws + pad + 1,
                     ADAP_ERR_SHAPE, "conv2d_bwd_weight: output %dx%d does not fit input %dx%d", Hout, Wout, Hs, Ws);
    }
    const WgPlan q = wg_plan(B, Hout, Wout, Cin, Cout, KH, KW, dbias != nullptr);
    ADAP_REQUIRE(workspace_bytes >= q.total, ADAP_ERR_SHAPE, "conv2d_bwd_weight: workspace %ld < %ld bytes", workspace_bytes,
                 q.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int rc;
    if (dbias) {
        rc = adap_colsum(dy, dy_dtype, lddy, q.M, q.M, Cout, dbias, accumulate, (float*)(ws + q.off_cs), stream);
        if (rc) return rc;
    }
    if (!dw) return ADAP_OK;
    uint16_t* dyt = (uint16_t*)(ws + q.off_dyt);
    uint16_t* xt = (uint16_t*)(ws + q.off_xt);
    rc = launch_transpose(dy, dy_dtype, lddy, B, Hout, Wout, Cout, Hout, Wout, 1, 1, 1, 0, 0, q.M, q.Mp, dyt, s);
    if (rc) return rc;
    rc = launch_transpose(x, x_dtype, ldx, B, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, up, q.M, q.Mp, xt, s);
    if (rc) return rc;
    float* skws = q.skf > 0 ? (float*)(ws + q.off_sk) : nullptr;
    if (q.swap) {
        float* Tt = (float*)(ws + q.off_t);
        rc = adap_conv2d_nhwc(xt, 1, q.Mp, dyt, nullptr, nullptr, 0, nullptr, 0, Tt, Cout, nullptr, 0, 1, q.N, 1, (int)q.Mp, q.N,
                              1, Cout, 1, 1, 1, 0, 0, 1.0f, 0, skws, 1, 0, 0, 0, 0, stream);
        if (rc) return rc;
        dim3 grid((Cin + 31) / 32, (Cout + 31) / 32);
        hipLaunchKernelGGL(wg_scatter_t_kernel, grid, dim3(256), (size_t)32 * (32 * q.taps + 1) * sizeof(float), s, Tt, dw, Cout,
                           Cin, q.taps, accumulate);
        return adap_check_launch("wgrad scatter");
    }
    float* T = q.taps > 1 ? (float*)(ws + q.off_t) : dw;
    const float* residual = (q.taps == 1 && accumulate) ? dw : nullptr;
    rc = adap_conv2d_nhwc(dyt, 1, q.Mp, xt, nullptr, nullptr, 0, residual, q.N, T, q.N, nullptr, 0, 1, Cout, 1, (int)q.Mp, Cout,
                          1, q.N, 1, 1, 1, 0, 0, 1.0f, 0, skws, 1, 0, 0, 0, 0, stream);
    if (rc) return rc;
    if (q.taps > 1) {
        const long total = (long)Cout * q.N;
        long g = (total + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(wg_scatter_kernel, dim3((unsigned)g), dim3(256), 0, s, T, dw, Cout, Cin, q.taps, accumulate);
        return adap_check_launch("wgrad scatter");
    }
    return ADAP_OK;
}

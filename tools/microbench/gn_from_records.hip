// GroupNorm(32) + SiLU on a UNet-sized tensor when the statistics already exist as per-64-pixel records (what a producing
// contraction's epilogue can leave, csrc/conv_gemm.hip: adap_conv2d_next_gn_partial) -- how fast is the pass that remains, with
// the records' reduction folded into it (no finish launch)?  Standalone, self-checking (host fp64), timed with HIP events beside
// a plain "read f32, write bf16" pass over the same bytes.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gn_from_records tools/microbench/gn_from_records.hip && /tmp/gn_from_records
//
//   x f32 [B][HW][C] (NHWC), records f32 [B][HW/64][32][2] = (sum, sum of squares) of a 64-pixel run and group,
//   y bf16 [B][HW][C] = silu((x - mean_g) * rstd_g * gamma_c + beta_c)
//
// The decision it informs: the library's single-launch GroupNorm spends ~4 of its 12.7 us (320 ch @ 64 x 64,
// bs 4: 31 MB) on the statistics exchange between a sample's workgroups; with records from the producer that exchange and the
// statistics arithmetic go away and what is left is this kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define GROUPS 32
#define RUN 64            // pixels per record
#define MAXC 1280

__device__ __forceinline__ uint32_t pack2(float a, float b) {          // two floats -> two bf16 (round to nearest even)
    uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ float silu(float v) { return v * __frcp_rn(1.f + __expf(-v)); }

// the upper bound: the same loads and stores, no statistics, no per-channel affine
__global__ __launch_bounds__(256) void copy_cast(const float* __restrict__ x, uint16_t* __restrict__ y, long octs) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < octs; i += (long)gridDim.x * 256) {
        const float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
        uint4 o;
        o.x = pack2(a.x, a.y); o.y = pack2(a.z, a.w); o.z = pack2(b.x, b.y); o.w = pack2(b.z, b.w);
        ((uint4*)y)[i] = o;
    }
}

// grid (pixel chunks, B): every workgroup reduces its sample's records itself (HW/64 x 32 x 2 floats, L2-resident after the
// first workgroup; fp64, fixed order), forms the per-channel affine in LDS, then streams PIX pixels.
// SPREAD = false: 32 threads each walk all of a group's records (a chain of HW/64 dependent-latency steps: measured 19 of the
// kernel's 23 us at 64 x 64).  SPREAD = true: the records dealt over 4 slices x (sum, sum of squares) x 32 groups = 256 threads,
// 16 independent loads each, combined through LDS in a fixed order.
template <int PIX, bool SPREAD>
__global__ __launch_bounds__(256) void gn_apply_from_records(const float* __restrict__ x, const float* __restrict__ rec,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             uint16_t* __restrict__ y, int HW, int C, float eps) {
    __shared__ float sMean[GROUPS], sRstd[GROUPS];
    __shared__ float sA[MAXC], sB[MAXC];
    __shared__ double sPart[4][2 * GROUPS];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int runs = HW / RUN, cpg = C / GROUPS;
    if (SPREAD) {
        const int which = tid & (2 * GROUPS - 1), slice = tid >> 6;        // which = 2 * group + (0: sum, 1: sum of squares)
        const float* r = rec + (size_t)b * runs * GROUPS * 2 + which;     // a record row is 64 consecutive floats: coalesced
        double a = 0.0;
#pragma unroll 8
        for (int i = slice; i < runs; i += 4) a += (double)r[(size_t)i * GROUPS * 2];
        sPart[slice][which] = a;
        __syncthreads();
        if (tid < GROUPS) {
            const double s = ((sPart[0][2 * tid] + sPart[1][2 * tid]) + sPart[2][2 * tid]) + sPart[3][2 * tid];
            const double ss = ((sPart[0][2 * tid + 1] + sPart[1][2 * tid + 1]) + sPart[2][2 * tid + 1]) + sPart[3][2 * tid + 1];
            const double n = (double)HW * cpg, mean = s / n;
            double var = ss / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            sMean[tid] = (float)mean;
            sRstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
        }
    } else if (tid < GROUPS) {
        double s = 0.0, ss = 0.0;
        const float* r = rec + ((size_t)b * runs * GROUPS + tid) * 2;
        for (int i = 0; i < runs; ++i) {
            s += (double)r[(size_t)i * GROUPS * 2];
            ss += (double)r[(size_t)i * GROUPS * 2 + 1];
        }
        const double n = (double)HW * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sMean[tid] = (float)mean;
        sRstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float a = sRstd[g] * gamma[c];
        sA[c] = a;
        sB[c] = beta[c] - sMean[g] * a;
    }
    __syncthreads();
    const int octs_per_pix = C >> 3;
    const long p0 = (long)blockIdx.x * PIX;
    const int npix = (int)min((long)PIX, (long)HW - p0);
    const float* xb = x + ((size_t)b * HW + p0) * C;
    uint16_t* yb = y + ((size_t)b * HW + p0) * C;
    const int total = npix * octs_per_pix;
    for (int i = tid; i < total; i += 256) {
        const int o = i % octs_per_pix, c0 = 8 * o;                     // consecutive threads: consecutive 32-byte pieces
        const float4 u = ((const float4*)xb)[2 * i], v = ((const float4*)xb)[2 * i + 1];
        uint4 out;
        out.x = pack2(silu(fmaf(u.x, sA[c0 + 0], sB[c0 + 0])), silu(fmaf(u.y, sA[c0 + 1], sB[c0 + 1])));
        out.y = pack2(silu(fmaf(u.z, sA[c0 + 2], sB[c0 + 2])), silu(fmaf(u.w, sA[c0 + 3], sB[c0 + 3])));
        out.z = pack2(silu(fmaf(v.x, sA[c0 + 4], sB[c0 + 4])), silu(fmaf(v.y, sA[c0 + 5], sB[c0 + 5])));
        out.w = pack2(silu(fmaf(v.z, sA[c0 + 6], sB[c0 + 6])), silu(fmaf(v.w, sA[c0 + 7], sB[c0 + 7])));
        ((uint4*)yb)[i] = out;
    }
}

static float bf2f(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

template <class F>
static float time_us(F launch, int reps = 100) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return 1e3f * ms / reps;
}

static int run_shape(int B, int HW, int C) {
    const float eps = 1e-5f;
    const int runs = HW / RUN, cpg = C / GROUPS;
    const size_t n = (size_t)B * HW * C;
    std::vector<float> hx(n), hg(C), hb(C), hrec((size_t)B * runs * GROUPS * 2, 0.f);
    for (auto& v : hx) v = 1.5f * frand() + 0.3f;
    for (int c = 0; c < C; ++c) { hg[c] = 1.f + 0.2f * frand(); hb[c] = 0.1f * frand(); }
    std::vector<double> mean((size_t)B * GROUPS, 0.0), rstd((size_t)B * GROUPS);
    for (int b = 0; b < B; ++b) {
        std::vector<double> s(GROUPS, 0.0), ss(GROUPS, 0.0);
        for (int p = 0; p < HW; ++p)
            for (int c = 0; c < C; ++c) {
                const double v = hx[((size_t)b * HW + p) * C + c];
                const int g = c / cpg;
                s[g] += v;
                ss[g] += v * v;
                float* r = &hrec[(((size_t)b * runs + p / RUN) * GROUPS + g) * 2];
                r[0] += (float)v;                        // (the epilogue forms these in fp32 as well)
                r[1] += (float)(v * v);
            }
        for (int g = 0; g < GROUPS; ++g) {
            const double cnt = (double)HW * cpg, m = s[g] / cnt;
            mean[(size_t)b * GROUPS + g] = m;
            rstd[(size_t)b * GROUPS + g] = 1.0 / sqrt(ss[g] / cnt - m * m + eps);
        }
    }
    float *dx, *dg, *db, *drec;
    uint16_t* dy;
    CK(hipMalloc(&dx, n * 4));
    CK(hipMalloc(&dy, n * 2));
    CK(hipMalloc(&dg, C * 4));
    CK(hipMalloc(&db, C * 4));
    CK(hipMalloc(&drec, hrec.size() * 4));
    CK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drec, hrec.data(), hrec.size() * 4, hipMemcpyHostToDevice));
    const double mb = n * 6 / 1e6;
    printf("B %d HW %d C %d: %.1f MB (read f32 + write bf16)\n", B, HW, C, mb);
    const long octs = (long)(n / 8);
    long cg = (octs + 255) / 256;
    if (cg > 4096) cg = 4096;
    float us = time_us([&] { hipLaunchKernelGGL(copy_cast, dim3((unsigned)cg), dim3(256), 0, 0, dx, dy, octs); });
    printf("  %-26s %7.2f us  %6.0f GB/s\n", "copy + cast (bound)", us, mb / us * 1e3);
    int bad = 0;
    auto check = [&](const char* name, float t) {
        std::vector<uint16_t> hy(n);
        CK(hipMemcpy(hy.data(), dy, n * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < n; i += 7) {             // every 7th element: all channels and groups are visited
            const int c = (int)(i % C), b = (int)(i / ((size_t)HW * C)), g = c / cpg;
            const double v = (hx[i] - mean[(size_t)b * GROUPS + g]) * rstd[(size_t)b * GROUPS + g] * hg[c] + hb[c];
            const double ref = v / (1.0 + exp(-v));
            const double err = fabs(bf2f(hy[i]) - ref) / (fabs(ref) + 1e-2);
            if (err > worst) worst = err;
        }
        const bool ok = worst < 1.2e-2;                 // bf16 output: 2^-8 relative, plus the fp32 records' rounding
        printf("  %-26s %7.2f us  %6.0f GB/s   worst rel err %.2e %s\n", name, t, mb / t * 1e3, worst, ok ? "ok" : "MISMATCH");
        if (!ok) ++bad;
    };
    CK(hipMemset(dy, 0, n * 2));
    us = time_us([&] { hipLaunchKernelGGL((gn_apply_from_records<32, false>), dim3((HW + 31) / 32, B), dim3(256), 0, 0, dx, drec, dg, db, dy, HW, C, eps); });
    check("records, serial, 32 px", us);
    CK(hipMemset(dy, 0, n * 2));
    us = time_us([&] { hipLaunchKernelGGL((gn_apply_from_records<32, true>), dim3((HW + 31) / 32, B), dim3(256), 0, 0, dx, drec, dg, db, dy, HW, C, eps); });
    check("records, spread, 32 px", us);
    CK(hipMemset(dy, 0, n * 2));
    us = time_us([&] { hipLaunchKernelGGL((gn_apply_from_records<64, true>), dim3((HW + 63) / 64, B), dim3(256), 0, 0, dx, drec, dg, db, dy, HW, C, eps); });
    check("records, spread, 64 px", us);
    CK(hipMemset(dy, 0, n * 2));
    us = time_us([&] { hipLaunchKernelGGL((gn_apply_from_records<16, true>), dim3((HW + 15) / 16, B), dim3(256), 0, 0, dx, drec, dg, db, dy, HW, C, eps); });
    check("records, spread, 16 px", us);
    CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dg)); CK(hipFree(db)); CK(hipFree(drec));
    return bad;
}

int main() {
    srand(11);
    int bad = 0;
    bad += run_shape(4, 4096, 320);       // the north star's shape: 320 ch @ 64 x 64, bs 4 (the library's single launch: 12.7 us)
    bad += run_shape(4, 4096, 640);       // decoder side, after the concat
    bad += run_shape(4, 1024, 640);       // 32 x 32
    bad += run_shape(4, 256, 1280);       // 16 x 16
    printf(bad ? "FAILED: %d mismatches\n" : "all ok\n", bad);
    return bad ? 1 : 0;
}

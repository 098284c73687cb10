// The token-map kernels of the recon iteration (csrc/attention.hip: attn_tokmap_fwd_kernel, attn_tokmap_kw_kernel,
// attn_tokmap_bwd_gq_kernel) beside reformulations that spread the work differently -- standalone, self-checking (host
// reference in fp64), timed with HIP events.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tokmap_family tools/microbench/tokmap_family.hip && /tmp/tokmap_family
//
//   kw[bh][g][c]   = sum_m w[b][m][g] k[b][m][head*d + c]                 (G*d numbers per (batch, head), M = 77 keys)
//   T[b][h][n][g]  = scale * <q[b][n][head*d + :], kw[bh][g][:]>          (the forward: token maps without score rows)
//   part[bh][chunk][g][c] = sum_{n in chunk} dT[b][h][n][g] q[b][n][head*d + c]     (stage 1 of gq, 128-row chunks)
//
// "cur"    = the kernels as they are in the library (every thread walks all keys for its (g, c); one bf16 per lane in gq)
// "skip"   = kw with the zero weights skipped (the library's form before DESIGN 8 r3-k)
// "spread" = kw with the keys dealt over KS thread slices and one LDS reduction
// "oct"    = gq with 16-byte loads: d/8 lanes per row, 256/(d/8) rows at a time, one LDS reduction over the row lanes
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define ROWS 256          // query rows per workgroup in the forward
#define CHUNK 128         // query rows per workgroup in gq stage 1
#define MAXP 640          // G <= 4, d <= 160

__device__ __forceinline__ float bf(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

// ---------------------------------------------------------------------------------------------- kw
template <bool SKIP>
__global__ __launch_bounds__(256) void kw_cur(const float* __restrict__ w, const uint16_t* __restrict__ k, long ldk,
                                              float* __restrict__ kw, int H, int M, int d, int G) {
    const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
    for (int idx = threadIdx.x; idx < G * d; idx += 256) {
        const int g = idx / d, c = idx - g * d;
        float a = 0.f;
        if (SKIP) {
            for (int m = 0; m < M; ++m) {
                const float wv = w[((size_t)b * M + m) * G + g];
                if (wv != 0.f) a += wv * bf(k[((size_t)b * M + m) * ldk + head * d + c]);
            }
        } else {
            const float* wp = w + (size_t)b * M * G + g;
            const uint16_t* kp = k + (size_t)b * M * ldk + head * d + c;
#pragma unroll 8
            for (int m = 0; m < M; ++m) a = fmaf(wp[(size_t)m * G], bf(kp[(size_t)m * ldk]), a);
        }
        kw[(size_t)bh * G * d + idx] = a;
    }
}

// keys dealt over KS slices: item (s, p) sums the keys m = s, s + KS, ...; then a fixed-order sum over the slices
__device__ __forceinline__ void kw_spread_phase(float* part, float* out_lds, const float* __restrict__ w,
                                                const uint16_t* __restrict__ k, long ldk, int b, int head, int M, int d, int G,
                                                float scale) {
    const int P = G * d;
    int KS = 512 / P;
    KS = KS < 1 ? 1 : (KS > 8 ? 8 : KS);
    for (int idx = threadIdx.x; idx < P * KS; idx += 256) {
        const int s = idx / P, p = idx - s * P, g = p / d, c = p - g * d;
        float a = 0.f;
        for (int m = s; m < M; m += KS)
            a = fmaf(w[((size_t)b * M + m) * G + g], bf(k[((size_t)b * M + m) * ldk + head * d + c]), a);
        part[idx] = a;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) {
        float v = 0.f;
        for (int s = 0; s < KS; ++s) v += part[s * P + p];
        out_lds[p] = v * scale;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void kw_spread(const float* __restrict__ w, const uint16_t* __restrict__ k, long ldk,
                                                 float* __restrict__ kw, int H, int M, int d, int G) {
    __shared__ float part[8 * MAXP];
    __shared__ float res[MAXP];
    const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
    kw_spread_phase(part, res, w, k, ldk, b, head, M, d, G, 1.f);
    for (int p = threadIdx.x; p < G * d; p += 256) kw[(size_t)bh * G * d + p] = res[p];
}

// ---------------------------------------------------------------------------------------------- forward
template <bool SPREAD>
__global__ __launch_bounds__(256) void tokmap_fwd(const uint16_t* __restrict__ q, long ldq, const uint16_t* __restrict__ k,
                                                  long ldk, const float* __restrict__ w, float* __restrict__ T, int G, int H,
                                                  int N, int M, int d, float scale) {
    __shared__ float part[SPREAD ? 8 * MAXP : 1];
    __shared__ float sKW[MAXP];
    const int tid = threadIdx.x;
    const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
    if (SPREAD) {
        kw_spread_phase(part, sKW, w, k, ldk, b, head, M, d, G, scale);
    } else {
        for (int idx = tid; idx < G * d; idx += 256) {
            const int g = idx / d, c = idx - g * d;
            float a = 0.f;
            const float* wp = w + (size_t)b * M * G + g;
            const uint16_t* kp = k + (size_t)b * M * ldk + head * d + c;
#pragma unroll 8
            for (int m = 0; m < M; ++m) a = fmaf(wp[(size_t)m * G], bf(kp[(size_t)m * ldk]), a);
            sKW[idx] = a * scale;
        }
        __syncthreads();
    }
    const int n = blockIdx.x * ROWS + tid;
    if (n >= N) return;
    const uint16_t* qr = q + ((size_t)b * N + n) * ldq + head * d;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < d; c += 4) {
        const uint2 raw = *(const uint2*)(qr + c);
        const float x0 = __builtin_bit_cast(float, raw.x << 16), x1 = __builtin_bit_cast(float, raw.x & 0xffff0000u);
        const float x2 = __builtin_bit_cast(float, raw.y << 16), x3 = __builtin_bit_cast(float, raw.y & 0xffff0000u);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < G) {
                const float* kwp = sKW + g * d + c;
                acc[g] += (x0 * kwp[0] + x1 * kwp[1]) + (x2 * kwp[2] + x3 * kwp[3]);
            }
    }
    float* out = T + ((size_t)bh * N + n) * G;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g < G) out[g] = acc[g];
}

// ---------------------------------------------------------------------------------------------- gq stage 1
__global__ __launch_bounds__(256) void gq_cur(const float* __restrict__ dt, const uint16_t* __restrict__ q, long ldq,
                                              float* __restrict__ part, int B, int H, int N, int d, int G) {
    __shared__ float red[4][4 * 3][64];
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
    const int n0 = blockIdx.x * CHUNK;
    const int rows = min(CHUNK, N - n0);
    float acc[4][3];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[g][j] = 0.f;
    const float* dtb = dt + (((size_t)b * H + head) * N + n0) * G;
    const uint16_t* qb = q + ((size_t)b * N + n0) * ldq + head * d;
#pragma unroll 4
    for (int r = rl; r < rows; r += 4) {
        float qv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) qv[j] = (cl + 64 * j < d) ? bf(qb[(size_t)r * ldq + cl + 64 * j]) : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float t = g < G ? dtb[(size_t)r * G + g] : 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[g][j] += t * qv[j];
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 3; ++j) red[rl][g * 3 + j][cl] = acc[g][j];
    __syncthreads();
    for (int idx = tid; idx < 4 * 3 * 64; idx += 256) {
        const int gj = idx >> 6, c0 = idx & 63;
        const int g = gj / 3, j = gj - 3 * g, c = c0 + 64 * j;
        if (g < G && c < d) {
            const float v = ((red[0][gj][c0] + red[1][gj][c0]) + red[2][gj][c0]) + red[3][gj][c0];
            part[(((size_t)bh * gridDim.x + blockIdx.x) * G + g) * d + c] = v;
        }
    }
}

// d/8 lanes per row (one 16-byte load each), RL = 256 / (d/8) rows in flight, fixed-order LDS reduction over the row lanes
__global__ __launch_bounds__(256) void gq_oct(const float* __restrict__ dt, const uint16_t* __restrict__ q, long ldq,
                                              float* __restrict__ part, int B, int H, int N, int d, int G) {
    extern __shared__ float red[];                 // [RL][G * d]
    const int tid = threadIdx.x;
    const int octs = d >> 3, RL = 256 / octs;
    const int o = tid % octs, rl = tid / octs;
    const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
    const int n0 = blockIdx.x * CHUNK;
    const int rows = min(CHUNK, N - n0);
    const int P = G * d;
    float acc[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    if (rl < RL) {
        const float* dtb = dt + (((size_t)b * H + head) * N + n0) * G;
        const uint16_t* qb = q + ((size_t)b * N + n0) * ldq + head * d + 8 * o;
        for (int r = rl; r < rows; r += RL) {
            const uint4 raw = *(const uint4*)(qb + (size_t)r * ldq);
            float x[8];
            x[0] = __builtin_bit_cast(float, raw.x << 16); x[1] = __builtin_bit_cast(float, raw.x & 0xffff0000u);
            x[2] = __builtin_bit_cast(float, raw.y << 16); x[3] = __builtin_bit_cast(float, raw.y & 0xffff0000u);
            x[4] = __builtin_bit_cast(float, raw.z << 16); x[5] = __builtin_bit_cast(float, raw.z & 0xffff0000u);
            x[6] = __builtin_bit_cast(float, raw.w << 16); x[7] = __builtin_bit_cast(float, raw.w & 0xffff0000u);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g < G) {
                    const float t = dtb[(size_t)r * G + g];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(t, x[e], acc[g][e]);
                }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < G) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[(size_t)rl * P + g * d + 8 * o + e] = acc[g][e];
            }
    }
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        float v = 0.f;
        for (int i = 0; i < RL; ++i) v += red[(size_t)i * P + p];
        part[((size_t)bh * gridDim.x + blockIdx.x) * P + p] = v;
    }
}

// ---------------------------------------------------------------------------------------------- host
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

static double rel_err(const std::vector<float>& a, const std::vector<double>& ref) {
    double num = 0, den = 0;
    for (size_t i = 0; i < ref.size(); ++i) { num += (a[i] - ref[i]) * (a[i] - ref[i]); den += ref[i] * ref[i]; }
    return sqrt(num / (den + 1e-300));
}

template <class F>
static float time_us(F launch, int reps = 50) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
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

static int run_shape(int B, int H, int N, int M, int d, int G) {
    const long ld = (long)H * d;
    const int nchunks = (N + CHUNK - 1) / CHUNK, P = G * d, BH = B * H;
    const float scale = 1.f / sqrtf((float)d);
    std::vector<uint16_t> hq((size_t)B * N * ld), hk((size_t)B * M * ld);
    std::vector<float> hw((size_t)B * M * G, 0.f), hdt((size_t)BH * N * G);
    for (auto& v : hq) v = f2bf(frand());
    for (auto& v : hk) v = f2bf(frand());
    for (int b = 0; b < B; ++b)                       // ~16 subject tokens in group 0, a few background tokens in the others
        for (int m = 0; m < M; ++m)
            for (int g = 0; g < G; ++g)
                if ((g == 0 && m >= 5 && m < 21) || (g > 0 && (m % 19) == g)) hw[((size_t)b * M + m) * G + g] = 1.f + 0.5f * g;
    for (auto& v : hdt) v = frand();
    // fp64 references
    std::vector<double> rkw((size_t)BH * P, 0.0), rT((size_t)BH * N * G), rgq((size_t)BH * P, 0.0);
    for (int bh = 0; bh < BH; ++bh) {
        const int b = bh / H, head = bh % H;
        for (int g = 0; g < G; ++g)
            for (int c = 0; c < d; ++c) {
                double a = 0;
                for (int m = 0; m < M; ++m) a += (double)hw[((size_t)b * M + m) * G + g] * bf2f(hk[((size_t)b * M + m) * ld + head * d + c]);
                rkw[(size_t)bh * P + g * d + c] = a;
            }
        for (int n = 0; n < N; ++n)
            for (int g = 0; g < G; ++g) {
                double a = 0;
                for (int c = 0; c < d; ++c) a += (double)bf2f(hq[((size_t)b * N + n) * ld + head * d + c]) * rkw[(size_t)bh * P + g * d + c];
                rT[((size_t)bh * N + n) * G + g] = a * scale;
                const double t = hdt[((size_t)bh * N + n) * G + g];
                for (int c = 0; c < d; ++c) rgq[(size_t)bh * P + g * d + c] += t * bf2f(hq[((size_t)b * N + n) * ld + head * d + c]);
            }
    }
    uint16_t *dq, *dk;
    float *dw, *ddt, *dkw, *dT, *dpart;
    CK(hipMalloc(&dq, hq.size() * 2));
    CK(hipMalloc(&dk, hk.size() * 2));
    CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMalloc(&ddt, hdt.size() * 4));
    CK(hipMalloc(&dkw, (size_t)BH * P * 4));
    CK(hipMalloc(&dT, (size_t)BH * N * G * 4));
    CK(hipMalloc(&dpart, (size_t)BH * nchunks * P * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddt, hdt.data(), hdt.size() * 4, hipMemcpyHostToDevice));
    int bad = 0;
    std::vector<float> out;
    auto check = [&](const char* name, float* dev, size_t n, const std::vector<double>& ref, float us) {
        out.resize(n);
        CK(hipMemcpy(out.data(), dev, n * 4, hipMemcpyDeviceToHost));
        const double e = rel_err(out, ref);
        printf("  %-14s %8.2f us   rel err %.2e %s\n", name, us, e, e < 1e-5 ? "ok" : "MISMATCH");
        if (!(e < 1e-5)) ++bad;
    };
    auto check_gq = [&](const char* name, float us) {
        std::vector<float> part((size_t)BH * nchunks * P);
        CK(hipMemcpy(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost));
        out.assign((size_t)BH * P, 0.f);
        for (int bh = 0; bh < BH; ++bh)
            for (int ch = 0; ch < nchunks; ++ch)
                for (int p = 0; p < P; ++p) out[(size_t)bh * P + p] += part[((size_t)bh * nchunks + ch) * P + p];
        const double e = rel_err(out, rgq);
        printf("  %-14s %8.2f us   rel err %.2e %s\n", name, us, e, e < 1e-5 ? "ok" : "MISMATCH");
        if (!(e < 1e-5)) ++bad;
    };
    printf("B %d H %d N %d M %d d %d G %d  (q %.1f MB)\n", B, H, N, M, d, G, hq.size() * 2 / 1e6);
    float us;
    CK(hipMemset(dkw, 0, (size_t)BH * P * 4));
    us = time_us([&] { hipLaunchKernelGGL(kw_cur<false>, dim3(BH), dim3(256), 0, 0, dw, dk, ld, dkw, H, M, d, G); });
    check("kw cur", dkw, (size_t)BH * P, rkw, us);
    CK(hipMemset(dkw, 0, (size_t)BH * P * 4));
    us = time_us([&] { hipLaunchKernelGGL(kw_cur<true>, dim3(BH), dim3(256), 0, 0, dw, dk, ld, dkw, H, M, d, G); });
    check("kw skip", dkw, (size_t)BH * P, rkw, us);
    CK(hipMemset(dkw, 0, (size_t)BH * P * 4));
    us = time_us([&] { hipLaunchKernelGGL(kw_spread, dim3(BH), dim3(256), 0, 0, dw, dk, ld, dkw, H, M, d, G); });
    check("kw spread", dkw, (size_t)BH * P, rkw, us);
    const dim3 gf((N + ROWS - 1) / ROWS, BH);
    CK(hipMemset(dT, 0, (size_t)BH * N * G * 4));
    us = time_us([&] { hipLaunchKernelGGL(tokmap_fwd<false>, gf, dim3(256), 0, 0, dq, ld, dk, ld, dw, dT, G, H, N, M, d, scale); });
    check("fwd cur", dT, (size_t)BH * N * G, rT, us);
    CK(hipMemset(dT, 0, (size_t)BH * N * G * 4));
    us = time_us([&] { hipLaunchKernelGGL(tokmap_fwd<true>, gf, dim3(256), 0, 0, dq, ld, dk, ld, dw, dT, G, H, N, M, d, scale); });
    check("fwd spread", dT, (size_t)BH * N * G, rT, us);
    const dim3 gg(nchunks, BH);
    CK(hipMemset(dpart, 0, (size_t)BH * nchunks * P * 4));
    us = time_us([&] { hipLaunchKernelGGL(gq_cur, gg, dim3(256), 0, 0, ddt, dq, ld, dpart, B, H, N, d, G); });
    check_gq("gq cur", us);
    CK(hipMemset(dpart, 0, (size_t)BH * nchunks * P * 4));
    const size_t lds = (size_t)(256 / (d / 8)) * P * 4;
    us = time_us([&] { hipLaunchKernelGGL(gq_oct, gg, dim3(256), lds, 0, ddt, dq, ld, dpart, B, H, N, d, G); });
    check_gq("gq oct", us);
    CK(hipFree(dq)); CK(hipFree(dk)); CK(hipFree(dw)); CK(hipFree(ddt)); CK(hipFree(dkw)); CK(hipFree(dT)); CK(hipFree(dpart));
    return bad;
}

int main() {
    srand(7);
    int bad = 0;
    bad += run_shape(4, 8, 4096, 77, 40, 2);      // the UNet's 64 x 64 cross-attention layers at bs 4
    bad += run_shape(4, 8, 1024, 77, 80, 2);      // 32 x 32
    bad += run_shape(4, 8, 256, 77, 160, 2);      // 16 x 16
    printf(bad ? "FAILED: %d mismatches\n" : "all ok\n", bad);
    return bad ? 1 : 0;
}

// Issue-rate microbenchmark for the instruction mix of the attention softmax on gfx950 (run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/microbench/valu_rate.hip && /tmp/valu_rate).
// Each wave runs a loop of independent instructions of one kind (or a mix) from inline asm; cycles per instruction per SIMD =
// wave cycles (s_memtime, 100 MHz domain scaled by the measured clock) / instructions, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned long long* clk) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float s = 0.999f, b = 1e-3f;
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)0.5f; fb[i] = (__bf16)0.25f; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {          // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 1) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(b));
        } else if (OP == 2) {   // 4 x (fma, exp) pairs + 2 cvt_pk + 2 max3: the softmax mix per 4 scores
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3\n"
                         "v_max3_f32 %6, %6, %0, %1\n v_max3_f32 %7, %7, %2, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(b));
        } else if (OP == 3) {   // 2 MFMA 32x32x16 alone (independent accumulators)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                         : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));
        } else if (OP == 4) {   // the attention step's ratio: 2 MFMA + (8 fma, 8 exp, 4 cvt, 4 max3) -- 14 MFMA per 64 exps / 2
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                         "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         "v_cvt_pk_bf16_f32 %8, %4, %5\n v_cvt_pk_bf16_f32 %9, %6, %7\n"
                         "v_max3_f32 %10, %10, %4, %5\n v_max3_f32 %11, %11, %6, %7\n"
                         "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
                         "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         "v_cvt_pk_bf16_f32 %8, %4, %5\n v_cvt_pk_bf16_f32 %9, %6, %7\n"
                         "v_max3_f32 %10, %10, %4, %5\n v_max3_f32 %11, %11, %6, %7\n"
                         : "+v"(acc0), "+v"(acc1) , "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s), "v"(b));
        } else if (OP == 5) {   // 8 independent v_cvt_pk_bf16_f32
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2\n v_cvt_pk_bf16_f32 %4, %1, %2\n v_cvt_pk_bf16_f32 %5, %1, %2\n"
                         "v_cvt_pk_bf16_f32 %0, %6, %7\n v_cvt_pk_bf16_f32 %3, %6, %7\n v_cvt_pk_bf16_f32 %4, %6, %7\n v_cvt_pk_bf16_f32 %5, %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 6) {   // 8 independent v_max3_f32
            asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(b));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[0] + (float)fa[0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, int per_iter, int blocks, int iters, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e0, 0);
    const int L = 5;
    for (int w = 0; w < L; ++w) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= L;
    unsigned long long h[4096];
    hipMemcpy(h, clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    // wall time per instruction per SIMD: each SIMD executed waves_per_simd * iters * per_iter instructions in ms
    const double ns_per_inst = ms * 1e6 / (waves_per_simd * (double)iters * per_iter);
    printf("%-44s blocks %4d (%.0f waves/SIMD): %.3f ms, %.3f ns per instruction per SIMD = %.2f cycles at 2.4 GHz, %.2f at 1.9; "
           "wave clock ticks/inst %.3f\n", name, blocks, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 1.9,
           avg / ((double)iters * per_iter));
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 8);
    const int iters = 20000;
    for (int blocks : {256, 512}) {
        run<0>("v_exp_f32 x8", 8, blocks, iters, out, clk);
        run<1>("v_fma_f32 x8", 8, blocks, iters, out, clk);
        run<5>("v_cvt_pk_bf16_f32 x8", 8, blocks, iters, out, clk);
        run<6>("v_max3_f32 x8", 8, blocks, iters, out, clk);
        run<2>("softmax mix (4 fma 4 exp 2 cvt 2 max3)", 12, blocks, iters, out, clk);
        run<3>("v_mfma_f32_32x32x16_bf16 x2", 2, blocks, iters, out, clk);
        run<4>("2 mfma + 2 x softmax mix (26 inst)", 26, blocks, iters, out, clk);
    }
    return 0;
}

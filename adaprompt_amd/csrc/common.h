// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the SD-1.5 hot path.
// Wavefront = 64 lanes; MFMA 16x16x32 bf16 with fp32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// the public C ABI: every extern "C" definition in csrc/ is checked against its prototype at compile time
#include "../../include/adaprompt_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ADAP_OK 0
#define ADAP_ERR_SHAPE (-1)
#define ADAP_ERR_ALIGN (-2)
#define ADAP_ERR_UNSUPPORTED (-3)
#define ADAP_ERR_HIP (-4)

#define WAVE 64

// error plumbing (capi.cpp)
int adap_set_error(int code, const char* fmt, ...);
int adap_check_launch(const char* what);

#define ADAP_REQUIRE(cond, code, ...)                          \
    do {                                                       \
        if (!(cond)) return adap_set_error((code), __VA_ARGS__); \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;   // round-to-nearest-even; v_cvt_pk_bf16_f32 at -O3, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
    f[0] = __builtin_bit_cast(float, v.x << 16);
    f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, v.y << 16);
    f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
    f[4] = __builtin_bit_cast(float, v.z << 16);
    f[5] = __builtin_bit_cast(float, v.z & 0xffff0000u);
    f[6] = __builtin_bit_cast(float, v.w << 16);
    f[7] = __builtin_bit_cast(float, v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]);
    v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]);
    v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp): an IEEE division costs ~12 more VALU instructions per element, which is
// what the HBM-bound norm / activation kernels were spending their issue slots on
__device__ __forceinline__ float sigmoid_f(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }

// d silu(z) / dz
__device__ __forceinline__ float dsilu_f(float z) {
    float s = sigmoid_f(z);
    return s * (1.0f + z * (1.0f - s));
}

// XCD-aware bijective remap of a 1-D block id: blocks are dealt round-robin over the 8 XCDs, so
// give each XCD a contiguous chunk of the logical grid (guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int xcd = bid % nx, q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / nx;
}

// C-ABI plumbing shared by all kernels: last-error string, launch check, version / device query.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int adap_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int adap_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return adap_set_error(ADAP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return ADAP_OK;
}

extern "C" const char* adap_last_error(void) { return g_err; }

extern "C" int adap_abi_version(void) { return 3; }

// 0 on success; fills name (<= 255 chars + NUL), CU count and gcn arch string of the current device
extern "C" int adap_device_info(char* name, int name_cap, int* num_cus, char* arch, int arch_cap) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return adap_set_error(ADAP_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return adap_set_error(ADAP_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name && name_cap > 0) { strncpy(name, prop.name, name_cap - 1); name[name_cap - 1] = 0; }
    if (arch && arch_cap > 0) { strncpy(arch, prop.gcnArchName, arch_cap - 1); arch[arch_cap - 1] = 0; }
    if (num_cus) *num_cus = prop.multiProcessorCount;
    return ADAP_OK;
}


// ---------------------------------------------------------------------------------------------
// Diagnostic: a kernel that merely OCCUPIES part of the chip for a while -- `blocks` workgroups of `threads` threads, each
// holding ~`vgprs_hint` live registers per lane, until `usec` microseconds of the 100 MHz real-time clock have passed -- the
// shape of a resident collective kernel (RCCL's all-reduce: tens of workgroups for milliseconds).  tests/test_parallel_gpu.py
// runs the single-launch GroupNorm beside it on another stream.  Every wave reaches the exit condition (bounded by time).
// ---------------------------------------------------------------------------------------------
__global__ void adap_occupy_kernel(unsigned long long ticks, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = (float)(threadIdx.x + i);
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = r[i] * 1.0000001f + 0.5f;          // keeps the registers live
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += r[i];
    if (s == 12345.678f && sink) sink[0] = s;                                 // never true: the loop is not dead code
}

extern "C" int adap_debug_occupy(int blocks, int threads, int usec, float* sink, void* stream) {
    ADAP_REQUIRE(blocks >= 1 && blocks <= 1024 && threads >= 64 && threads <= 1024 && threads % 64 == 0 && usec >= 0 && usec <= 200000,
                 ADAP_ERR_UNSUPPORTED, "debug_occupy: %d blocks x %d threads for %d us", blocks, threads, usec);
    hipLaunchKernelGGL(adap_occupy_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, sink);
    return adap_check_launch("debug_occupy");
}

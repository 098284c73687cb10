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

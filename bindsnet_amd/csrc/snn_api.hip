// snn_api.hip -- library-level entry points: version, error strings, device probing.
#include <hip/hip_runtime.h>
#include "../../include/snnhip.h"
#include "snn_common.hpp"

static thread_local hipError_t g_last = hipSuccess;

int snn_check(hipError_t e) {
    if (e == hipSuccess) return SNN_OK;
    g_last = e;
    return SNN_ERR_LAUNCH;
}

int snn_check_launch() { return snn_check(hipGetLastError()); }

extern "C" int snn_abi_version(void) { return SNN_ABI_VERSION; }

extern "C" const char *snn_error_string(int code) {
    switch (code) {
        case SNN_OK: return "ok";
        case SNN_ERR_INVALID: return "invalid argument";
        case SNN_ERR_UNSUPPORTED: return "size or mode not supported by the gfx950 kernels";
        case SNN_ERR_LAUNCH: return "HIP runtime / launch error";
        case SNN_ERR_NOISE: return "one_spike noise stream exhausted";
        case SNN_ERR_NO_DEVICE: return "no HIP device";
        case SNN_ERR_TIMEOUT: return "in-kernel workgroup hand-off timed out";
        case SNN_ERR_RETRY: return "lean kernel form met an unsupported step (re-run with plan = 3)";
        default: return "unknown error";
    }
}

extern "C" const char *snn_last_hip_error(void) { return hipGetErrorString(g_last); }

extern "C" int snn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SNN_ERR_NO_DEVICE;
    return n;
}

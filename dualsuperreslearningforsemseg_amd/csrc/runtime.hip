// Error reporting, device binding, version.
#include "common.h"
#include <string.h>

namespace dsrl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DSRL_E_LAUNCH;
    }
    return DSRL_OK;
}

int bind_stream_device(hipStream_t s) {
    if (s == nullptr) return DSRL_OK;      // legacy default stream of the caller's current device
    hipDevice_t dev;
    if (hipStreamGetDevice(s, &dev) != hipSuccess) {
        (void)hipGetLastError();
        return DSRL_OK;
    }
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != (int)dev) {
        if (hipSetDevice((int)dev) != hipSuccess) {
            set_error("hipSetDevice(%d) failed", (int)dev);
            return DSRL_E_LAUNCH;
        }
    }
    return DSRL_OK;
}

}  // namespace dsrl

extern "C" int dsrl_version(void) { return DSRL_ABI_VERSION; }
extern "C" const char* dsrl_last_error(void) { return dsrl::g_err; }

extern "C" int dsrl_device_check(int* cu_count) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { dsrl::set_error("no HIP device"); (void)hipGetLastError(); return DSRL_E_LAUNCH; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { dsrl::set_error("hipGetDeviceProperties failed"); return DSRL_E_LAUNCH; }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        dsrl::set_error("device %d is %s; libdsrl_hip.so is built for gfx950 only", dev, p.gcnArchName);
        return DSRL_E_UNSUPPORTED;
    }
    return DSRL_OK;
}

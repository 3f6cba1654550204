// Error reporting and build identification for libpunet_hip.so.
#include "common.h"
#include "punet_hip.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void punet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* punet_last_error(void) { return g_err; }

extern "C" const char* punet_build_info(void) { return "libpunet_hip gfx950 fp32-mfma " __DATE__ " " __TIME__; }

// 1 if a HIP device is visible and is a gfx950 part, 0 otherwise (never throws, never allocates on the device)
extern "C" int punet_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        punet_set_error("no HIP device visible");
        return 0;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    if (__builtin_strstr(prop.gcnArchName, "gfx950") == nullptr) {
        punet_set_error("device 0 is %s, this library is built for gfx950 only", prop.gcnArchName);
        return 0;
    }
    return 1;
}

int punet_num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    return n;
}

extern "C" int punet_compute_units(void) { return punet_num_cus(); }

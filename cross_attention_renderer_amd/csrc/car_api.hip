// car_api.hip — version, error string and device queries of the C ABI (include/car_hip.h).
#include "car_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void car_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int car_version(void) { return CAR_VERSION; }
extern "C" const char* car_last_error(void) { return g_err; }

extern "C" int car_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { car_set_error("no HIP device"); return CAR_E_NODEVICE; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { car_set_error("hipGetDeviceProperties failed"); return CAR_E_NODEVICE; }
    return p.multiProcessorCount;
}

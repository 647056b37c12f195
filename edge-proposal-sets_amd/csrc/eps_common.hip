// Host-side plumbing of libeps_hip.so: thread-local error text, device query.
#include "eps_common.h"

#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void eps_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *eps_last_error(void) { return g_err; }

extern "C" int eps_version(void) { return EPS_ABI_VERSION; }

int eps_num_cus()
{
    static thread_local int cached_dev = -1;
    static thread_local int cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached_cus = n;
        cached_dev = dev;
    }
    return cached_cus;
}

extern "C" int eps_device_info(int *n_cu, char *name, int name_len)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        eps_set_error("eps_device_info: no HIP device");
        return EPS_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        eps_set_error("eps_device_info: hipGetDeviceProperties failed");
        return EPS_ENODEV;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return EPS_OK;
}

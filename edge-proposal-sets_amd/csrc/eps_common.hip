// Host-side plumbing of libeps_hip.so: thread-local error text, device query.
#include "eps_common.h"

#include <stdarg.h>
#include <string.h>

#include <atomic>

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

// Work counters for the kernels that hand out work dynamically (one device word per launch): a small pool of device
// words inside the code object -- nothing is allocated at run time.  Each launch takes the next slot and zeroes it on
// its own stream before the kernel; 64 slots bound the number of such launches that may be in flight at once.
#define EPS_COUNTER_SLOTS 64
__device__ unsigned int g_work_counter[EPS_COUNTER_SLOTS];
static std::atomic<unsigned int> g_counter_turn{0};

int eps_take_counter(unsigned int **counter, hipStream_t stream, const char *who)
{
    if (hipGetSymbolAddress((void **)counter, HIP_SYMBOL(g_work_counter)) != hipSuccess) {
        eps_set_error("%s: cannot resolve the work counter", who);
        return EPS_ELAUNCH;
    }
    *counter += g_counter_turn.fetch_add(1) % EPS_COUNTER_SLOTS;
    if (hipMemsetAsync(*counter, 0, sizeof(unsigned int), stream) != hipSuccess) {
        eps_set_error("%s: cannot reset the work counter", who);
        return EPS_ELAUNCH;
    }
    return EPS_OK;
}

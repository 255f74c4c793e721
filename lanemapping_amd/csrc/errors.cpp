// Thread-local last-error string for the C-ABI (lm_last_error).
#include "common.h"
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = "";

void lm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

LM_API const char* lm_last_error(void) { return g_err; }
LM_API int lm_abi_version(void) { return 1; }

LM_API int lm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The dynamic-LDS limit of a kernel is per-DEVICE state of the HIP runtime: remember the largest size set per (device, kernel) so that a
// process that switches devices (or launches from several threads) never starts a > 64 KB kernel without it.
int lm_ensure_dynamic_lds(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> set;
    int dev = 0;
    LM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = set[std::make_pair(dev, fn)];
    if (bytes > cur) {
        LM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        cur = bytes;
    }
    return LM_OK;
}

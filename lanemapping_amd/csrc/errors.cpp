// Thread-local last-error string for the C-ABI (lm_last_error).
#include "common.h"
#include <cstring>

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

// Shared helpers for the lanemap_hip C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#define LM_API extern "C" __attribute__((visibility("default")))

enum {
    LM_OK = 0,
    LM_ERR_ARG = 1,      // unsupported shape / null pointer
    LM_ERR_HIP = 2,      // HIP runtime error (see lm_last_error)
    LM_ERR_NO_DEVICE = 3
};

void lm_set_error(const char* fmt, ...);

#define LM_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            lm_set_error(__VA_ARGS__);   \
            return LM_ERR_ARG;           \
        }                                \
    } while (0)

#define LM_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            lm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LM_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

#define LM_LAUNCH_CHECK() LM_HIP(hipGetLastError())

static inline int lm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// activation codes shared by the GEMM/conv epilogues
enum { LM_ACT_NONE = 0, LM_ACT_RELU = 1, LM_ACT_GELU = 2 };

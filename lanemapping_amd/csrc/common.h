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
int lm_ensure_dynamic_lds(const void* kernel, size_t bytes);   // per device and kernel, thread-safe (errors.cpp)

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

#ifdef __HIPCC__
// GroupNorm(C,C) + ReLU + bilinear (align_corners=True) building blocks with a FIXED operation order: `#pragma clang fp contract(off)`
// keeps the compiler from fusing their multiplies and adds differently in different kernels (HIP's __fmul_rn / __fadd_rn are plain
// operators and do not prevent that), explicit fmaf where a fused multiply-add is wanted.  The same bits in every kernel that uses
// them (norm_resize.hip, the fused Winograd input transform, the bilinear residual of the convolution epilogue).
// Source index follows ATen: scale = (in-1)/(out-1) in fp32, src = scale*dst, i0 = floor(src), i1 = min(i0+1, in-1), w1 = src - i0.
// (lm_bilin_axis_scaled: the same with the scale computed once by the caller - on the host the IEEE single-precision quotient is the
// same float, hipcc divides correctly rounded by default)
__device__ __forceinline__ void lm_bilin_axis_scaled(int o, int in, float scale, int& i0, int& i1, float& w0, float& w1) {
#pragma clang fp contract(off)
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    w0 = 1.f - w1;
}
__device__ __forceinline__ void lm_bilin_axis(int o, int in, int out, int& i0, int& i1, float& w0, float& w1) {
#pragma clang fp contract(off)
    const float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    lm_bilin_axis_scaled(o, in, scale, i0, i1, w0, w1);
}
#endif
// Division of n < 2^31 by an invariant divisor d >= 1 (Granlund & Montgomery, round-up variant): q = (mulhi(n, mul) + n) >> sh with
// sh = ceil(log2 d), mul = floor(2^32 (2^sh - d) / d) + 1; the sum stays below 2^32 for n < 2^31.  3 VALU instructions instead of the
// ~25 of a 32-bit (or ~60 of a 64-bit) division by a run-time value.
struct LmFastDiv {
    unsigned d, mul, sh;
};
static inline LmFastDiv lm_fastdiv_make(unsigned d) {
    LmFastDiv f;
    f.d = d;
    f.sh = 0;
    while ((1ull << f.sh) < d) ++f.sh;
    f.mul = (unsigned)((((1ull << f.sh) - d) << 32) / d + 1);
    return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ unsigned lm_fastdiv(unsigned n, const LmFastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.sh; }
__device__ __forceinline__ void lm_gn_affine(float mean, float rstd, float gamma, float beta, float& a, float& g) {
#pragma clang fp contract(off)
    a = rstd * gamma;                        // gn(v) = v * a + g
    g = __builtin_fmaf(-mean, a, beta);
}
__device__ __forceinline__ float lm_gn_relu(float v, float a, float g) { return fmaxf(__builtin_fmaf(v, a, g), 0.f); }
__device__ __forceinline__ float lm_bilerp(float v00, float v01, float v10, float v11, float wy0, float wy1, float wx0, float wx1) {
#pragma clang fp contract(off)
    const float top = wx0 * v00 + wx1 * v01;
    const float bot = wx0 * v10 + wx1 * v11;
    return wy0 * top + wy1 * bot;
}
#endif

/* C-ABI smoke test: a plain C program (no Python, no torch) drives liblanemap_hip.so through include/lanemap_hip.h.
 *   1. LAS -> BEV rasteriser on a seeded cloud, compared bit for bit with the C oracle (oracle/raster_ref.c, linked in);
 *   2. a 3x3 convolution on the matrix cores (lm_conv2d_nhwc_mfma_f32) against a scalar C loop.
 * Built and run by tests/test_gpu_parity.py::test_c_abi_from_plain_c (gcc tests/c_abi/smoke.c oracle/raster_ref.c -llanemap_hip -lamdhip64). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lanemap_hip.h"

typedef struct {
    float quat[4], trans[3], bev_img_offset[2], img_reso[2], local_min_ele, ele_reso, inten_lo, inten_hi;
} RasterParams;
void raster_ref(const float* pts, long n, const RasterParams* P, uint32_t* acc, uint8_t* out_u8, int H, int W);

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_LM(x) do { int e_ = (x); if (e_ != 0) { printf("lanemap error %d: %s\n", e_, lm_last_error()); return 3; } } while (0)

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static double urand(void) {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (double)(rng >> 11) / 9007199254740992.0;
}

int main(void) {
    if (lm_device_count() < 1) { printf("no device\n"); return 1; }
    /* ---- 1. rasteriser ---- */
    const int H = 256, W = 256;
    const long N = 200000;
    float* pts = (float*)malloc(N * 4 * sizeof(float));
    for (long i = 0; i < N; ++i) {
        pts[4 * i + 0] = (float)(urand() * H * 0.05);
        pts[4 * i + 1] = (float)(urand() * W * 0.05);
        pts[4 * i + 2] = (float)(urand() * 2.0 - 0.5);
        pts[4 * i + 3] = (float)floor(urand() * 40000.0);
    }
    LmRasterParams P = {{0.98f, 0.01f, -0.02f, 0.05f}, {0.1f, -0.2f, 0.05f}, {0.f, 0.f}, {0.05f, 0.05f}, -0.5f, 0.02f, 800.f, 33000.f};
    RasterParams R;
    memcpy(&R, &P, sizeof(R));
    uint32_t* acc = (uint32_t*)malloc((size_t)H * W * 4);
    uint8_t* want = (uint8_t*)malloc((size_t)H * W * 3);
    raster_ref(pts, N, &R, acc, want, H, W);
    float* d_pts; uint8_t* d_u8; float* d_chw; void* d_ws;
    const long offs[2] = {0, N};
    const long ws_bytes = lm_bev_raster_workspace_bytes(1, N, H, W);
    CHECK_HIP(hipMalloc((void**)&d_pts, N * 16));
    CHECK_HIP(hipMalloc((void**)&d_u8, (size_t)H * W * 3));
    CHECK_HIP(hipMalloc((void**)&d_chw, (size_t)H * W * 3 * 4));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_pts, pts, N * 16, hipMemcpyHostToDevice));
    CHECK_LM(lm_bev_raster_batch(NULL, d_pts, offs, &P, 1, d_ws, ws_bytes, d_chw, d_u8, H, W));
    uint8_t* got = (uint8_t*)malloc((size_t)H * W * 3);
    CHECK_HIP(hipMemcpy(got, d_u8, (size_t)H * W * 3, hipMemcpyDeviceToHost));
    long bad = 0, occupied = 0;
    for (long i = 0; i < (long)H * W * 3; ++i) { bad += got[i] != want[i]; occupied += want[i] != 0; }
    printf("raster: %ld of %d bytes differ, %ld non-zero\n", bad, H * W * 3, occupied);
    if (bad || occupied < 1000) return 4;
    /* an argument error must come back as a code + message, not a crash */
    if (lm_bev_raster_batch(NULL, d_pts, offs, &P, 1, d_ws, 16, d_chw, d_u8, H, W) == 0) { printf("missing error\n"); return 5; }
    printf("expected error: %s\n", lm_last_error());
    /* ---- 2. MFMA convolution: B=1, 40x40x32 -> 64 channels, 3x3, pad 1, ReLU ---- */
    const int Hc = 40, Wc = 40, Ci = 32, Co = 64, CoP = 128;
    float* x = (float*)malloc((size_t)Hc * Wc * Ci * 4);
    float* w = (float*)calloc((size_t)9 * CoP * Ci, 4);            /* [tap][CoutP][Cin] */
    float* shift = (float*)malloc(Co * 4);
    for (int i = 0; i < Hc * Wc * Ci; ++i) x[i] = (float)(urand() - 0.5);
    for (int t = 0; t < 9; ++t) for (int o = 0; o < Co; ++o) for (int c = 0; c < Ci; ++c) w[((size_t)t * CoP + o) * Ci + c] = (float)((urand() - 0.5) * 0.1);
    for (int o = 0; o < Co; ++o) shift[o] = (float)(urand() - 0.5);
    float *d_x, *d_w, *d_s, *d_y;
    CHECK_HIP(hipMalloc((void**)&d_x, (size_t)Hc * Wc * Ci * 4));
    CHECK_HIP(hipMalloc((void**)&d_w, (size_t)9 * CoP * Ci * 4));
    CHECK_HIP(hipMalloc((void**)&d_s, Co * 4));
    CHECK_HIP(hipMalloc((void**)&d_y, (size_t)Hc * Wc * Co * 4));
    CHECK_HIP(hipMemcpy(d_x, x, (size_t)Hc * Wc * Ci * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_w, w, (size_t)9 * CoP * Ci * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_s, shift, Co * 4, hipMemcpyHostToDevice));
    CHECK_LM(lm_conv2d_nhwc_mfma_f32(NULL, d_x, Ci, d_w, CoP, NULL, d_s, NULL, 0, 0, d_y, Co, 1, Hc, Wc, Ci, Co, 3, 3, 1, 1, 1, 1, 1 /* ReLU */));
    float* y = (float*)malloc((size_t)Hc * Wc * Co * 4);
    CHECK_HIP(hipMemcpy(y, d_y, (size_t)Hc * Wc * Co * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int oy = 0; oy < Hc; ++oy) for (int ox = 0; ox < Wc; ++ox) for (int o = 0; o < Co; ++o) {
        double a = shift[o];
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (iy < 0 || iy >= Hc || ix < 0 || ix >= Wc) continue;
            for (int c = 0; c < Ci; ++c) a += (double)x[((size_t)iy * Wc + ix) * Ci + c] * w[((size_t)t * CoP + o) * Ci + c];
        }
        if (a < 0) a = 0;
        const double d = fabs(a - y[((size_t)oy * Wc + ox) * Co + o]);
        if (d > worst) worst = d;
    }
    printf("conv: max |diff| vs the scalar loop = %.3e\n", worst);
    if (worst > 1e-5) return 6;
    printf("C-ABI smoke OK (abi version %d)\n", lm_abi_version());
    return 0;
}

// Probe (not product): does the sustained fp32 MFMA rate depend on the operand DATA (power management)?  Pure MFMA stream, 2 chains per
// wave, 2 waves per SIMD, operands held in registers; operand values: small integers / full-mantissa pseudo-random floats / zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float rnd(unsigned x) {   // full-mantissa value in [-2, 2)
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return __uint_as_float(0x3f800000u | (x & 0x007fffffu)) * ((x >> 31) ? 1.f : -1.f);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = MODE == 0 ? 0.f : MODE == 1 ? (float)((t + i) & 7) : rnd(t * 16 + i) * 0.01f;
        b[i] = MODE == 0 ? 0.f : MODE == 1 ? (float)((t + 3 * i) & 3) : rnd(t * 16 + 8 + i) * 0.01f;
    }
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int s = 0; s < iters; ++s) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 7], b[(k + 1) & 7], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 3) & 7], b[k & 7], acc[1], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[c][r];
    out[t] = sum;
}

template <int MODE>
void run(float* d, const char* what) {
    const int blocks = 512 * 8, iters = 512;     // 4096 x 4 waves x 512 x 32 MFMAs = 1.1 TFLOP per launch (~7 ms at peak)
    hipEvent_t ev[9];
    for (auto& e : ev) hipEventCreate(&e);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, 4);
    hipDeviceSynchronize();
    for (int i = 0; i < 8; ++i) {
        hipEventRecord(ev[i], 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    }
    hipEventRecord(ev[8], 0);
    hipEventSynchronize(ev[8]);
    printf("%-44s", what);
    for (int i = 0; i < 8; ++i) {
        float ms;
        hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
        printf(" %6.1f", (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
    }
    printf("  TFLOP/s per consecutive 1.1-TFLOP launch\n");
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    run<0>(d, "operands all zero");
    run<1>(d, "operands small integers (0..7)");
    run<2>(d, "operands full-mantissa pseudo-random");
    run<0>(d, "operands all zero (again)");
    run<2>(d, "full-mantissa pseudo-random (again)");
    return 0;
}

// Probe (not product): how many independent accumulator chains per wave / waves per SIMD does v_mfma_f32_32x32x2_f32 need
// to keep the matrix pipe busy?  Pure register MFMA loops, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
void run(float* d, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 * iters * 8 * NCH * 4096.0;
    printf("chains/wave %d, waves/SIMD %d: %.1f TFLOP/s\n", NCH, wgs_per_cu, flop / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; ++w) {
        run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<8>(d, w);
    }
    return 0;
}

// Probe (not product): fp32 MFMA rate with the accumulators in VGPRs vs AGPRs, at 1..4 waves per SIMD (LDS-limited occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int AG>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0) {
    extern __shared__ float lds[];
    if (a0 == 12345.f) lds[threadIdx.x] = a0;
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    float a = a0 + (t & 7), b = a0 * 0.5f + (t & 3);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    for (int s = 0; s < iters; ++s) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (AG) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(b), "v"(a));
            } else {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(b), "v"(a));
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
    out[t] = sum;
}

template <int AG>
void run(float* d, int wgs_per_cu, const char* what) {
    const int lds = 160 * 1024 / wgs_per_cu - 1024;
    hipFuncSetAttribute((const void*)probe<AG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 5184, iters = 128;       // 4096 MFMAs per wave, like one Winograd GEMM workgroup
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<AG>, dim3(blocks), dim3(256), lds, 0, d, 4, 1.f);
    hipEventRecord(a, 0);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(probe<AG>, dim3(blocks), dim3(256), lds, 0, d, iters, 1.f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-30s %d workgroups/CU (%d waves/SIMD): %.3f ms per launch, %.1f TFLOP/s\n", what, wgs_per_cu, wgs_per_cu, ms / 4,
           4.0 * blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 5184 * 256 * 4);
    for (int w = 1; w <= 4; ++w) {
        run<0>(d, w, "accumulators in VGPRs");
        run<1>(d, w, "accumulators in AGPRs");
    }
    return 0;
}

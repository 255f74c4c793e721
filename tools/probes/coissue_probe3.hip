// Probe (not product), part 3 (round 6): the same question as coissue_probe2 for the fp16 matrix instruction of the split second line.
// TWO waves per SIMD stream v_mfma_f32_32x32x8_f16 (8 steps of 8 per iteration); both carry NV v_fma_f32 per step behind the step's first
// MFMA.  Compared with ONE wave per SIMD carrying the same work (256 threads).  Does VALU work hide under fp16 MFMAs - in the wave, across waves?
// Prints wall cycles per MFMA of the SIMD (floor: 32 = 8 passes of v_mfma_f32_32x32x8_f16; f32 32x32x2 for reference: floor 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NV, int F16>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float a0, float b0, unsigned long long* cyc) {
    const int tid = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + tid, b = b0;
    f16x4 ah = {(_Float16)a, (_Float16)b, (_Float16)1.f, (_Float16)2.f}, bh = {(_Float16)b, (_Float16)a, (_Float16)3.f, (_Float16)1.f};
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    f16x8 a8 = {ah[0], ah[1], ah[2], ah[3], ah[0], ah[1], ah[2], ah[3]}, b8 = {bh[0], bh[1], bh[2], bh[3], bh[0], bh[1], bh[2], bh[3]};
    float v[8] = {a, b, a + 1.f, b + 1.f, a, b, a, b};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (F16 == 2) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[0], 0, 0, 0);
            else if (F16) acc[0] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bh, acc[0], 0, 0, 0);
            else acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 1; t < 8; ++t) {           // (eight independent accumulators: throughput, not the dependent-chain latency)
                if (F16 == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[t], 0, 0, 0);
                else if (F16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bh, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    const long long t2 = clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(t2 - t0);
}

static float* d_out;
static unsigned long long* d_cyc;

template <int NV, int F16>
void run(int threads) {
    const int blocks = 256, iters = 200;
    hipLaunchKernelGGL((probe<NV, F16>), dim3(blocks), dim3(threads), 0, 0, d_out, 10, 1.f, 2.f, d_cyc);
    hipLaunchKernelGGL((probe<NV, F16>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.f, 2.f, d_cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    static unsigned long long h[4096];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const int waves_per_simd = threads / 256;
    const double mfmas_per_simd = (double)iters * 64 * waves_per_simd;
    printf("%s, %d wave(s) per SIMD, %2d v_fma per step of 8 MFMAs in every wave: %6.1f cycles per MFMA of the SIMD\n",
           F16 == 2 ? "v_mfma_f32_32x32x16_f16" : F16 ? "v_mfma_f32_32x32x8_f16 " : "v_mfma_f32_32x32x2_f32 ", waves_per_simd, NV, s / blocks / mfmas_per_simd);
}

int main() {
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_cyc, 4096 * sizeof(unsigned long long));
    run<0, 2>(256); run<8, 2>(256); run<16, 2>(256); run<32, 2>(256);
    run<0, 2>(512); run<8, 2>(512); run<16, 2>(512); run<32, 2>(512);
    run<0, 1>(256); run<8, 1>(256); run<16, 1>(256); run<32, 1>(256);
    run<0, 1>(512); run<8, 1>(512); run<16, 1>(512); run<32, 1>(512);
    run<0, 0>(256); run<8, 0>(256); run<16, 0>(256);
    run<0, 0>(512); run<8, 0>(512); run<16, 0>(512);
    return 0;
}

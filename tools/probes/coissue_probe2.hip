// Probe (not product), part 2: TWO waves per SIMD that BOTH stream f32 MFMAs (8 steps of 8 per iteration); one of them also carries
// NV v_fma_f32 per step behind the step's first MFMA (the in-wave transform pieces of the octo Winograd kernel).  Who pays for those
// VALU instructions - does the partner's MFMA fill the gap?  Variants: which wave carries the VALU work (the older = lower wave id,
// or the younger), and s_setprio levels of the two waves (constant, or lowered only around the VALU section).
// Prints cycles per (MFMA of the pair) = wall cycles / (2 x MFMAs per wave): floor 32 (= 64 per MFMA, two waves sharing the pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// WHO: 0 = waves 0..3 carry the VALU work, 1 = waves 4..7.  PV / PM: s_setprio of the VALU-carrying / the pure-MFMA wave.
// DYN: 1 = the VALU-carrying wave drops to priority 0 for its VALU section and returns to PV after it.
template <int NV, int WHO, int PV, int PM, int DYN>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float a0, float b0, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool carrier = (wave >> 2) == WHO;
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + tid, b = b0;
    float v[8] = {a, b, a + 1.f, b + 1.f, a, b, a, b};
    lds[tid] = 0.f;
    __syncthreads();
    if (carrier) __builtin_amdgcn_s_setprio(PV); else __builtin_amdgcn_s_setprio(PM);
    const long long t0 = clock64();
    if (carrier) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (DYN) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
                if (DYN) __builtin_amdgcn_s_setprio(PV);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 1; t < 8; ++t) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    const long long t2 = clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 255) == 0) cyc[(tid >> 8) * 4096 + blockIdx.x] = (unsigned long long)(t1 - t0);
    if (tid == 0) cyc[8192 + blockIdx.x] = (unsigned long long)(t2 - t0);
}

static float* d_out;
static unsigned long long* d_cyc;

template <int NV, int WHO, int PV, int PM, int DYN>
void run() {
    const int blocks = 256, iters = 200;
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute((const void*)probe<NV, WHO, PV, PM, DYN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((probe<NV, WHO, PV, PM, DYN>), dim3(blocks), dim3(512), lds, 0, d_out, 10, 1.f, 2.f, d_cyc);
    hipLaunchKernelGGL((probe<NV, WHO, PV, PM, DYN>), dim3(blocks), dim3(512), lds, 0, d_out, iters, 1.f, 2.f, d_cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    static unsigned long long h[3 * 4096];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < blocks; ++i) { s0 += (double)h[i]; s1 += (double)h[4096 + i]; s2 += (double)h[8192 + i]; }
    const double n = (double)iters * 64;
    printf("%2d v_fma per step in the %s wave, prio VALU-wave %d%s / MFMA-wave %d: pair %.1f cycles per 2 MFMAs (floor 128) | wave 0 done after %.0f %%, wave 4 after %.0f %% of the pair\n",
           NV, WHO ? "YOUNGER" : "OLDER", PV, DYN ? " (0 around the VALU section)" : "", PM, s2 / blocks / n, 100 * s0 / s2, 100 * s1 / s2);
}

int main() {
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_cyc, 3 * 4096 * sizeof(unsigned long long));
    hipMemset(d_cyc, 0, 3 * 4096 * sizeof(unsigned long long));
    run<0, 0, 0, 0, 0>();
    run<8, 0, 0, 0, 0>();  run<8, 1, 0, 0, 0>();
    run<8, 0, 0, 1, 0>();  run<8, 1, 0, 1, 0>();
    run<8, 0, 1, 0, 0>();  run<8, 1, 1, 0, 0>();
    run<8, 0, 1, 1, 1>();  run<8, 1, 1, 1, 1>();
    run<8, 0, 0, 3, 0>();  run<8, 1, 0, 3, 0>();
    run<8, 0, 3, 3, 1>();  run<8, 1, 3, 3, 1>();
    run<3, 0, 0, 0, 0>();  run<3, 1, 0, 0, 0>();
    run<3, 0, 0, 1, 0>();  run<3, 1, 0, 1, 0>();
    run<16, 0, 0, 0, 0>(); run<16, 1, 0, 0, 0>();
    run<16, 0, 0, 1, 0>(); run<16, 1, 0, 1, 0>();
    return 0;
}

// Probe (not product): issue rate of v_mfma_f32_32x32x16_bf16 in the patterns of the split-precision Winograd kernel (one wave per SIMD,
// 12 MFMAs per pair step): NCH accumulator chains interleaved, NLDS ds_read_b128 behind the first two MFMAs, NVM 16-byte global loads
// (L2-resident, ring of 8 sets waited three pair steps later), BAR = s_barrier every 8 pair steps.  Prints cycles per MFMA (floor 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BF(x) __builtin_bit_cast(bf16x8, x)

template <int NCH, int NLDS, int NVM, int BAR, int NV = 0, int KIND = 0>
__global__ __launch_bounds__(256) void probe(float* out, const float* src, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int k = tid; k < 8192; k += 256) lds[k] = 1.f;
    __syncthreads();
    f32x4 a[6], b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { a[k] = f32x4{1.f, 2.f, 3.f, 4.f}; b[k] = f32x4{1.f, 2.f, 3.f, 4.f}; }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 vv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) vv[k] = f32x2{1.f + tid, 2.f + k};
    const float* gp = src + tid * 4 + (blockIdx.x & 7) * 8192;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            if (NVM) {
#pragma unroll
                for (int k = 0; k < NVM; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[k]) : "v"(gp + k * 1024 + ps * 64) : "memory");
            }
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                acc[m % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[m % 6]), BF(b[(m + 1) % 6]), acc[m % NCH], 0, 0, 0);
                if (NV) {       // NV VALU instructions behind every MFMA: KIND 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_and_b32, 3 = v_perm_b32
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vv[k & 7][0]) : "v"(vv[8][0]), "v"(vv[8][1]));
                        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vv[k & 7]) : "v"(vv[8]), "v"(vv[9]));
                        if (KIND == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(vv[k & 7][0]) : "v"(vv[8][0]));
                        if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(vv[k & 7][0]) : "v"(vv[8][0]), "v"(vv[9][1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (m == 1) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NLDS; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[k]) : "v"(tid * 16), "n"(k * 4096) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (NVM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += vv[k][0] + vv[k][1];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = (unsigned long long)(t1 - t0);
}

static float *d_out, *d_src;
static unsigned long long* d_cyc;

template <int NCH, int NLDS, int NVM, int BAR, int NV = 0, int KIND = 0>
void run() {
    const int blocks = 256, iters = 200;
    const size_t lds = 136 * 1024;
    hipFuncSetAttribute((const void*)probe<NCH, NLDS, NVM, BAR, NV, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((probe<NCH, NLDS, NVM, BAR, NV, KIND>), dim3(blocks), dim3(256), lds, 0, d_out, d_src, 10, d_cyc);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NCH, NLDS, NVM, BAR, NV, KIND>), dim3(blocks), dim3(256), lds, 0, d_out, d_src, iters, d_cyc);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[1024];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += (double)h[i];
    const double n = (double)iters * 96;
    if (NV) printf("%d VALU of kind %d behind every MFMA | ", NV, KIND);
    printf("%d chains, %d ds_read_b128, %d global loads per pair step, barrier %d: %.1f counter ticks per MFMA, %.2f ns per MFMA (wall)\n", NCH, NLDS, NVM, BAR,
           s / 1024 / n, ms * 1e6 / n);
}

int main() {
    hipMalloc(&d_out, 256 * 256 * 4);
    hipMalloc(&d_src, 1 << 20);
    hipMemset(d_src, 0, 1 << 20);
    hipMalloc(&d_cyc, 1024 * sizeof(unsigned long long));
    run<1, 0, 0, 0>(); run<2, 0, 0, 0>(); run<3, 0, 0, 0>(); run<4, 0, 0, 0>();
    run<2, 6, 0, 0>(); run<4, 6, 0, 0>();
    run<2, 6, 0, 1>();
    run<2, 0, 0, 0, 1, 0>(); run<2, 0, 0, 0, 2, 0>(); run<2, 0, 0, 0, 4, 0>(); run<2, 0, 0, 0, 6, 0>(); run<2, 0, 0, 0, 8, 0>();
    run<2, 0, 0, 0, 2, 1>(); run<2, 0, 0, 0, 4, 1>(); run<2, 0, 0, 0, 6, 1>();
    run<2, 0, 0, 0, 4, 2>(); run<2, 0, 0, 0, 6, 2>(); run<2, 0, 0, 0, 4, 3>(); run<2, 0, 0, 0, 6, 3>();
    run<2, 6, 6, 1, 4, 0>(); run<2, 6, 6, 1, 4, 1>();
    run<2, 0, 6, 0>(); run<2, 6, 6, 0>(); run<2, 6, 6, 1>(); run<4, 6, 6, 1>();
    return 0;
}

// Probe (not product): which ingredient of the Winograd GEMM's inner loop costs MFMA issue slots?  2 accumulator tiles per
// wave, 4 waves per workgroup, 2 workgroups per CU (LDS sized accordingly), features switched on one by one.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// F bit 0: distinct operand registers per MFMA;  bit 1: operands from LDS (3 ds_read_b128 per 8 MFMAs);
// bit 2: __syncthreads() per 32 MFMAs;  bit 3: every 256 MFMAs fold the accumulators into 4 output sets and clear them;
// bit 4: 6 global_load_lds_dwordx4 per thread and slab into the other LDS buffer (streaming 24 KB per workgroup and slab)
template <int F>
__global__ __launch_bounds__(256) void probe(float* out, int slabs, float a0, const float* __restrict__ src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = a0 + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[2], o[4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    f32x4 ca = {a0, a0 + 1, a0 + 2, a0 + 3}, cb0 = {1.f, 2.f, 3.f, 4.f}, cb1 = {2.f, 3.f, 4.f, 5.f};
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int wave = threadIdx.x >> 6;
    const float* gp = src + ((long)blockIdx.x * 6144 * 4 + threadIdx.x * 4);
    for (int s = 0; s < slabs; ++s) {
        if (F & 16) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(gp + ((s & 3) * 6144 + i * 1024)), (lptr_t*)(lds + ((s + 1) & 1) * 6144 + i * 1024 + wave * 256), 16, 0, 0);
        }
        const float* base = lds + (s & 1) * 6144 + (lane & 31) * 32;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af = ca, b0 = cb0, b1 = cb1;
            if (F & 2) {
                const int fo = ((2 * kk + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4;
                af = *reinterpret_cast<const f32x4*>(base + fo);
                b0 = *reinterpret_cast<const f32x4*>(base + 2048 + fo);
                b1 = *reinterpret_cast<const f32x4*>(base + 4096 + fo);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int u = (F & 1) ? t : 0;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u], b0[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u], b1[u], acc[1], 0, 0, 0);
            }
        }
        if (F & 4) __syncthreads();
        if ((F & 8) && (s & 7) == 7) {
            const float c0 = (s & 8) ? 1.f : -1.f, c1 = (s & 16) ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[c][r] = fmaf(acc[c][r], c0, o[c][r]);
                    o[2 + c][r] = fmaf(acc[c][r], c1, o[2 + c][r]);
                    acc[c][r] = 0.f;
                }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[c][r] + o[c][r] + o[2 + c][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int F>
void run(float* d, const char* what) {
    const int blocks = 512 * 4, slabs = 1024;
    hipFuncSetAttribute((const void*)probe<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    static float* src = nullptr;
    if (!src) { hipMalloc(&src, (size_t)2048 * 6144 * 4 * 4); hipMemset(src, 0, (size_t)2048 * 6144 * 4 * 4); }
    hipLaunchKernelGGL(probe<F>, dim3(blocks), dim3(256), 72 * 1024, 0, d, 8, 1.f, src);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(probe<F>, dim3(blocks), dim3(256), 72 * 1024, 0, d, slabs, 1.f, src);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-62s %.1f TFLOP/s\n", what, (double)blocks * 4 * slabs * 32 * 4096.0 / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 2048 * 256 * 4);
    run<0>(d, "2 chains, same operand registers");
    run<1>(d, "+ distinct operand registers");
    run<3>(d, "+ operands from LDS (3 ds_read_b128 / 8 MFMA)");
    run<7>(d, "+ barrier per 32 MFMAs");
    run<15>(d, "+ fold into 4 output sets every 256 MFMAs");
    run<11>(d, "LDS operands + fold, no barrier");
    run<5>(d, "distinct registers + barrier, no LDS");
    run<31>(d, "everything + 6 global_load_lds per thread and slab");
    run<23>(d, "everything except the fold + global_load_lds");
    return 0;
}

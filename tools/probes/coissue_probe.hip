// Probe (not product): what hides under v_mfma_f32_32x32x2_f32 on gfx950?
//  (1) IN-WAVE, one wave per SIMD: a step = 8 MFMAs on one accumulator (the implicit Winograd kernel's step); M filler groups per step
//      are issued behind the step's first MFMA (placement 0) or one behind each of the first M MFMAs (placement 1).
//      Kinds: 0 = transform micro-pipeline (1 ds_read_b64 + 4 v_add_f32 + 1 ds_write_b64 per group)
//             1 = 4 v_add_f32   2 = 2 v_pk_add_f32   3 = 1 ds_read_b64   4 = 1 ds_write_b64   5 = 1 ds_write_b128   6 = 1 ds_read_b128
//             7 = 4 v_accvgpr_read-like v_mov   8 = 1 global_store_dwordx4   9 = 1 global_load_dwordx4 (L2 hit)
//  (2) CROSS-WAVE, two waves per SIMD (512 threads): waves 0-3 run the pure MFMA steps, waves 4-7 loop over filler groups only.
// Prints shader cycles per MFMA of wave 0 (floor 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void filler(f32x4& r0, f32x4& r1, f32x4& w, unsigned lds_addr, float* gp) {
    if constexpr (KIND == 0) {
        asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(*(f32x2*)&r1) : "v"(lds_addr) : "memory");
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(w[0]) : "v"(r0[0]), "v"(r0[1]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[1]) : "v"(r0[1]), "v"(r0[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(w[2]) : "v"(r0[2]), "v"(r0[3]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[3]) : "v"(r0[3]), "v"(r0[2]));
        asm volatile("ds_write_b64 %0, %1 offset:32768" ::"v"(lds_addr), "v"(*(f32x2*)&w) : "memory");
    } else if constexpr (KIND == 1) {
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(w[0]) : "v"(r0[0]), "v"(r0[1]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[1]) : "v"(r0[1]), "v"(r0[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(w[2]) : "v"(r0[2]), "v"(r0[3]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[3]) : "v"(r0[3]), "v"(r0[2]));
    } else if constexpr (KIND == 2) {
        const f32x2 lo = __builtin_shufflevector(r0, r0, 0, 1), hi = __builtin_shufflevector(r0, r0, 2, 3);
        f32x2 o0, o1;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o0) : "v"(lo), "v"(hi));
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o1) : "v"(lo), "v"(hi));
        w = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
    } else if constexpr (KIND == 3) {
        asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(*(f32x2*)&r1) : "v"(lds_addr) : "memory");
    } else if constexpr (KIND == 4) {
        asm volatile("ds_write_b64 %0, %1 offset:32768" ::"v"(lds_addr), "v"(*(f32x2*)&r0) : "memory");
    } else if constexpr (KIND == 5) {
        asm volatile("ds_write_b128 %0, %1 offset:32768" ::"v"(lds_addr * 2), "v"(r0) : "memory");
    } else if constexpr (KIND == 6) {
        asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r1) : "v"(lds_addr * 2) : "memory");
    } else if constexpr (KIND == 7) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(w[0]) : "v"(r0[0]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(w[1]) : "v"(r0[1]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(w[2]) : "v"(r0[2]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(w[3]) : "v"(r0[3]));
    } else if constexpr (KIND == 8) {
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(gp), "v"(r0) : "memory");
    } else if constexpr (KIND == 9) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r1) : "v"(gp) : "memory");
    } else if constexpr (KIND == 10) {
        asm volatile("s_nop 15" ::: "memory");
    }
}

// MODE 0: in-wave fillers (256 threads); MODE 1: cross-wave (512 threads, waves 4..7 do fillers only)
template <int KIND, int M, int PLACE, int MODE, int NOPS = 0>
__global__ __launch_bounds__(MODE ? 512 : 256) void probe(float* out, int iters, float a0, float b0, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + tid, b = b0;
    f32x4 r0 = {a, b, a + 1.f, b + 1.f}, r1 = r0, w = r0;
    const unsigned lds_addr = (unsigned)(tid & 255) * 8u;           // conflict-free b64 rows
    float* gp = out + 65536 + (size_t)(blockIdx.x * 512 + tid) * 4; // private 16 B per thread
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 0 || wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0 && (KIND == 0 || KIND == 3 || KIND == 4 || KIND == 5 || KIND == 6)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (MODE == 0 && (KIND == 8 || KIND == 9)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NOPS; ++q) asm volatile("s_nop 15" ::: "memory");
                    if (MODE == 0) {
                        if (PLACE == 0 && t == 0) {
#pragma unroll
                            for (int m = 0; m < M; ++m) filler<KIND>(r0, r1, w, lds_addr + (m & 1) * 2048, gp);
                        }
                        if (PLACE == 1 && t < M) filler<KIND>(r0, r1, w, lds_addr + (t & 1) * 2048, gp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (MODE == 0 && KIND == 0) { f32x4 tmp = r0; r0 = r1; r1 = tmp; }
            }
        }
    } else {
        for (int it = 0; it < iters * 8; ++it) {       // filler waves: M groups per "step", running for as long as it takes
            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(8)" ::: "memory");
#pragma unroll
            for (int m = 0; m < M; ++m) filler<KIND>(r0, r1, w, lds_addr + (m & 1) * 2048, gp);
            if (KIND == 0) { f32x4 tmp = r0; r0 = r1; r1 = tmp; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    float s = r1[0] + r1[1] + r1[2] + r1[3] + w[0] + w[1] + w[2] + w[3];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (MODE == 1 && tid == 256) cyc[4096 + blockIdx.x] = (unsigned long long)(t1 - t0);
}

static float* d_out;
static unsigned long long* d_cyc;

template <int KIND, int M, int PLACE, int MODE, int NOPS = 0>
void run(const char* name) {
    const int blocks = 256, iters = 300;
    const size_t lds = 100 * 1024;      // one workgroup per CU
    hipFuncSetAttribute((const void*)probe<KIND, M, PLACE, MODE, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((probe<KIND, M, PLACE, MODE, NOPS>), dim3(blocks), dim3(MODE ? 512 : 256), lds, 0, d_out, 10, 1.f, 2.f, d_cyc);
    hipLaunchKernelGGL((probe<KIND, M, PLACE, MODE, NOPS>), dim3(blocks), dim3(MODE ? 512 : 256), lds, 0, d_out, iters, 1.f, 2.f, d_cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    unsigned long long h[8192];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, s2 = 0;
    for (int i = 0; i < blocks; ++i) { s += (double)h[i]; s2 += (double)h[4096 + i]; }
    const double per = s / blocks / ((double)iters * 64);
    if (NOPS) printf("[%d x s_nop 15 behind every MFMA] ", NOPS);
    if (MODE == 0)
        printf("in-wave   kind %d %-28s M=%d place %d: %6.1f cycles per MFMA (+%5.1f per step of 8)\n", KIND, name, M, PLACE, per, (per - 64.0) * 8);
    else
        printf("x-wave    kind %d %-28s M=%d        : %6.1f cycles per MFMA of the MFMA wave; filler wave ran %.0f cycles per group\n", KIND, name, M, per,
               s2 / blocks / ((double)iters * 8 * (M ? M : 1)));
}

#define RUN_KIND(K, NAME)                       \
    run<K, 1, 0, 0>(NAME); run<K, 2, 0, 0>(NAME); run<K, 4, 0, 0>(NAME); run<K, 8, 0, 0>(NAME); \
    run<K, 1, 1, 0>(NAME); run<K, 2, 1, 0>(NAME); run<K, 4, 1, 0>(NAME); run<K, 8, 1, 0>(NAME); \
    run<K, 1, 0, 1>(NAME); run<K, 4, 0, 1>(NAME);

int main() {
    hipMalloc(&d_out, (65536 + 256 * 512 * 4) * sizeof(float) + 4096);
    hipMalloc(&d_cyc, 8192 * sizeof(unsigned long long));
    hipMemset(d_cyc, 0, 8192 * sizeof(unsigned long long));
    run<1, 0, 0, 0, 1>("bare MFMA steps"); run<1, 0, 0, 0, 2>("bare MFMA steps"); run<1, 0, 0, 0, 3>("bare MFMA steps"); run<1, 0, 0, 0, 4>("bare MFMA steps");
    run<0, 4, 0, 1, 1>("T micro-pipeline partner"); run<0, 4, 0, 1, 2>("T micro-pipeline partner"); run<0, 4, 0, 1, 3>("T micro-pipeline partner"); run<0, 4, 0, 1, 4>("T micro-pipeline partner");
    run<1, 4, 0, 1, 1>("16 VALU partner"); run<1, 4, 0, 1, 2>("16 VALU partner"); run<1, 4, 0, 1, 3>("16 VALU partner"); run<1, 4, 0, 1, 4>("16 VALU partner");
    run<2, 4, 0, 1, 2>("8 pk_add partner"); run<2, 4, 0, 1, 3>("8 pk_add partner");
    run<1, 0, 0, 0>("bare MFMA steps");
    run<1, 0, 0, 1>("bare MFMA steps, idle partner");
    RUN_KIND(0, "read64+4add+write64")
    RUN_KIND(1, "4 v_add_f32")
    RUN_KIND(2, "2 v_pk_add_f32")
    RUN_KIND(3, "ds_read_b64")
    RUN_KIND(4, "ds_write_b64")
    RUN_KIND(5, "ds_write_b128")
    RUN_KIND(6, "ds_read_b128")
    RUN_KIND(7, "4 v_mov_b32")
    RUN_KIND(8, "global_store_dwordx4")
    RUN_KIND(9, "global_load_dwordx4")
    return 0;
}

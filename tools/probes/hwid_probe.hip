// Probe (not product): what does HW_REG_HW_ID report for the waves of co-resident workgroups?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned id = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID[15:0]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id;
    // keep the workgroup resident for a while so that two of them share a CU
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(20);
    if (lds[(threadIdx.x + 1) & 255] < 0) out[0] = 0;
}
int main() {
    unsigned* d; unsigned h[512 * 4];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 60000);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 60000, 0, d);    // 60 KB LDS -> 2 workgroups per CU
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[16] = {0};
    for (int b = 0; b < 512; ++b) for (int w = 0; w < 4; ++w) hist[h[b * 4 + w] & 15]++;
    printf("wave_id histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
    for (int b = 0; b < 6; ++b) { printf("block %d:", b); for (int w = 0; w < 4; ++w) { unsigned v = h[b * 4 + w]; printf("  [wave_id %u simd %u cu %u se %u]", v & 15, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7); } printf("\n"); }
    for (int b = 256; b < 260; ++b) { printf("block %d:", b); for (int w = 0; w < 4; ++w) { unsigned v = h[b * 4 + w]; printf("  [wave_id %u simd %u cu %u se %u]", v & 15, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7); } printf("\n"); }
    return 0;
}

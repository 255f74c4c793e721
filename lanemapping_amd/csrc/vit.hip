// Multi-head self-attention core of the GFC-T / ViT block: softmax(Q K^T * scale) V per (batch, head).
//
// Replaces Attention.forward (baseline/models/backbone/vitsegnet.py:58-68): qkv is the fused
// to_qkv output [B*N, 3*H*64] (q | k | v, each laid out 'b n (h d)'), out is 'b n (h d)'.
// The projections themselves run on lm_conv2d_nhwc_mfma_f32.
//
// One workgroup = one (batch, head, 36-query chunk).  K (then V, re-using the same LDS) for the
// whole head is staged once: 324 x 64 fp32 = 83 KB of the CU's 160 KB LDS; the 36 x 324 score
// block stays in LDS as well, so scores never touch HBM.  Fixed summation order => deterministic.
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int QC = 36;      // query rows per workgroup
constexpr int KLD = DH + 1; // padded K/V row (floats): conflict-free column access

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N,
                                                        int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int SLD = N + 4;
    float* KV = smem;                  // [N][KLD]
    float* S = KV + N * KLD;           // [QC][SLD]
    float* Q = S + QC * SLD;           // [QC][DH]
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * QC, h = blockIdx.y, b = blockIdx.z;
    const int inner = heads * DH;
    const long row0 = (long)b * N;
    const int ld = 3 * inner;

    for (int i = tid; i < N * DH; i += 256) {
        const int n = i >> 6, d = i & 63;
        KV[n * KLD + d] = qkv[(row0 + n) * ld + inner + h * DH + d];
    }
    for (int i = tid; i < QC * DH; i += 256) {
        const int n = i >> 6, d = i & 63;
        Q[i] = (q0 + n < N) ? qkv[(row0 + q0 + n) * ld + h * DH + d] : 0.f;
    }
    __syncthreads();
    // scores
    for (int idx = tid; idx < QC * N; idx += 256) {
        const int i = idx / N, j = idx - i * N;
        const float* qr = Q + i * DH;
        const float* kr = KV + j * KLD;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc = fmaf(qr[d], kr[d], acc);
        S[i * SLD + j] = acc * scale;
    }
    __syncthreads();
    // V replaces K while the softmax runs on S
    for (int i = tid; i < N * DH; i += 256) {
        const int n = i >> 6, d = i & 63;
        KV[n * KLD + d] = qkv[(row0 + n) * ld + 2 * inner + h * DH + d];
    }
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < QC; i += 4) {
        float* sr = S + i * SLD;
        float m = -INFINITY;
        for (int j = lane; j < N; j += 64) m = fmaxf(m, sr[j]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int j = lane; j < N; j += 64) {
            const float e = expf(sr[j] - m);
            sr[j] = e;
            sum += e;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        for (int j = lane; j < N; j += 64) sr[j] *= inv;
    }
    __syncthreads();
    // out = P V : thread -> column d, rows i = wave, wave+4, ...
    const int d = lane;
    float acc[QC / 4];
#pragma unroll
    for (int r = 0; r < QC / 4; ++r) acc[r] = 0.f;
    for (int j = 0; j < N; ++j) {
        const float v = KV[j * KLD + d];
#pragma unroll
        for (int r = 0; r < QC / 4; ++r) acc[r] = fmaf(S[(wave + 4 * r) * SLD + j], v, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < QC / 4; ++r) {
        const int i = q0 + wave + 4 * r;
        if (i < N) out[(row0 + i) * inner + h * DH + d] = acc[r];
    }
}

}  // namespace

LM_API int lm_attention_f32(void* stream, const float* qkv, float* out, int B, int N, int heads, int dim_head, float scale) {
    LM_REQUIRE(qkv && out, "attention: null pointer");
    LM_REQUIRE(dim_head == DH, "attention: dim_head=%d must be %d", dim_head, DH);
    const size_t lds = ((size_t)N * KLD + (size_t)QC * (N + 4) + QC * DH) * sizeof(float);
    LM_REQUIRE(lds <= 160 * 1024, "attention: N=%d does not fit LDS", N);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        LM_HIP(hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(lm_cdiv(N, QC), heads, B), dim3(256), lds, (hipStream_t)stream, qkv, out, N, heads, scale);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

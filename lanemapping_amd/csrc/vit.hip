// Multi-head self-attention core of the GFC-T / ViT block: softmax(Q K^T * scale) V per (batch, head).
//
// Replaces Attention.forward (baseline/models/backbone/vitsegnet.py:58-68): qkv is the fused
// to_qkv output [B*N, 3*H*64] (q | k | v, each laid out 'b n (h d)'), out is 'b n (h d)'.
// The projections themselves run on lm_conv2d_nhwc_mfma_f32.
//
// Two kernels: attention_mfma_kernel (below) for the ViT block's 324 tokens, and this VALU kernel for every other
// sequence length (the RowRef head's few lane tokens).
// VALU kernel: one workgroup = one (batch, head, 36-query chunk).  K (then V, re-using the same LDS) for the
// whole head is staged once: 324 x 64 fp32 = 83 KB of the CU's 160 KB LDS; the 36 x 324 score
// block stays in LDS as well, so scores never touch HBM.  Fixed summation order => deterministic.
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int QC = 36;      // query rows per workgroup
constexpr int KLD = DH + 1; // padded K/V row (floats): conflict-free column access

// valid (optional, [B][NT] ints, NT <= 64 tokens per batch element): only the tokens with a non-zero flag take part as KEYS, in token
// order - the scores, the softmax and P V are computed on the compacted key list exactly as if the call had been made on the valid
// tokens alone (the RowRef head's lane tokens live on a fixed grid, their subset is data dependent).  Query rows are computed for
// every token; rows of a batch element without any valid token are written as zeros.
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int NT,
                                                        int heads, float scale, const int* __restrict__ valid) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int kidx[64];
    __shared__ int kcount;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * QC, h = blockIdx.y, b = blockIdx.z;
    const int inner = heads * DH;
    const long row0 = (long)b * NT;
    const int ld = 3 * inner;
    int N = NT;                        // keys
    if (valid) {
        if (tid < 64) {                // one wave: ballot compaction in token order
            const bool v = tid < NT && valid[(long)b * NT + tid] != 0;
            const unsigned long long m = __ballot(v);
            if (v) kidx[__popcll(m & ((1ull << tid) - 1ull))] = tid;
            if (tid == 0) kcount = (int)__popcll(m);
        }
        __syncthreads();
        N = kcount;
        if (N == 0) {
            for (int i = tid; i < QC * DH; i += 256) {
                const int n = q0 + (i >> 6);
                if (n < NT) out[(row0 + n) * inner + h * DH + (i & 63)] = 0.f;
            }
            return;
        }
    }
    const int SLD = N + 4;
    float* KV = smem;                  // [N][KLD]
    float* S = KV + N * KLD;           // [QC][SLD]
    float* Q = S + QC * SLD;           // [QC][DH]
#define LM_KROW(n) (valid ? kidx[n] : (n))

    for (int i = tid; i < N * DH; i += 256) {
        const int n = i >> 6, d = i & 63;
        KV[n * KLD + d] = qkv[(row0 + LM_KROW(n)) * ld + inner + h * DH + d];
    }
    for (int i = tid; i < QC * DH; i += 256) {
        const int n = i >> 6, d = i & 63;
        Q[i] = (q0 + n < NT) ? qkv[(row0 + q0 + n) * ld + h * DH + d] : 0.f;
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
        KV[n * KLD + d] = qkv[(row0 + LM_KROW(n)) * ld + 2 * inner + h * DH + d];
    }
#undef LM_KROW
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
        if (i < NT) out[(row0 + i) * inner + h * DH + d] = acc[r];
    }
}

// ---- MFMA version for the ViT block's shape (N = 324 tokens -> 11 key blocks of 32, dim_head 64) ---------------------
// One workgroup = (batch, head, 4 query tiles of 32), one wave per query tile, exact-fp32 v_mfma_f32_32x32x2_f32.
// The scores are computed TRANSPOSED, S^T = K Q^T (keys = MFMA rows, queries = MFMA columns), so that
//   * a lane's column is its query: the softmax max / sum are per-lane reductions over its 11 x 16 accumulator registers
//     plus one exchange with lane ^ 32, and the final 1/sum scales the lane's own output registers;
//   * the probabilities never leave the registers: for O^T = V^T P^T the accumulator register r of key block kb IS the
//     B operand of contraction step (kb, r) - the contraction simply visits the keys in the order the accumulator layout
//     holds them (lanes 0-31: key (r&3) + 8(r>>2), lanes 32-63: + 4), and the A operand reads V in that same order.
// K (padded rows, conflict-free 16-byte fragment reads, one read feeds 4 MFMAs) and then V share one 96 KB LDS buffer.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NKB = 11;            // key blocks of 32
constexpr int NPAD = NKB * 32;     // 352
constexpr int KP = DH + 4;         // padded K row (floats): 8 consecutive keys x 16 B cover all 32 banks

__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int heads,
                                                             float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // K: [NPAD][KP], later V: [NPAD][DH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int inner = heads * DH, ld = 3 * inner;
    const long row0 = (long)b * N;
    const int col = lane & 31, half = lane >> 5;
    const int q = (blockIdx.x * 4 + wave) * 32 + col;                 // this lane's query (MFMA column)
    // K -> LDS (rows >= N zero)
    for (int i = tid; i < NPAD * (DH / 4); i += 256) {
        const int n = i >> 4, c4 = (i & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N) v = *reinterpret_cast<const f32x4*>(qkv + (row0 + n) * ld + inner + h * DH + c4);
        *reinterpret_cast<f32x4*>(smem + n * KP + c4) = v;
    }
    // Q fragments: B[k = d][j = q]; lanes 0-31 carry d = 8g..8g+3, lanes 32-63 d = 8g+4..8g+7 (any bijection of k is fine
    // as long as the A operand uses the same one)
    f32x4 qf[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        qf[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < N) qf[g] = *reinterpret_cast<const f32x4*>(qkv + (row0 + q) * ld + h * DH + 8 * g + 4 * half);
    }
    __syncthreads();
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(smem + (kb * 32 + col) * KP + 8 * g + 4 * half);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[g][t], s[kb], 0, 0, 0);
        }
    }
    // softmax over the keys of this lane's query: register r of block kb is key kb*32 + (r&3) + 8(r>>2) + 4*half
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            s[kb][r] = key < N ? s[kb][r] * scale : -INFINITY;
            m = fmaxf(m, s[kb][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(s[kb][r] - m);      // exp(-inf) = 0 for the padded keys
            s[kb][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    __syncthreads();                                  // every wave is done with K
    for (int i = tid; i < NPAD * (DH / 4); i += 256) {
        const int n = i >> 4, c4 = (i & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N) v = *reinterpret_cast<const f32x4*>(qkv + (row0 + n) * ld + 2 * inner + h * DH + c4);
        *reinterpret_cast<f32x4*>(smem + n * DH + c4) = v;
    }
    __syncthreads();
    // O^T[d][q] = sum_key V[key][d] P^T[key][q]
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = smem + (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * DH + col;
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[kb][r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[kb][r], o[1], 0, 0, 0);
        }
    if (q < N) {
        const float inv = 1.0f / sum;
        float* orow = out + (row0 + q) * inner + h * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {       // registers 4*r4 .. 4*r4+3 are d = dt*32 + 8*r4 + 4*half .. +3
                f32x4 v = {o[dt][4 * r4] * inv, o[dt][4 * r4 + 1] * inv, o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv};
                *reinterpret_cast<f32x4*>(orow + dt * 32 + 8 * r4 + 4 * half) = v;
            }
    }
}

}  // namespace

LM_API int lm_attention_f32(void* stream, const float* qkv, float* out, int B, int N, int heads, int dim_head, float scale) {
    LM_REQUIRE(qkv && out, "attention: null pointer");
    LM_REQUIRE(dim_head == DH, "attention: dim_head=%d must be %d", dim_head, DH);
    if (N > NPAD - 32 && N <= NPAD) {       // the ViT block (324 tokens): matrix cores
        const size_t lds_m = (size_t)NPAD * KP * sizeof(float);
        if (int e = lm_ensure_dynamic_lds((const void*)attention_mfma_kernel, lds_m)) return e;
        hipLaunchKernelGGL(attention_mfma_kernel, dim3(lm_cdiv(lm_cdiv(N, 32), 4), heads, B), dim3(256), lds_m, (hipStream_t)stream, qkv, out,
                           N, heads, scale);
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    const size_t lds = ((size_t)N * KLD + (size_t)QC * (N + 4) + QC * DH) * sizeof(float);
    LM_REQUIRE(lds <= 160 * 1024, "attention: N=%d does not fit LDS", N);
    if (int e = lm_ensure_dynamic_lds((const void*)attention_kernel, lds)) return e;
    hipLaunchKernelGGL(attention_kernel, dim3(lm_cdiv(N, QC), heads, B), dim3(256), lds, (hipStream_t)stream, qkv, out, N, heads, scale,
                       (const int*)nullptr);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// The same attention with a key mask: valid [B][N] ints (device), N <= 64.  Per batch element only the tokens with a non-zero flag are
// keys (compacted in token order: the arithmetic of a call on the valid tokens alone); every token gets an output row.
LM_API int lm_attention_masked_f32(void* stream, const float* qkv, float* out, const int* valid, int B, int N, int heads, int dim_head,
                                   float scale) {
    LM_REQUIRE(qkv && out && valid, "attention_masked: null pointer");
    LM_REQUIRE(dim_head == DH && N >= 1 && N <= 64, "attention_masked: dim_head=%d must be %d, N=%d at most 64", dim_head, DH, N);
    const size_t lds = ((size_t)N * KLD + (size_t)QC * (N + 4) + QC * DH) * sizeof(float);
    if (int e = lm_ensure_dynamic_lds((const void*)attention_kernel, lds)) return e;
    hipLaunchKernelGGL(attention_kernel, dim3(lm_cdiv(N, QC), heads, B), dim3(256), lds, (hipStream_t)stream, qkv, out, N, heads, scale, valid);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// Column-proposal head glue kernels (ColumnProposal2.forward, live branch only:
// column_att=False, spatial_att=True; baseline/models/heads/polyline_fpn_vit_vertex_2.py:390-421).
//
//  lm_head_tokens       : for every proposal p, row h, window column w and channel c
//                           tok[(b,p,h), c*10+w] = avg_pool8x8( up_{(288,20)->(1152,80)}( seg_p ) )[h,w] * row_fea_pad[b,c,h,2p+w]
//                         (:392-405).  seg = bi_seg_proposal(relu(col_fea_up)) is computed once for the whole
//                         288x288 map; zero-padded columns (:383) evaluate to the conv bias.  The 1152x80
//                         per-proposal map (prop_bi_seg, 26.5 MB/tile) is never materialised.
//  lm_head_stage2       : second Conv1d of ext2 / cls2 / offset2 (:210,218,226) on the BN'd hidden rows
//  lm_head_proposal_conf: proposal_confidence Linear(23040 -> 2) (:200-204)
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FW = 10;      // prop_fea_width = prop_width + 2*half_buff
constexpr int NCH = 16;     // header_fea_dim

__device__ __forceinline__ void bilin_axis(int o, int in, int out, int& i0, int& i1, float& w0, float& w1) {
    const float scale = (float)(in - 1) / (float)(out - 1);
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    w0 = 1.f - w1;
}

// 8 x 8 average of the bilinearly up-sampled (x 4 rows, x 4 columns) proposal window at token (h, w): `tap(y, x, ok)` returns the
// segmentation value of source row y, window column x (ok: inside the un-padded map, else the conv bias).  ONE expression for both
// kernels below, so the compiler forms the same multiply-adds in both: identical bits.
template <typename Tap>
__device__ __forceinline__ float pooled_window(int h, int w, int Hs, int Hr, int win, int col0, int Ws, float seg_bias, Tap tap) {
    float sum = 0.f;
    for (int r = 0; r < 8; ++r) {
        int y0, y1;
        float wy0, wy1;
        bilin_axis(8 * h + r, Hs, 8 * Hr, y0, y1, wy0, wy1);
        for (int q = 0; q < 8; ++q) {
            int x0, x1;
            float wx0, wx1;
            bilin_axis(8 * w + q, win, 8 * FW, x0, x1, wx0, wx1);
            const int c0 = col0 + x0, c1 = col0 + x1;
            const bool ok0 = (unsigned)c0 < (unsigned)Ws, ok1 = (unsigned)c1 < (unsigned)Ws;
            const float v00 = ok0 ? tap(y0, x0, c0) : seg_bias;
            const float v01 = ok1 ? tap(y0, x1, c1) : seg_bias;
            const float v10 = ok0 ? tap(y1, x0, c0) : seg_bias;
            const float v11 = ok1 ? tap(y1, x1, c1) : seg_bias;
            sum += wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
        }
    }
    return sum * (1.0f / 64.0f);
}

__device__ __forceinline__ void write_tokens(const float* __restrict__ row, float* __restrict__ tok, float pooled, int b, int p, int h, int w,
                                             int P, int Hr, int Wr, int prop_width, int half_buff) {
    const int rc = prop_width * p + w - half_buff;      // column in the un-padded row feature map
    float* tr = tok + (((long)b * P + p) * Hr + h) * (NCH * FW) + w;
    if ((unsigned)rc < (unsigned)Wr) {
        const float* rp = row + (((long)b * Hr + h) * Wr + rc) * NCH;
#pragma unroll
        for (int c = 0; c < NCH; ++c) tr[c * FW] = pooled * rp[c];
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) tr[c * FW] = pooled * 0.f;
    }
}

// seg [B,Hs,Ws] (Hs=Ws=288), row [B,Hr,Wr,16] NHWC (Hr=Wr=144), tok [B*P*Hr, 160]
// (rounds 1-3: one thread per token, its 256 taps gathered from global memory - the vector-memory path serves a 64-lane gather of
// 4-byte elements at a fraction of its line rate: 0.275 ms per 16 tiles; kept behind LM_HEAD_TOKENS_GATHER=1)
__global__ __launch_bounds__(256) void head_tokens_kernel(const float* __restrict__ seg, const float* __restrict__ row,
                                                          float* __restrict__ tok, float seg_bias, int P, int Hr, int Wr,
                                                          int prop_width, int half_buff, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % FW);
    long t = i / FW;
    const int h = (int)(t % Hr);
    t /= Hr;
    const int p = (int)(t % P);
    const int b = (int)(t / P);
    const int Hs = 2 * Hr, Ws = 2 * Wr;
    const int win = 2 * FW;                             // 20 source columns per proposal
    const int col0 = 2 * prop_width * p - 2 * half_buff;   // first source column of the window (may be < 0)
    const float* sb = seg + (long)b * Hs * Ws;
    const float pooled = pooled_window(h, w, Hs, Hr, win, col0, Ws, seg_bias, [&](int y, int, int c) { return sb[(long)y * Ws + c]; });
    write_tokens(row, tok, pooled, b, p, h, w, P, Hr, Wr, prop_width, half_buff);
}

// Round 4: one workgroup per (image, proposal, block of HT token rows); the source rows its tokens touch x the proposal's 20 window
// columns are staged in LDS once (<= HT_ROWS x 20 floats), the 256 taps per token come from there.  Same expression, same order
// (pooled_window): bit-identical to the gather kernel.
constexpr int HT = 24;             // token rows per workgroup (x 10 window columns = 240 of 256 threads)
constexpr int HT_ROWS = 64;        // source rows staged at most (24 token rows span 8 * 24 / 4 + 2 = 50)
__global__ __launch_bounds__(256) void head_tokens_lds_kernel(const float* __restrict__ seg, const float* __restrict__ row,
                                                              float* __restrict__ tok, float seg_bias, int P, int Hr, int Wr,
                                                              int prop_width, int half_buff, int hblocks) {
    __shared__ float win_s[HT_ROWS * 2 * FW];
    const int tid = threadIdx.x;
    const int hb = blockIdx.x % hblocks, bp = blockIdx.x / hblocks;
    const int p = bp % P, b = bp / P;
    const int Hs = 2 * Hr, Ws = 2 * Wr;
    const int win = 2 * FW;
    const int col0 = 2 * prop_width * p - 2 * half_buff;
    const int h0 = hb * HT, h1 = min(h0 + HT, Hr);
    int ya, yb, dummy;
    float f0, f1;
    bilin_axis(8 * h0, Hs, 8 * Hr, ya, dummy, f0, f1);                  // first source row of the block
    bilin_axis(8 * (h1 - 1) + 7, Hs, 8 * Hr, dummy, yb, f0, f1);        // last one (second tap of the last output row)
    const int nrows = yb - ya + 1;                                       // (the launcher checks the bound HT_ROWS)
    const float* sb = seg + (long)b * Hs * Ws;
    for (int i = tid; i < nrows * win; i += 256) {
        const int r = i / win, x = i - r * win;
        const int c = col0 + x;
        win_s[i] = (unsigned)c < (unsigned)Ws ? sb[(long)(ya + r) * Ws + c] : seg_bias;
    }
    __syncthreads();
    const int hl = tid / FW, w = tid - hl * FW;
    const int h = h0 + hl;
    if (hl >= HT || h >= h1) return;
    const float pooled = pooled_window(h, w, Hs, Hr, win, col0, Ws, seg_bias, [&](int y, int x, int) { return win_s[(y - ya) * win + x]; });
    write_tokens(row, tok, pooled, b, p, h, w, P, Hr, Wr, prop_width, half_buff);
}

// hid [M, ldh] (ext | cls | off hidden, D each) -> ext2 [M,3], cls2 [M,10], off2 [M,10]
// grid (ceil(M / 256), 3): one thread = one row of one branch, all of that branch's outputs in registers; the branch is uniform per
// block, so its weight rows come through scalar loads (was: one thread per output re-reading the row 3 / 10 times with 4-byte loads
// and a 64-bit division per thread, 0.31 ms for M = 82,944).  Same k-ascending fmaf chain per output as before: identical bits.
template <int NOUT>
__device__ __forceinline__ void stage2_rows(const float* __restrict__ hr, int D, const float* __restrict__ w, const float* __restrict__ b,
                                            float* __restrict__ out) {
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int k = 0; k < D; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(hr + k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int n = 0; n < NOUT; ++n) acc[n] = fmaf(v[e], w[n * D + k + e], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) out[n] = acc[n] + b[n];
}

__global__ __launch_bounds__(256) void head_stage2_kernel(const float* __restrict__ hid, int ldh, int D,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ ext2, float* __restrict__ cls2,
                                                          float* __restrict__ off2, long M) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int br = blockIdx.y;                                  // rows of w2: 3 ext, 10 cls, 10 off
    const float* hr = hid + m * ldh + br * D;
    if (br == 0) stage2_rows<3>(hr, D, w2, b2, ext2 + m * 3);
    else if (br == 1) stage2_rows<10>(hr, D, w2 + 3L * D, b2 + 3, cls2 + m * 10);
    else stage2_rows<10>(hr, D, w2 + 13L * D, b2 + 13, off2 + m * 10);
}

// Round 4: the same rows through LDS.  In the kernel above a lane reads ITS row with 16-byte loads one row pitch (1.2 KB) apart from its
// neighbours' - 64 cache lines per wave instruction - and the weights come 40 bytes at a time through the scalar cache: 0.19 ms per 16
// tiles for 0.2 GB.  Here a workgroup of 128 threads stages its 128 rows x D floats of one branch with coalesced loads (row stride D + 1
// in LDS: conflict-free 4-byte reads), then every thread walks its row: the same k-ascending fmaf chain, identical bits.
constexpr int S2R = 128;
template <int NOUT>
__device__ __forceinline__ void stage2_rows_lds(const float* hl, int D, const float* __restrict__ w, const float* __restrict__ b,
                                                float* __restrict__ out) {
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int k = 0; k < D; k += 4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = hl[k + e];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int n = 0; n < NOUT; ++n) acc[n] = fmaf(v[e], w[n * D + k + e], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) out[n] = acc[n] + b[n];
}

__global__ __launch_bounds__(S2R) void head_stage2_lds_kernel(const float* __restrict__ hid, int ldh, int D, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float* __restrict__ ext2,
                                                             float* __restrict__ cls2, float* __restrict__ off2, long M) {
    extern __shared__ float rows_s[];                          // [S2R][D + 1]
    const int tid = threadIdx.x;
    const long m0 = (long)blockIdx.x * S2R;
    const int br = blockIdx.y;
    const int d4 = D / 4, nrow = (int)min((long)S2R, M - m0);
    const float* hb = hid + m0 * ldh + br * D;
    for (int i0 = 0; i0 < nrow * d4; i0 += S2R * 4) {          // 16-byte chunks, four in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * S2R + tid, nrow * d4 - 1);
            const int r = i / d4, c = i - r * d4;
            v[u] = *reinterpret_cast<const f32x4*>(hb + (long)r * ldh + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * S2R + tid;
            if (i < nrow * d4) {
                const int r = i / d4, c = i - r * d4;
#pragma unroll
                for (int e = 0; e < 4; ++e) rows_s[r * (D + 1) + 4 * c + e] = v[u][e];
            }
        }
    }
    __syncthreads();
    const long m = m0 + tid;
    if (m >= M) return;
    const float* hl = rows_s + tid * (D + 1);
    if (br == 0) stage2_rows_lds<3>(hl, D, w2, b2, ext2 + m * 3);
    else if (br == 1) stage2_rows_lds<10>(hl, D, w2 + 3L * D, b2 + 3, cls2 + m * 10);
    else stage2_rows_lds<10>(hl, D, w2 + 13L * D, b2 + 13, off2 + m * 10);
}

// tok [B*P, L] (L = Hr*160, already in (h, cw) order), wt [2][L] -> conf [B*P, 2]
__global__ __launch_bounds__(256) void head_conf_kernel(const float* __restrict__ tok, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ conf, int L) {
    __shared__ float red[2][256];
    const long bp = blockIdx.x;
    const float* tr = tok + bp * L;
    float a0 = 0.f, a1 = 0.f;
    for (int k = threadIdx.x; k < L; k += 256) {
        const float v = tr[k];
        a0 = fmaf(v, wt[k], a0);
        a1 = fmaf(v, wt[L + k], a1);
    }
    red[0][threadIdx.x] = a0;
    red[1][threadIdx.x] = a1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) conf[bp * 2 + threadIdx.x] = red[threadIdx.x][0] + bias[threadIdx.x];
}

}  // namespace

LM_API int lm_head_tokens(void* stream, const float* seg, const float* row_nhwc16, float* tok, float seg_bias,
                          int B, int P, int Hr, int Wr, int prop_width, int half_buff) {
    LM_REQUIRE(seg && row_nhwc16 && tok, "head_tokens: null pointer");
    LM_REQUIRE(prop_width + 2 * half_buff == FW, "head_tokens: prop_fea_width must be %d", FW);
    const long total = (long)B * P * Hr * FW;
    static const bool gather = [] { const char* e = getenv("LM_HEAD_TOKENS_GATHER"); return e && atoi(e) != 0; }();
    const int hblocks = lm_cdiv(Hr, HT);
    // source rows a block of HT token rows can touch: (8 HT - 1) * (2 Hr - 1) / (8 Hr - 1) + 3
    const bool fits = (long)(8 * HT - 1) * (2 * Hr - 1) / (8 * Hr - 1) + 3 <= HT_ROWS && (long)B * P * hblocks < (1L << 31);
    if (!gather && fits) {
        hipLaunchKernelGGL(head_tokens_lds_kernel, dim3((unsigned)((long)B * P * hblocks)), dim3(256), 0, (hipStream_t)stream, seg, row_nhwc16, tok,
                           seg_bias, P, Hr, Wr, prop_width, half_buff, hblocks);
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    hipLaunchKernelGGL(head_tokens_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       seg, row_nhwc16, tok, seg_bias, P, Hr, Wr, prop_width, half_buff, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_head_stage2(void* stream, const float* hid, int ldh, int D, const float* w2, const float* b2,
                          float* ext2, float* cls2, float* off2, long M) {
    LM_REQUIRE(hid && w2 && b2 && ext2 && cls2 && off2, "head_stage2: null pointer");
    LM_REQUIRE(D % 4 == 0 && ldh % 4 == 0, "head_stage2: D=%d and ldh=%d must be multiples of 4", D, ldh);
    static const bool direct = [] { const char* e = getenv("LM_HEAD_STAGE2_DIRECT"); return e && atoi(e) != 0; }();
    const size_t lds = (size_t)S2R * (D + 1) * sizeof(float);
    if (!direct && lds <= 64 * 1024 && M < (1L << 31)) {
        if (int e = lm_ensure_dynamic_lds((const void*)head_stage2_lds_kernel, lds)) return e;
        hipLaunchKernelGGL(head_stage2_lds_kernel, dim3(lm_cdiv(M, S2R), 3), dim3(S2R), lds, (hipStream_t)stream, hid, ldh, D, w2, b2, ext2, cls2,
                           off2, M);
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    hipLaunchKernelGGL(head_stage2_kernel, dim3(lm_cdiv(M, 256), 3), dim3(256), 0, (hipStream_t)stream,
                       hid, ldh, D, w2, b2, ext2, cls2, off2, M);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_head_proposal_conf(void* stream, const float* tok, const float* wt, const float* bias, float* conf,
                                 int BP, int L) {
    LM_REQUIRE(tok && wt && bias && conf, "head_conf: null pointer");
    hipLaunchKernelGGL(head_conf_kernel, dim3(BP), dim3(256), 0, (hipStream_t)stream, tok, wt, bias, conf, L);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

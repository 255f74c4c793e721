// Glue kernels of the K-Lane "RowRef" head (config 4; baseline/models/heads/row_shared_not_reduc_ref.py).
// The Conv1d/BN1d stacks, the token Linear layers and the lane-token transformer run on lm_conv2d_nhwc_mfma_f32 /
// lm_layernorm_rows / lm_attention_f32; this file holds what is specific to the head:
//   lm_softmax_rows    softmax(dim=2) of the ext / cls logits                                   (:179-180, :239-240)
//   lm_rowref_select   per (b, lane): mean_h ext[b,h,lane,0], the lane-selection flag mean > thr_ext, argmax_w cls[b,h,lane,:]  (:199-204)
//   lm_rowref_gather   5-column window around the arg-max column of every row -> token input     (:207-211)
//   lm_rowref_scatter  write the refined windows back, later lanes over earlier ones, lane i only on rows
//                      0 .. 142-i: the reference's leaked/shrinking loop variable (:227-230)
// Round 3: the data-dependent lane set no longer goes through the host.  The reference compacts the selected (b, lane) pairs into a
// token list (:199-204); here the tokens live on the FIXED grid t = b * L + lane, `valid[b][lane]` says which of them exist, gather /
// token MLP / transformer / expansion run on all B * L rows (a few wasted rows of tiny GEMMs), the attention core compacts the valid
// keys of a tile in lane order (lm_attention_masked_f32: same arithmetic as on the compacted list) and the scatter derives a lane's
// rank among the selected lanes of its tile - what the shrinking-range quirk is indexed by - from the flags.
//   lm_rowref_decode   row exists iff argmax(ext2)==0, column = argmax(cls2) -> conf / cls maps   (:334-363)
// Layouts: feature x [B,H,W,8] NHWC; ext [B,H,L,2]; cls [B,H,L,W]; tokens [T, 8*H*5] in (c h w) order.
#include "common.h"

namespace {

constexpr int CF = 8;     // dim_feat
constexpr int KW = 5;     // 2*off_grid + 1

__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long rows, int cols) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* r = x + row * cols;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, r[j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const float e = expf(r[j] - m);
        r[j] = e;
        s += e;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    for (int j = lane; j < cols; j += 64) r[j] = r[j] / s;
}

// grid (L, B), 256 threads
__global__ __launch_bounds__(256) void rowref_select_kernel(const float* __restrict__ ext, const float* __restrict__ cls,
                                                            float* __restrict__ mean_out, int* __restrict__ valid, float thr,
                                                            int* __restrict__ corr, int H, int W, int L) {
    __shared__ float red[256];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float s = 0.f;
    for (int h = tid; h < H; h += 256) {
        s += ext[(((long)b * H + h) * L + c) * 2];
        const float* p = cls + (((long)b * H + h) * L + c) * W;
        int best = 0;
        float bv = p[0];
        for (int w = 1; w < W; ++w)
            if (p[w] > bv) {
                bv = p[w];
                best = w;
            }
        corr[((long)b * L + c) * H + h] = best;
    }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) {
        const float m = red[0] / (float)H;
        mean_out[b * L + c] = m;
        valid[b * L + c] = m > thr ? 1 : 0;                    // (:199-200: exist_mean > thr_ext, fp32 like the reference's tensor compare)
    }
}

// token t = b * L + lane (fixed grid).  tok[t][(cf*H + h)*5 + j] = x_pad[b, cf, h, corr + j]
__global__ __launch_bounds__(256) void rowref_gather_kernel(const float* __restrict__ x, const int* __restrict__ corr,
                                                            float* __restrict__ tok, int H, int W, int L, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over B*L*H*KW
    if (i >= total) return;
    const int j = (int)(i % KW);
    const int h = (int)((i / KW) % H);
    const int t = (int)(i / ((long)KW * H));
    const int b = t / L, c = t - b * L;
    const int w = corr[((long)b * L + c) * H + h] + j - KW / 2;
    float* o = tok + (long)t * (CF * H * KW) + (long)h * KW + j;
    if ((unsigned)w < (unsigned)W) {
        const float* p = x + (((long)b * H + h) * W + w) * CF;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) o[(long)cf * H * KW] = p[cf];
    } else {
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) o[(long)cf * H * KW] = 0.f;
    }
}

// Selected lane number n (0-based among the selected lanes of its tile, lane order) is written on rows h < H-1-n only; among covering
// lanes the last one wins.  No lane of the tile selected: y = x.
__global__ __launch_bounds__(256) void rowref_scatter_kernel(const float* __restrict__ x, const float* __restrict__ tok,
                                                             const int* __restrict__ corr, const int* __restrict__ valid,
                                                             float* __restrict__ y, int H, int W, int L, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over B*H*W
    if (i >= total) return;
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int b = (int)(i / ((long)W * H));
    const float* src = x + i * CF;
    long tsel = -1;
    int jsel = 0;
    int n = 0;
    for (int c = 0; c < L; ++c) n += valid[b * L + c] ? 1 : 0;
    for (int c = L - 1; c >= 0; --c) {                      // from the last selected lane down: the first hit is the winner
        if (!valid[b * L + c]) continue;
        --n;                                                // rank of lane c among the selected lanes of the tile
        if (h >= H - 1 - n) continue;
        const int j = w - corr[((long)b * L + c) * H + h] + KW / 2;
        if ((unsigned)j < (unsigned)KW) {
            tsel = (long)b * L + c;
            jsel = j;
            break;
        }
    }
    float* o = y + i * CF;
    if (tsel >= 0) {
        const float* p = tok + tsel * (long)(CF * H * KW) + (long)h * KW + jsel;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) o[cf] = p[(long)cf * H * KW];
    } else {
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) o[cf] = src[cf];
    }
}

__global__ __launch_bounds__(256) void rowref_decode_kernel(const float* __restrict__ ext, const float* __restrict__ cls,
                                                            unsigned char* __restrict__ conf, unsigned char* __restrict__ cmap,
                                                            int* __restrict__ col_idx, int H, int W, int L, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over B*L*H
    if (i >= total) return;
    const int h = (int)(i % H);
    const int c = (int)((i / H) % L);
    const int b = (int)(i / ((long)H * L));
    const float* e = ext + (((long)b * H + h) * L + c) * 2;
    int col = -1;
    if (!(e[1] > e[0])) {                                   // argmax == 0 (first maximum on ties)
        const float* p = cls + (((long)b * H + h) * L + c) * W;
        col = 0;
        float bv = p[0];
        for (int w = 1; w < W; ++w)
            if (p[w] > bv) {
                bv = p[w];
                col = w;
            }
        if (cmap) {
            cmap[(((long)b * (L + 1) + c) * H + h) * W + col] = 1;
            cmap[(((long)b * (L + 1) + L) * H + h) * W + col] = 1;
        }
        if (conf) conf[((long)b * H + h) * W + col] = 1;
    }
    col_idx[i] = col;
}

}  // namespace

LM_API int lm_softmax_rows(void* stream, float* x, long rows, int cols) {
    LM_REQUIRE(x && cols >= 1, "softmax_rows: bad args");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(lm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, cols);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_rowref_select(void* stream, const float* ext, const float* cls, float* mean_out, int* valid, float thr_ext, int* corr,
                            int B, int H, int W, int L) {
    LM_REQUIRE(ext && cls && mean_out && valid && corr, "rowref_select: null pointer");
    hipLaunchKernelGGL(rowref_select_kernel, dim3(L, B), dim3(256), 0, (hipStream_t)stream, ext, cls, mean_out, valid, thr_ext, corr, H, W, L);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_rowref_gather(void* stream, const float* x_nhwc8, const int* corr, float* tok, int B, int H, int W, int L) {
    LM_REQUIRE(x_nhwc8 && corr && tok && B >= 1 && L >= 1, "rowref_gather: bad args");
    const long total = (long)B * L * H * KW;
    hipLaunchKernelGGL(rowref_gather_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x_nhwc8, corr, tok, H, W, L, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_rowref_scatter(void* stream, const float* x_nhwc8, const float* tok, const int* corr, const int* valid,
                             float* y_nhwc8, int B, int H, int W, int L) {
    LM_REQUIRE(x_nhwc8 && tok && corr && valid && y_nhwc8, "rowref_scatter: null pointer");
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(rowref_scatter_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x_nhwc8, tok, corr, valid,
                       y_nhwc8, H, W, L, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_rowref_decode(void* stream, const float* ext2, const float* cls2, unsigned char* conf, unsigned char* cls_map,
                            int* col_idx, int B, int H, int W, int L) {
    // conf / cls_map (the reference's dense one-hot maps, :334-363) are optional: the tile pipeline only needs col_idx
    LM_REQUIRE(ext2 && cls2 && col_idx, "rowref_decode: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (conf) LM_HIP(hipMemsetAsync(conf, 0, (size_t)B * H * W, s));
    if (cls_map) LM_HIP(hipMemsetAsync(cls_map, 0, (size_t)B * (L + 1) * H * W, s));
    const long total = (long)B * L * H;
    hipLaunchKernelGGL(rowref_decode_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, s, ext2, cls2, conf, cls_map, col_idx, H, W, L, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

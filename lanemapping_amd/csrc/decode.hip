// Decode kernels: softmax / argmax / threshold of the raw network outputs and the top-K endpoint
// selection.  Replaces the device half of ColumnProposal2.get_exist_coor_endp_dict
// (baseline/models/heads/polyline_fpn_vit_vertex_2.py:602-759) and of PostProjector2.infer_validate
// (baseline/models/pcencoder/postprojector.py:115-183).
//
//  lm_decode_proposals : :610 (prop_conf softmax), :694-697 (existence 3-way softmax + thresholds),
//                        :701-702 (10-way softmax + argmax), :726-738 (idx + offset, clamp, + 2p-4).
//                        Replaces the 10 368-iteration Python loop per tile by one lane per (b,p,h).
//  lm_decode_orient    : :615 argmax over the 11 orientation channels.
//  lm_decode_semantic  : :627-632 3-way softmax over the 1152x1152 map, class thresholds, s1+s2.
//  lm_segmentor_semantic: postprojector.py:122-127 (raw-logit thresholds, quirk C11).
//  lm_endp_topk        : :647-668 sigmoid of the cropped endpoint logits and the K best pixels in
//                        (score desc, flat index asc) order, by a 3-level radix select on the fp32 bit
//                        pattern + one in-LDS bitonic sort (replaces a full argsort of 1.24 M scores).
// Ties: argmax -> lowest index; equal scores -> lower flat index first.
#include "common.h"

#define LM_PACK_MAX_SEGMENTS 8

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int N>
__device__ __forceinline__ int softmax_argmax(const float* in, float* out) {
    float m = in[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = fmaxf(m, in[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[i] = expf(in[i] - m);
        s += out[i];
    }
    int best = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[i] = out[i] / s;
        if (out[i] > out[best]) best = i;
    }
    return best;
}

__global__ __launch_bounds__(256) void decode_proposals_kernel(
    const float* __restrict__ pconf, const float* __restrict__ ext2, const float* __restrict__ cls2,
    const float* __restrict__ off2, float* __restrict__ prop_conf, float* __restrict__ v_ext,
    float* __restrict__ cls_conf, int* __restrict__ cls_idx, double* __restrict__ cls_offset, int P, int R,
    float exist_thre, int prop_width, int half_buff, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int h = (int)(i % R);
    const long bp = i / R;
    const int p = (int)(bp % P);
    if (h == 0) {
        float o[2];
        softmax_argmax<2>(pconf + bp * 2, o);
        prop_conf[bp * 2] = o[0];
        prop_conf[bp * 2 + 1] = o[1];
    }
    float e[3];
    softmax_argmax<3>(ext2 + i * 3, e);
    float v = 0.f;
    if (e[1] > e[2] && e[1] > exist_thre) v = 1.f;
    if (e[2] > e[1] && e[2] > exist_thre) v = 2.f;
    v_ext[i] = v;
    float c[10];
    const int idx = softmax_argmax<10>(cls2 + i * 10, c);
#pragma unroll
    for (int k = 0; k < 10; ++k) cls_conf[i * 10 + k] = c[k];
    cls_idx[i] = idx;
    const float fsum = (float)idx + off2[i * 10 + idx];   // fp32 sum stored in f64 (:726)
    double co = (double)fsum;
    if (co > 10.0) co = 10.0;                             // :732 (prop_w = 10)
    cls_offset[i] = co + (double)(prop_width * p - half_buff);
}

__global__ __launch_bounds__(256) void decode_orient_kernel(const float* __restrict__ x, int ldx, int C,
                                                            unsigned char* __restrict__ y, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float* xp = x + i * ldx;
    int best = 0;
    float bv = xp[0];
    for (int c = 1; c < C; ++c) {
        const float v = xp[c];
        if (v > bv) {
            bv = v;
            best = c;
        }
    }
    y[i] = (unsigned char)best;
}

// planar logits [B,3,HW] -> sem u8 [B,HW], biseg f32 [B,HW], rows f32 [B, H/8, W] (image rows 3::8)
__global__ __launch_bounds__(256) void decode_semantic_kernel(const float* __restrict__ logit, unsigned char* __restrict__ sem,
                                                              float* __restrict__ biseg, float* __restrict__ rows, int H, int W,
                                                              float thre, int raw_mode, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over B*H*W
    if (i >= total) return;
    const long HW = (long)H * W;
    const long b = i / HW, r = i - b * HW;
    const float* lp = logit + b * 3 * HW + r;
    const float l0 = lp[0], l1 = lp[HW], l2 = lp[2 * HW];
    float s1, s2;
    if (raw_mode) {
        s1 = l1;
        s2 = l2;
    } else {
        const float m = fmaxf(l0, fmaxf(l1, l2));
        const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
        const float s = (e0 + e1) + e2;
        s1 = e1 / s;
        s2 = e2 / s;
    }
    unsigned char c = 0;
    if (s1 > s2 && s1 > thre) c = 1;
    if (s2 > s1 && s2 > thre) c = 2;
    sem[i] = c;
    if (biseg) {
        const float bs = s1 + s2;
        biseg[i] = bs;
        const int y = (int)(r / W);
        if (rows && (y & 7) == 3) rows[(b * (H / 8) + (y >> 3)) * W + (r - (long)y * W)] = bs;
    }
}

// ------------------------------------------------------------------------------------ endpoint top-K
constexpr int NB = 1024;        // bins per radix level
constexpr int CAP = 4096;       // candidate capacity per tile
struct TopkParams {
    const float* logit;          // [B,1,H,W] planar
    unsigned* hist;              // [B][3][NB]
    unsigned* cand_cnt;          // [B]
    unsigned long long* cand;    // [B][CAP]
    int H, W, clip, K;
    LmFastDiv div_wc;            // flat crop index -> (row, column) without a run-time (64-bit) division per element
};

__device__ __forceinline__ unsigned score_key(float logit) {
    const float s = 1.0f / (1.0f + expf(-logit));   // torch.sigmoid in fp32
    return __float_as_uint(s);                      // s in [0,1]: unsigned order == float order
}

__device__ __forceinline__ unsigned level_bin(unsigned key, int level) {
    return level == 0 ? (key >> 20) : (level == 1 ? ((key >> 10) & 1023u) : (key & 1023u));
}

// Walk hist level `lv` from the top bin down: the bin where the running count reaches `need`, and in `need` what is still wanted inside
// that bin - by a whole workgroup (>= 256 threads; every thread gets the result): thread t < 256 owns bins 4t .. 4t+3, an LDS suffix
// scan over the 256 four-bin sums finds the owner of the threshold bin.  (Rounds 1-3 walked the bins in one thread: up to 1024 dependent loads per
// level, in EVERY workgroup of the later passes - pass 3, three levels, took 157 us per 16 tiles against 39 us for pass 0.)
// tmp: 258 words of LDS.  Conventions of the walk: (NB - 1, 0) for need == 0 and (0, 0) when fewer than `need` keys exist.
__device__ void pick_bin_block(const unsigned* h, unsigned& need, unsigned& bin, unsigned* tmp) {
    const int t = threadIdx.x;
    unsigned c[4] = {0u, 0u, 0u, 0u}, s = 0;
    if (t < 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(h + 4 * t);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        s = v.x + v.y + v.z + v.w;
        tmp[t] = s;
    }
    if (t == 0) {
        tmp[256] = need == 0 ? (unsigned)(NB - 1) : 0u;
        tmp[257] = 0u;
    }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                      // inclusive suffix sums: tmp[t] = keys in bins 4t .. NB-1
        unsigned add = 0;
        if (t + off < 256) add = tmp[t + off];
        __syncthreads();
        if (t < 256) tmp[t] += add;
        __syncthreads();
    }
    if (t < 256 && need > 0) {
        const unsigned incl = tmp[t], above = incl - s;
        if (above < need && incl >= need) {                        // exactly one thread
            unsigned acc = above;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                if (acc + c[k] >= need) {
                    tmp[256] = (unsigned)(4 * t + k);
                    tmp[257] = need - acc;
                    break;
                }
                acc += c[k];
            }
        }
    }
    __syncthreads();
    bin = tmp[256];
    need = tmp[257];
    __syncthreads();
}

__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

template <int LEVEL>   // 0,1,2 = histogram passes; 3 = candidate compaction
__global__ __launch_bounds__(256) void topk_pass_kernel(TopkParams p) {
    __shared__ unsigned lh[NB];
    __shared__ unsigned pb_tmp[258];
    const int b = blockIdx.y;
    const int Hc = p.H - 2 * p.clip, Wc = p.W - 2 * p.clip;
    const long n = (long)Hc * Wc;
    unsigned* hist = p.hist + (long)b * 3 * NB;
    if (LEVEL < 3)
        for (int i = threadIdx.x; i < NB; i += 256) lh[i] = 0;
    unsigned prefix = 0;     // key bits fixed by previous levels
    bool take_ties = false;
    {
        unsigned need = (unsigned)p.K, bin = 0;
#pragma unroll
        for (int lv = 0; lv < LEVEL && lv < 3; ++lv) {
            pick_bin_block(hist + lv * NB, need, bin, pb_tmp);
            prefix |= bin << (20 - 10 * lv);
        }
        // LEVEL 3: `need` = how many keys EQUAL to the threshold T are still wanted, hist[2][bin] = how many exist.  If more exist
        // than are wanted, the ties are ranked by index in topk_ties_kernel (lowest flat index first) and this pass skips them
        if (LEVEL == 3) take_ties = hist[2 * NB + bin] == need;
    }
    __syncthreads();
    const float* lp = p.logit + (long)b * p.H * p.W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int y = (int)lm_fastdiv((unsigned)i, p.div_wc), x = (int)((unsigned)i - (unsigned)y * (unsigned)Wc);
        const unsigned key = score_key(lp[(long)(y + p.clip) * p.W + x + p.clip]);
        if (LEVEL == 0) {
            atomicAdd(&lh[key >> 20], 1u);
        } else if (LEVEL == 1) {
            if ((key >> 20) == (prefix >> 20)) atomicAdd(&lh[(key >> 10) & 1023u], 1u);
        } else if (LEVEL == 2) {
            if ((key >> 10) == (prefix >> 10)) atomicAdd(&lh[key & 1023u], 1u);
        } else {
            if (key > prefix || (key == prefix && take_ties)) {   // prefix == exact threshold key T; at most K candidates in total
                const unsigned slot = atomicAdd(p.cand_cnt + b, 1u);
                if (slot < CAP) p.cand[(long)b * CAP + slot] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
            }
        }
    }
    if (LEVEL < 3) {
        __syncthreads();
        for (int i = threadIdx.x; i < NB; i += 256)
            if (lh[i]) atomicAdd(&hist[LEVEL * NB + i], lh[i]);
    }
}

// Tie-safe compaction (the reference's argsort, polyline_fpn_vit_vertex_2.py:655-660, never fails on tied scores; the build's rule is
// "ties -> lower flat index"): when MORE pixels carry exactly the threshold key T than are still wanted (flat or
// saturated endpoint maps), the wanted ones are the lowest-indexed.  One workgroup per tile scans the crop in index order, ranks the
// tied pixels with a block-wide prefix sum and stops as soon as enough are taken.  Exits at once in the common all-ties-wanted case.
__global__ __launch_bounds__(1024) void topk_ties_kernel(TopkParams p) {
    __shared__ unsigned sh[4];
    __shared__ unsigned wave_tot[16];
    const int b = blockIdx.x;
    const unsigned* hist = p.hist + (long)b * 3 * NB;
    __shared__ unsigned pb_tmp[258];
    unsigned need = (unsigned)p.K, bin = 0, T = 0;
#pragma unroll
    for (int lv = 0; lv < 3; ++lv) {
        pick_bin_block(hist + lv * NB, need, bin, pb_tmp);
        T |= bin << (20 - 10 * lv);
    }
    if (threadIdx.x == 0) sh[3] = p.cand_cnt[b];                // candidates with key > T (pass 3 is complete)
    __syncthreads();
    const unsigned base = sh[3];
    if (hist[2 * NB + bin] == need || need == 0) return;
    const int Wc = p.W - 2 * p.clip;
    const long n = (long)(p.H - 2 * p.clip) * Wc;
    const float* lp = p.logit + (long)b * p.H * p.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned taken = 0;
    for (long i0 = 0; i0 < n && taken < need; i0 += 1024) {
        const long i = i0 + threadIdx.x;
        bool tie = false;
        if (i < n) {
            const int y = (int)lm_fastdiv((unsigned)i, p.div_wc), x = (int)((unsigned)i - (unsigned)y * (unsigned)Wc);
            tie = score_key(lp[(long)(y + p.clip) * p.W + x + p.clip]) == T;
        }
        const unsigned long long m = __ballot(tie);
        const unsigned before = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned off = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) off += wave_tot[w];
            tot += wave_tot[w];
        }
        const unsigned rank = taken + off + before;
        if (tie && rank < need) p.cand[(long)b * CAP + base + rank] = ((unsigned long long)T << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        taken += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) p.cand_cnt[b] = base + need;
}

// one workgroup per tile: bitonic sort (descending) of CAP 64-bit composites in LDS, emit top K
__global__ __launch_bounds__(1024) void topk_sort_kernel(TopkParams p, int* __restrict__ out_idx, float* __restrict__ out_score,
                                                         int* __restrict__ out_status) {
    __shared__ unsigned long long v[CAP];
    const int b = blockIdx.x;
    const unsigned cnt = p.cand_cnt[b];
    const unsigned n = cnt < (unsigned)CAP ? cnt : (unsigned)CAP;
    for (int i = threadIdx.x; i < CAP; i += 1024) v[i] = (unsigned)i < n ? p.cand[(long)b * CAP + i] : 0ull;
    __syncthreads();
    for (int k = 2; k <= CAP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < CAP; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = v[i], c = v[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < c) : (a > c)) {
                        v[i] = c;
                        v[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < p.K; i += 1024) {
        const unsigned long long e = v[i];
        const bool ok = (unsigned)i < n;
        out_idx[(long)b * p.K + i] = ok ? (int)(0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFu)) : -1;
        out_score[(long)b * p.K + i] = ok ? __uint_as_float((unsigned)(e >> 32)) : 0.f;
    }
    if (threadIdx.x == 0) out_status[b] = (cnt > (unsigned)CAP) ? 1 : 0;   // cannot happen since the tie-safe compaction (<= K candidates); kept as a guard
}


// ---- read-back packing: the decode outputs the host consumes (prop_conf, v_ext, cls_offset, rows, idx, status) gathered into ONE block, so a
// batch costs one device-to-host copy instead of six.  Segment s = bytes[s] (a multiple of 4) from src[s] to dst + offs[s] (256-byte aligned).
struct PackArgs {
    const void* src[LM_PACK_MAX_SEGMENTS];
    long offs[LM_PACK_MAX_SEGMENTS];
    long words[LM_PACK_MAX_SEGMENTS];       // 4-byte words
    long first_chunk[LM_PACK_MAX_SEGMENTS + 1];   // prefix sum of 16-byte chunks (the last chunk of a segment may be partial)
    int n;
};

__global__ __launch_bounds__(256) void pack_segments_kernel(PackArgs a, unsigned char* __restrict__ dst) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.first_chunk[a.n]) return;
    int s = 0;
#pragma unroll
    for (int k = 1; k < LM_PACK_MAX_SEGMENTS; ++k) s += (k < a.n && c >= a.first_chunk[k]) ? 1 : 0;
    const long w0 = (c - a.first_chunk[s]) * 4;
    const unsigned* src = static_cast<const unsigned*>(a.src[s]) + w0;
    unsigned* out = reinterpret_cast<unsigned*>(dst + a.offs[s]) + w0;
    const long left = a.words[s] - w0;
    if (left >= 4 && ((reinterpret_cast<unsigned long long>(src) & 15ull) == 0)) {
        *reinterpret_cast<uint4*>(out) = *reinterpret_cast<const uint4*>(src);
    } else {
        for (int e = 0; e < 4 && e < left; ++e) out[e] = src[e];
    }
}

}  // namespace

LM_API int lm_decode_proposals(void* stream, const float* pconf, const float* ext2, const float* cls2, const float* off2,
                               float* prop_conf, float* v_ext, float* cls_conf, int* cls_idx, double* cls_offset,
                               int B, int P, int R, float exist_thre, int prop_width, int half_buff) {
    LM_REQUIRE(pconf && ext2 && cls2 && off2 && prop_conf && v_ext && cls_conf && cls_idx && cls_offset, "decode_proposals: null pointer");
    const long total = (long)B * P * R;
    hipLaunchKernelGGL(decode_proposals_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, pconf, ext2, cls2,
                       off2, prop_conf, v_ext, cls_conf, cls_idx, cls_offset, P, R, exist_thre, prop_width, half_buff, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_decode_orient(void* stream, const float* x_nhwc, int ldx, int C, unsigned char* y, long pixels) {
    LM_REQUIRE(x_nhwc && y && C >= 1, "decode_orient: bad args");
    hipLaunchKernelGGL(decode_orient_kernel, dim3(lm_cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream, x_nhwc, ldx, C, y, pixels);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_decode_semantic(void* stream, const float* logit_chw3, unsigned char* sem, float* biseg, float* rows,
                              int B, int H, int W, float thre, int raw_mode) {
    LM_REQUIRE(logit_chw3 && sem, "decode_semantic: null pointer");
    LM_REQUIRE(!rows || H % 8 == 0, "decode_semantic: H must be a multiple of 8 for the row gather");
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(decode_semantic_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, logit_chw3, sem,
                       biseg, rows, H, W, thre, raw_mode, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API long lm_endp_topk_workspace_bytes(int B) {
    return (long)B * (3 * NB * sizeof(unsigned) + sizeof(unsigned) * 4 + (long)CAP * sizeof(unsigned long long));
}

LM_API int lm_endp_topk(void* stream, const float* endp_logit, void* workspace, int* out_idx, float* out_score,
                        int* out_status, int B, int H, int W, int clip, int K) {
    LM_REQUIRE(endp_logit && workspace && out_idx && out_score && out_status, "endp_topk: null pointer");
    LM_REQUIRE(K >= 1 && K <= CAP / 2 && H > 2 * clip && W > 2 * clip, "endp_topk: bad K=%d or crop", K);
    hipStream_t s = (hipStream_t)stream;
    TopkParams p;
    p.logit = endp_logit;
    char* ws = (char*)workspace;
    p.hist = (unsigned*)ws;
    p.cand_cnt = (unsigned*)(ws + (size_t)B * 3 * NB * sizeof(unsigned));
    p.cand = (unsigned long long*)(ws + (size_t)B * (3 * NB + 4) * sizeof(unsigned));
    p.H = H; p.W = W; p.clip = clip; p.K = K;
    LM_REQUIRE(W > 2 * clip && H > 2 * clip && (long)(H - 2 * clip) * (W - 2 * clip) < (1L << 31), "endp_topk: bad crop (H=%d W=%d clip=%d)", H, W, clip);
    p.div_wc = lm_fastdiv_make((unsigned)(W - 2 * clip));
    // (a kernel, not hipMemsetAsync: the call sits inside HIP-graph captures of the tile pipeline, and replays of a captured memset
    // node were observed to leave the histograms of the previous replay in place)
    const long zero_words = (long)B * (3 * NB + 4);
    hipLaunchKernelGGL(zero_u32_kernel, dim3(lm_cdiv(zero_words, 256)), dim3(256), 0, s, (unsigned*)workspace, zero_words);
    LM_LAUNCH_CHECK();
    dim3 grid(256, B);
    hipLaunchKernelGGL(topk_pass_kernel<0>, grid, dim3(256), 0, s, p);
    LM_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_pass_kernel<1>, grid, dim3(256), 0, s, p);
    LM_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_pass_kernel<2>, grid, dim3(256), 0, s, p);
    LM_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_pass_kernel<3>, grid, dim3(256), 0, s, p);
    LM_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_ties_kernel, dim3(B), dim3(1024), 0, s, p);
    LM_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_sort_kernel, dim3(B), dim3(1024), 0, s, p, out_idx, out_score, out_status);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_pack_segments(void* stream, int n, const void* const* src, const long* bytes, const long* dst_offsets, void* dst) {
    LM_REQUIRE(n >= 1 && n <= LM_PACK_MAX_SEGMENTS && src && bytes && dst_offsets && dst, "pack_segments: 1..%d segments", LM_PACK_MAX_SEGMENTS);
    PackArgs a;
    a.n = n;
    a.first_chunk[0] = 0;
    for (int s = 0; s < LM_PACK_MAX_SEGMENTS; ++s) {
        const bool on = s < n;
        LM_REQUIRE(!on || (src[s] && bytes[s] >= 0 && bytes[s] % 4 == 0 && dst_offsets[s] % 16 == 0), "pack_segments: segment %d: bytes %% 4, offset %% 16", s);
        a.src[s] = on ? src[s] : nullptr;
        a.offs[s] = on ? dst_offsets[s] : 0;
        a.words[s] = on ? bytes[s] / 4 : 0;
        a.first_chunk[s + 1] = a.first_chunk[s] + (a.words[s] + 3) / 4;
    }
    const long chunks = a.first_chunk[n];
    if (chunks == 0) return LM_OK;
    hipLaunchKernelGGL(pack_segments_kernel, dim3(lm_cdiv(chunks, 256)), dim3(256), 0, (hipStream_t)stream, a, static_cast<unsigned char*>(dst));
    LM_LAUNCH_CHECK();
    return LM_OK;
}

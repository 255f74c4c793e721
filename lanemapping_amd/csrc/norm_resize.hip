// Normalisation and resampling kernels (HBM-bound, NHWC fp32, float4 per lane).
//
//  lm_gn_stats                : per-(b,c) mean / rstd of GroupNorm(C groups == C channels)
//                               (postprojector.py:512-515), deterministic two-level fp64 reduction
//  lm_gn_relu_upsample        : y (=|+=) bilinear_align_corners(relu(gn(x)))  -> fuses the
//                               `_upsample(F.relu(gn(conv(..))))` and `s2+s3+s4` steps (postprojector.py:615-651)
//  lm_gn_relu_upsample_sum    : y = ((t0 + t1) + t2) of up to three such terms in one pass (the `s2 + s3 + s4` sum)
//  lm_upsample_bilinear_nhwc  : F.interpolate(mode='bilinear', align_corners=True) (+ optional add)
//                               (postprojector.py:541-561; heads/polyline_fpn_vit_vertex_2.py:298-300)
//  lm_upsample_bilinear_to_chw: same, NHWC source -> planar [B,C,Ho,Wo] destination (bi_seg / endp maps)
//  lm_layernorm_rows          : nn.LayerNorm(dim) eps 1e-5 over rows (vitsegnet.py:20-26)
//  lm_unpatchify              : 'b (h w) (p1 p2 c) -> b c (h p1) (w p2)' into NHWC (vitsegnet.py:180)
//
// Bilinear source index follows ATen: scale = (in-1)/(out-1) in fp32, src = scale*dst,
// i0 = floor(src), i1 = min(i0+1, in-1), w1 = src - i0, w0 = 1 - w1.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void bilin_axis(int o, int in, int out, int& i0, int& i1, float& w0, float& w1) {
    lm_bilin_axis(o, in, out, i0, i1, w0, w1);
}

// ----------------------------------------------------------------------------- GroupNorm stats
constexpr int GN_CHUNK = 512;   // pixels per partial block

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                         int HW, int C, int nchunk) {
    __shared__ double red[2][256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int lanes_p = 256 / C;                 // pixel lanes per block (C in {64,128,256})
    const int c = threadIdx.x % C, pl = threadIdx.x / C;
    const int p0 = chunk * GN_CHUNK;
    const int p1 = min(p0 + GN_CHUNK, HW);
    double s = 0.0, ss = 0.0;
    const float* xb = x + (long)b * HW * C;
    for (int p = p0 + pl; p < p1; p += lanes_p) {
        const double v = (double)xb[(long)p * C + c];
        s += v;
        ss += v * v;
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    if (pl == 0) {
        for (int q = 1; q < lanes_p; ++q) {       // fixed order
            s += red[0][q * C + c];
            ss += red[1][q * C + c];
        }
        double* o = part + (((long)b * nchunk + chunk) * C + c) * 2;
        o[0] = s;
        o[1] = ss;
    }
}

// grid (ceil(C/32), B), 256 threads = 8 chunk lanes x 32 channels; fixed summation order => deterministic
// `split` equal channel groups: stats come out as [split][B][C / split][2], so each group is a contiguous [B][C / split][2] block (the two
// semantic branches share one merged GEMM and take one half each)
__global__ __launch_bounds__(256) void gn_final_kernel(const double* __restrict__ part, float* __restrict__ stats, int HW, int C,
                                                       int nchunk, float eps, int split) {
    __shared__ double red[2][8][32];
    const int b = blockIdx.y;
    const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, ss = 0.0;
    if (c < C)
        for (int k = kl; k < nchunk; k += 8) {
            const double* o = part + (((long)b * nchunk + k) * C + c) * 2;
            s += o[0];
            ss += o[1];
        }
    red[0][kl][cl] = s;
    red[1][kl][cl] = ss;
    __syncthreads();
    if (kl == 0 && c < C) {
        for (int q = 1; q < 8; ++q) {
            s += red[0][q][cl];
            ss += red[1][q][cl];
        }
        const double mean = s / HW;
        double var = ss / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const int cg = C / split, g = c / cg;
        float* o = stats + (((long)g * gridDim.y + b) * cg + (c - g * cg)) * 2;
        o[0] = (float)mean;
        o[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ----------------------------------------------------------------------------- GN + ReLU + bilinear (+=)
__global__ __launch_bounds__(256) void gn_relu_upsample_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ y, int Hi, int Wi, int Ho, int Wo, int C,
                                                               int accumulate, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4n = C / 4;
    const int c = (int)(i % c4n) * 4;
    long t = i / c4n;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilin_axis(oy, Hi, Ho, y0, y1, wy0, wy1);
    bilin_axis(ox, Wi, Wo, x0, x1, wx0, wx1);
    f32x4 a, g;   // per-channel affine of the normalisation: v*a + g  (statistics / gamma / beta as 16-byte loads)
    {
        const f32x4* st = reinterpret_cast<const f32x4*>(stats + ((long)b * C + c) * 2);
        const f32x4 s0 = st[0], s1 = st[1], gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
        const float mean[4] = {s0[0], s0[2], s1[0], s1[2]}, rstd[4] = {s0[1], s0[3], s1[1], s1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ae, ge;
            lm_gn_affine(mean[e], rstd[e], gm[e], bt[e], ae, ge);
            a[e] = ae;
            g[e] = ge;
        }
    }
    const float* xb = x + (long)b * Hi * Wi * C + c;
    auto tap = [&](int yy, int xx) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((long)yy * Wi + xx) * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lm_gn_relu(v[e], a[e], g[e]);
        return v;
    };
    const f32x4 v00 = tap(y0, x0), v01 = tap(y0, x1), v10 = tap(y1, x0), v11 = tap(y1, x1);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = lm_bilerp(v00[e], v01[e], v10[e], v11[e], wy0, wy1, wx0, wx1);
    f32x4* yp = reinterpret_cast<f32x4*>(y + i * 4);
    if (accumulate) {
        const f32x4 prev = *yp;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = prev[e] + o[e];
    }
    *yp = o;
}

// ----------------------------------------------------------------------------- sum of up to 3 GN + ReLU + bilinear terms
// y = ((t0 + t1) + t2), t_k = bilinear(relu(gn_k(x_k))): the reference's `s2 + s3 + s4` (postprojector.py:621,647) in one pass
// instead of one write and two read-modify-writes of y.  Same per-term arithmetic and summation order as three calls of the
// kernel above.  32-bit index math (the launch checks the sizes).
struct GnTerm {
    const float* x;
    const float* stats;
    int Hi, Wi, ld;      // ld: floats between pixels of x (>= C: a term may be a channel slice of a wider tensor)
    float sy, sx;        // lm_bilin_axis's source scales (Hi - 1) / (Ho - 1), (Wi - 1) / (Wo - 1): the same IEEE quotients, taken on the host
};
struct GnSum {
    GnTerm t[3];
    int n;
};

// Optional fused 1x1 projection (feature_layer 128 -> 8, output_layer_endp 128 -> 1; postprojector.py:628-651): the C/4 lanes of a
// pixel sit in one wave (C/4 a power of two <= 64), each multiplies its 4 channels into `cout` partial sums, a butterfly over those lanes
// adds them up and the first lane stores y1[pixel][0..cout) (+ bias).  With y == nullptr the summed tensor itself is never written.
struct Proj1x1 {
    const float* w;      // [C][16] (pack_small layout), nullptr = no projection
    const float* bias;   // [cout] or nullptr
    float* y1;           // [B*Ho*Wo][ldy1]
    int cout, ldy1;
};

// grid (ceil(Wo * C/4 / 256), Ho, B): the output row and the image come from the block index, the column with one division by C/4 (a
// shift for the power-of-two channel counts of the FPN).
// N terms, SAME = bit k set when term k has the output's size (one tap, returned as it is).  Both are TEMPLATE parameters: with run-time
// term counts / sizes every term sat behind a branch, its loads could not be hoisted above the previous term's arithmetic, and a thread
// paid up to three serial memory round trips - the counters showed 73 % of the wave cycles parked at s_waitcnt with 8 waves per SIMD
// (tools/prof_gn_pmc.sh).  Now the statistics and every tap of every term are loaded first (up to 6 + 6 16-byte loads in flight per
// lane), then the arithmetic runs in the old order: same bits.
// 16 bytes at element `elem` of a wave-uniform base: byte offset kept in 32 bits so that the access becomes scalar base + vector offset
__device__ __forceinline__ f32x4 ld_quad(const float* base, unsigned elem) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (elem << 2));
}

template <int N, int SAME>
__global__ __launch_bounds__(256) void gn_relu_upsample_sum_kernel(GnSum P, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ y, int Ho, int Wo, int C, int c4shift, Proj1x1 Q) {
    __shared__ __attribute__((aligned(16))) float wl[256 * 8];   // projection weights [C][8] (cout padded with zeros)
    const unsigned c4n = (unsigned)C / 4;
    const unsigned j0 = blockIdx.x * 256u + threadIdx.x;      // (column, channel quad) inside the output row
    const bool live = j0 < (unsigned)Wo * c4n;                // (a multiple of C/4: the lanes of a pixel are live together)
    const unsigned j = live ? j0 : 0u;                        // (dead threads of the row's last workgroup load pixel 0 and store nothing)
    const int ox = c4shift >= 0 ? (int)(j >> c4shift) : (int)(j / c4n);
    const int c = (int)(j - (unsigned)ox * c4n) * 4;
    const int oy = (int)blockIdx.y, b = (int)blockIdx.z;
    const unsigned i = ((unsigned)(b * Ho + oy) * (unsigned)Wo) * c4n + j;      // flat output quad
    // ---- phase 1: every load of the thread
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 st0[N], st1[N], tap[N][4];
    float wy0[N], wy1[N], wx0[N], wx1[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const GnTerm& T = P.t[k];
        // (wave-uniform 64-bit bases + 32-bit per-lane element offsets: scalar base / vector offset addressing instead of a 64-bit
        // multiply-add chain per tap; an image of a term stays below 2^31 elements - checked by the launcher)
        const float* stb = T.stats + (long)b * C * 2;                 // (mean, rstd) x 4 channels
        st0[k] = ld_quad(stb, (unsigned)(c * 2));
        st1[k] = ld_quad(stb, (unsigned)(c * 2 + 4));
        const float* xb = T.x + (long)b * T.Hi * T.Wi * T.ld;
        if ((SAME >> k) & 1) {
            tap[k][0] = ld_quad(xb, (unsigned)((oy * T.Wi + ox) * T.ld + c));
        } else {
            int y0, y1, x0, x1;
            lm_bilin_axis_scaled(oy, T.Hi, T.sy, y0, y1, wy0[k], wy1[k]);
            lm_bilin_axis_scaled(ox, T.Wi, T.sx, x0, x1, wx0[k], wx1[k]);
            tap[k][0] = ld_quad(xb, (unsigned)((y0 * T.Wi + x0) * T.ld + c));
            tap[k][1] = ld_quad(xb, (unsigned)((y0 * T.Wi + x1) * T.ld + c));
            tap[k][2] = ld_quad(xb, (unsigned)((y1 * T.Wi + x0) * T.ld + c));
            tap[k][3] = ld_quad(xb, (unsigned)((y1 * T.Wi + x1) * T.ld + c));
        }
    }
    // the projection weights are staged BEHIND the taps' loads: in front of them (rounds 2-3) every workgroup paid two memory round
    // trips in a row - weights, barrier, taps
    if (Q.w) {
        for (int k = threadIdx.x; k < C * 8; k += 256) wl[k] = (k & 7) < Q.cout ? Q.w[(k >> 3) * 16 + (k & 7)] : 0.f;
        __syncthreads();
    }
    if (!live) return;
    // ---- phase 2: arithmetic, term by term
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < N; ++k) {
        f32x4 a, g;
        const float mean[4] = {st0[k][0], st0[k][2], st1[k][0], st1[k][2]}, rstd[4] = {st0[k][1], st0[k][3], st1[k][1], st1[k][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ae, ge;
            lm_gn_affine(mean[e], rstd[e], gm[e], bt[e], ae, ge);
            a[e] = ae;
            g[e] = ge;
        }
        f32x4 o;
        if ((SAME >> k) & 1) {              // same size: the blend has weights (1, 0) and returns the tap itself, bit for bit
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = lm_gn_relu(tap[k][0][e], a[e], g[e]);
        } else {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q][e] = lm_gn_relu(tap[k][q][e], a[e], g[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = lm_bilerp(v[0][e], v[1][e], v[2][e], v[3][e], wy0[k], wy1[k], wx0[k], wx1[k]);
        }
        if (k == 0) {
            acc = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] + o[e];
        }
    }
    if (y) *reinterpret_cast<f32x4*>(y + (long)i * 4) = acc;
    if (Q.w) {
        // partial sums of this lane's 4 channels for the 8 (padded) outputs, weights from the LDS copy [C][8]
        float part[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) part[n] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (c + e) * 8), w1 = *reinterpret_cast<const f32x4*>(wl + (c + e) * 8 + 4);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                part[n] = fmaf(acc[e], w0[n], part[n]);
                part[4 + n] = fmaf(acc[e], w1[n], part[4 + n]);
            }
        }
        // transpose-reduce over the C/4 lanes of the pixel: each of the first three steps halves the outputs a lane keeps while doubling
        // the lanes summed (4 + 2 + 1 shuffles), then plain butterflies: 9-10 shuffles instead of 8 x log2(C/4)
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int half = 4, mask = 1; half >= 1; half >>= 1, mask <<= 1) {
            const bool upper = (lane & mask) != 0;
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                // (two plain selects: left to itself the compiler turns them into `part[upper ? jj : jj + half]`, a run-time index it
                // resolves with a compare + select per array element - ~200 of the kernel's 500 VALU instructions, profiles/README.md)
                float lo = part[jj], hi = part[jj + half];
                asm volatile("" : "+v"(lo), "+v"(hi));
                const float send = upper ? lo : hi, keep = upper ? hi : lo;
                part[jj] = keep + __shfl_xor(send, mask);
            }
        }
        for (unsigned o = 8; o < c4n; o <<= 1) part[0] += __shfl_xor(part[0], (int)o);
        const unsigned li = (unsigned)c >> 2;
        const int n = 4 * (int)(li & 1) + 2 * (int)((li >> 1) & 1) + (int)((li >> 2) & 1);
        if (li < 8 && n < Q.cout) Q.y1[(long)((b * Ho + oy) * Wo + ox) * Q.ldy1 + n] = part[0] + (Q.bias ? Q.bias[n] : 0.f);
    }
}

// ----------------------------------------------------------------------------- one GN + ReLU + bilinear term, source staged in LDS
// The one-term up-sampling call (s4 of the semantic branches: 256 channels, 144^2 -> 288^2, 0.68 GB written per 8 tiles) ran at 2.7 TB/s
// in the kernel above against 5.8 TB/s for the same-size case: four 16-byte taps per output quad from the vector cache, each normalised
// again by every output that touches it.  Here a workgroup owns UT_R x UT_C output pixels x all channels: the source block it needs
// (<= UT_SR x UT_SC pixels for scales <= 1/2 + eps, checked by the launcher) is read ONCE, normalised + ReLU'd once per element and kept
// in LDS; the outputs then blend four ds_read_b128 taps.  Same per-element arithmetic (lm_gn_relu, then lm_bilerp with lm_bilin_axis
// weights) => same bits as gn_relu_upsample_kernel / gn_relu_upsample_sum_kernel<1, 0>.
constexpr int UT_R = 8, UT_C = 16;

__global__ __launch_bounds__(256) void gn_relu_up_lds_kernel(GnTerm T, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, int Ho, int Wo, int C, int SR, int SC) {
    extern __shared__ __attribute__((aligned(16))) float ups[];       // [SR * SC][C] normalised source block
    __shared__ int ty0[UT_R], ty1[UT_R], tx0[UT_C], tx1[UT_C];        // taps (relative to the block origin) and weights of the tile's rows / columns
    __shared__ float twy0[UT_R], twy1[UT_R], twx0[UT_C], twx1[UT_C];
    const int tid = threadIdx.x, c4n = C / 4;
    const int oy0 = blockIdx.y * UT_R, ox0 = blockIdx.x * UT_C, b = blockIdx.z;
    const int ny = min(UT_R, Ho - oy0), nx = min(UT_C, Wo - ox0);
    int sy0, sx0, i1;
    float w0, w1;
    lm_bilin_axis_scaled(oy0, T.Hi, T.sy, sy0, i1, w0, w1);           // block origin = first tap of the first row / column
    lm_bilin_axis_scaled(ox0, T.Wi, T.sx, sx0, i1, w0, w1);
    if (tid < UT_R) {
        int a0, a1;
        lm_bilin_axis_scaled(min(oy0 + tid, Ho - 1), T.Hi, T.sy, a0, a1, w0, w1);
        ty0[tid] = a0 - sy0; ty1[tid] = a1 - sy0; twy0[tid] = w0; twy1[tid] = w1;
    } else if (tid >= 64 && tid < 64 + UT_C) {
        const int t = tid - 64;
        int a0, a1;
        lm_bilin_axis_scaled(min(ox0 + t, Wo - 1), T.Wi, T.sx, a0, a1, w0, w1);
        tx0[t] = a0 - sx0; tx1[t] = a1 - sx0; twx0[t] = w0; twx1[t] = w1;
    }
    // ---- phase 1: source block -> LDS.  256 % (C/4) == 0: a thread keeps one channel quad and walks the pixels
    const int cq = tid % c4n, pstep = 256 / c4n, p0 = tid / c4n;
    const int c = cq * 4;
    f32x4 a, g;
    {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
        const f32x4* st = reinterpret_cast<const f32x4*>(T.stats + ((long)b * C + c) * 2);      // (mean, rstd) x 4 channels
        const f32x4 st0 = st[0], st1 = st[1];
        const float mean[4] = {st0[0], st0[2], st1[0], st1[2]}, rstd[4] = {st0[1], st0[3], st1[1], st1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ae, ge;
            lm_gn_affine(mean[e], rstd[e], gm[e], bt[e], ae, ge);
            a[e] = ae;
            g[e] = ge;
        }
    }
    const int nsrc = SR * SC;
    const float* xb = T.x + (long)b * T.Hi * T.Wi * T.ld + c;
    constexpr int UNR = 4;                                             // loads in flight per thread and round
    for (int pb = p0; pb < nsrc; pb += pstep * UNR) {
        f32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int pp = pb + u * pstep;
            const int r = pp / SC, q = pp - r * SC;
            const int yy = min(sy0 + r, T.Hi - 1), xx = min(sx0 + q, T.Wi - 1);     // (clamped: rows / columns past the image are never blended)
            if (pp < nsrc) v[u] = *reinterpret_cast<const f32x4*>(xb + ((long)yy * T.Wi + xx) * T.ld);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int pp = pb + u * pstep;
            if (pp < nsrc) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = lm_gn_relu(v[u][e], a[e], g[e]);
                *reinterpret_cast<f32x4*>(ups + (long)pp * C + c) = o;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: outputs
    const long obase = ((long)b * Ho + oy0) * Wo + ox0;
    for (int pp = p0; pp < UT_R * UT_C; pp += pstep) {
        const int r = pp / UT_C, q = pp % UT_C;
        if (r >= ny || q >= nx) continue;
        const float* s0 = ups + (long)(ty0[r] * SC) * C + c;
        const float* s1 = ups + (long)(ty1[r] * SC) * C + c;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(s0 + tx0[q] * C), v01 = *reinterpret_cast<const f32x4*>(s0 + tx1[q] * C);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(s1 + tx0[q] * C), v11 = *reinterpret_cast<const f32x4*>(s1 + tx1[q] * C);
        const float wy0 = twy0[r], wy1 = twy1[r], wx0 = twx0[q], wx1 = twx1[q];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = lm_bilerp(v00[e], v01[e], v10[e], v11[e], wy0, wy1, wx0, wx1);
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y + (obase + (long)r * Wo + q) * C + c));
    }
}

// ----------------------------------------------------------------------------- s2 + s3 + s4 with the up-sampled term staged in LDS (round 5)
// The three-term call of the semantic branches - two same-size terms (s2, s4) and ONE up-sampled term (s3: 144^2 -> 288^2) followed by the
// branch's 1x1 output layer - ran at 2.9 TB/s in gn_relu_upsample_sum_kernel<3, 5>: per output quad six 16-byte loads, three
// (mean, rstd) pairs, three GroupNorm affines, the blend and the projection weights from LDS.  Here a workgroup owns UT_R x UT_C output
// pixels x all channels, a thread keeps ONE channel quad for all its pixels: the affines of the three terms and its 4 x 8 projection
// weights are made once per thread, the up-sampled term's source block is normalised once per element and staged in LDS (as in
// gn_relu_up_lds_kernel), and an output quad costs two 16-byte global loads + four ds_read_b128.  Same per-element arithmetic in the
// same order (lm_gn_relu per tap, lm_bilerp, (t0 + t1) + t2, the fmaf chain and the shuffle tree of the projection): SAME BITS as
// gn_relu_upsample_sum_kernel<3, 5> (test_gn_sum_three_terms_lds_bit_identical).  C/4 lanes of a pixel sit in one wave (C/4 | 64).
__global__ __launch_bounds__(256) void gn_sum3_lds_kernel(GnSum P, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ y, int Ho, int Wo, int C, int SR, int SC, Proj1x1 Q) {
    extern __shared__ __attribute__((aligned(16))) float ups[];       // [SR * SC][C] normalised + ReLU'd source block of term 1
    __shared__ int ty0[UT_R], ty1[UT_R], tx0[UT_C], tx1[UT_C];
    __shared__ float twy0[UT_R], twy1[UT_R], twx0[UT_C], twx1[UT_C];
    const int tid = threadIdx.x, c4n = C / 4;
    const int oy0 = blockIdx.y * UT_R, ox0 = blockIdx.x * UT_C, b = blockIdx.z;
    const int ny = min(UT_R, Ho - oy0), nx = min(UT_C, Wo - ox0);
    const GnTerm& T0 = P.t[0];
    const GnTerm& T1 = P.t[1];
    const GnTerm& T2 = P.t[2];
    int sy0, sx0, i1;
    float w0, w1;
    lm_bilin_axis_scaled(oy0, T1.Hi, T1.sy, sy0, i1, w0, w1);
    lm_bilin_axis_scaled(ox0, T1.Wi, T1.sx, sx0, i1, w0, w1);
    if (tid < UT_R) {
        int a0, a1;
        lm_bilin_axis_scaled(min(oy0 + tid, Ho - 1), T1.Hi, T1.sy, a0, a1, w0, w1);
        ty0[tid] = a0 - sy0; ty1[tid] = a1 - sy0; twy0[tid] = w0; twy1[tid] = w1;
    } else if (tid >= 64 && tid < 64 + UT_C) {
        const int t = tid - 64;
        int a0, a1;
        lm_bilin_axis_scaled(min(ox0 + t, Wo - 1), T1.Wi, T1.sx, a0, a1, w0, w1);
        tx0[t] = a0 - sx0; tx1[t] = a1 - sx0; twx0[t] = w0; twx1[t] = w1;
    }
    const int cq = tid % c4n, pstep = 256 / c4n, p0 = tid / c4n;
    const int c = cq * 4;
    // the three affines of this thread's channel quad
    f32x4 a[3], g[3];
    {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f32x4* st = reinterpret_cast<const f32x4*>(P.t[k].stats + ((long)b * C + c) * 2);      // (mean, rstd) x 4 channels
            const f32x4 st0 = st[0], st1 = st[1];
            const float mean[4] = {st0[0], st0[2], st1[0], st1[2]}, rstd[4] = {st0[1], st0[3], st1[1], st1[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ae, ge;
                lm_gn_affine(mean[e], rstd[e], gm[e], bt[e], ae, ge);
                a[k][e] = ae;
                g[k][e] = ge;
            }
        }
    }
    // projection weights of this thread's four channels, [e][8] (cout padded with zeros, like the LDS copy of the per-output kernel)
    float pw[4][8];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < 8; ++n) pw[e][n] = n < Q.cout ? Q.w[(c + e) * 16 + n] : 0.f;
    // ---- phase 1: source block of the up-sampled term -> LDS
    const int nsrc = SR * SC;
    {
        const float* xb = T1.x + (long)b * T1.Hi * T1.Wi * T1.ld + c;
        constexpr int UNR = 4;
        for (int pb = p0; pb < nsrc; pb += pstep * UNR) {
            f32x4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int pp = pb + u * pstep;
                const int r = pp / SC, q = pp - r * SC;
                const int yy = min(sy0 + r, T1.Hi - 1), xx = min(sx0 + q, T1.Wi - 1);
                if (pp < nsrc) v[u] = *reinterpret_cast<const f32x4*>(xb + ((long)yy * T1.Wi + xx) * T1.ld);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int pp = pb + u * pstep;
                if (pp < nsrc) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = lm_gn_relu(v[u][e], a[1][e], g[1][e]);
                    *reinterpret_cast<f32x4*>(ups + (long)pp * C + c) = o;
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: outputs, UNR2 pixels of this thread in flight
    const float* x0b = T0.x + (long)b * T0.Hi * T0.Wi * T0.ld + c;
    const float* x2b = T2.x + (long)b * T2.Hi * T2.Wi * T2.ld + c;
    const int lane = tid & 63;
    const unsigned li = (unsigned)cq;
    const int nsel = 4 * (int)(li & 1) + 2 * (int)((li >> 1) & 1) + (int)((li >> 2) & 1);
    const float pbias = (li < 8 && nsel < Q.cout && Q.bias) ? Q.bias[nsel] : 0.f;
    constexpr int UNR2 = 4;
    for (int pb = p0; pb < UT_R * UT_C; pb += pstep * UNR2) {
        f32x4 t0[UNR2], t2[UNR2];
        bool live[UNR2];
        int rr[UNR2], qq[UNR2];
#pragma unroll
        for (int u = 0; u < UNR2; ++u) {
            const int pp = pb + u * pstep;
            rr[u] = pp / UT_C; qq[u] = pp % UT_C;
            live[u] = rr[u] < ny && qq[u] < nx;
            const int oy = min(oy0 + rr[u], Ho - 1), ox = min(ox0 + qq[u], Wo - 1);       // (dead pixels load a valid one and store nothing)
            t0[u] = *reinterpret_cast<const f32x4*>(x0b + ((long)oy * T0.Wi + ox) * T0.ld);
            t2[u] = *reinterpret_cast<const f32x4*>(x2b + ((long)oy * T2.Wi + ox) * T2.ld);
        }
#pragma unroll
        for (int u = 0; u < UNR2; ++u) {
            const int r = min(rr[u], UT_R - 1), q = qq[u];
            const float* s0 = ups + (long)(ty0[r] * SC) * C + c;
            const float* s1 = ups + (long)(ty1[r] * SC) * C + c;
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(s0 + tx0[q] * C), v01 = *reinterpret_cast<const f32x4*>(s0 + tx1[q] * C);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(s1 + tx0[q] * C), v11 = *reinterpret_cast<const f32x4*>(s1 + tx1[q] * C);
            const float wy0 = twy0[r], wy1 = twy1[r], wx0 = twx0[q], wx1 = twx1[q];
            f32x4 acc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
                const float o0 = lm_gn_relu(t0[u][e], a[0][e], g[0][e]);
                const float o1 = lm_bilerp(v00[e], v01[e], v10[e], v11[e], wy0, wy1, wx0, wx1);
                const float o2 = lm_gn_relu(t2[u][e], a[2][e], g[2][e]);
                acc[e] = (o0 + o1) + o2;
            }
            const long opix = ((long)b * Ho + oy0 + rr[u]) * Wo + ox0 + qq[u];
            if (y && live[u]) *reinterpret_cast<f32x4*>(y + opix * C + c) = acc;
            // the 1x1 projection: the per-output kernel's fmaf chain and transpose-reduce, lane for lane
            float part[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) part[n] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int n = 0; n < 8; ++n) part[n] = fmaf(acc[e], pw[e][n], part[n]);
#pragma unroll
            for (int half = 4, mask = 1; half >= 1; half >>= 1, mask <<= 1) {
                const bool upper = (lane & mask) != 0;
#pragma unroll
                for (int jj = 0; jj < half; ++jj) {
                    float lo = part[jj], hi = part[jj + half];
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    const float send = upper ? lo : hi, keep = upper ? hi : lo;
                    part[jj] = keep + __shfl_xor(send, mask);
                }
            }
            for (unsigned o = 8; o < (unsigned)c4n; o <<= 1) part[0] += __shfl_xor(part[0], (int)o);
            if (live[u] && li < 8 && nsel < Q.cout) Q.y1[opix * Q.ldy1 + nsel] = part[0] + pbias;
        }
    }
}

// ----------------------------------------------------------------------------- plain bilinear, NHWC -> NHWC slice
__global__ __launch_bounds__(256) void upsample_nhwc_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ add, int lda,
                                                            float* __restrict__ y, int ldy, int Hi, int Wi, int Ho, int Wo,
                                                            int C, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4n = C / 4;
    const int c = (int)(i % c4n) * 4;
    long t = i / c4n;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilin_axis(oy, Hi, Ho, y0, y1, wy0, wy1);
    bilin_axis(ox, Wi, Wo, x0, x1, wx0, wx1);
    const float* xb = x + (long)b * Hi * Wi * ldx + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(xb + ((long)y0 * Wi + x0) * ldx);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(xb + ((long)y0 * Wi + x1) * ldx);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(xb + ((long)y1 * Wi + x0) * ldx);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(xb + ((long)y1 * Wi + x1) * ldx);
    const long opix = ((long)b * Ho + oy) * Wo + ox;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = lm_bilerp(v00[e], v01[e], v10[e], v11[e], wy0, wy1, wx0, wx1);
    if (add) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(add + opix * lda + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] + r[e];
    }
    *reinterpret_cast<f32x4*>(y + opix * ldy + c) = o;
}

// NHWC (few channels) -> planar [B,C,Ho,Wo]
__device__ __forceinline__ float upsample_blend(float v00, float v01, float v10, float v11, float wy0, float wy1, float wx0, float wx1) {
    return lm_bilerp(v00, v01, v10, v11, wy0, wy1, wx0, wx1);      // fixed-order blend (common.h): scalar and vector kernels agree bit for bit
}

__global__ __launch_bounds__(256) void upsample_to_chw_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                              int Hi, int Wi, int Ho, int Wo, int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wo);
    long t = i / Wo;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilin_axis(oy, Hi, Ho, y0, y1, wy0, wy1);
    bilin_axis(ox, Wi, Wo, x0, x1, wx0, wx1);
    const float* xb = x + (long)b * Hi * Wi * ldx + c;
    const float v00 = xb[((long)y0 * Wi + x0) * ldx], v01 = xb[((long)y0 * Wi + x1) * ldx];
    const float v10 = xb[((long)y1 * Wi + x0) * ldx], v11 = xb[((long)y1 * Wi + x1) * ldx];
    y[i] = upsample_blend(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
}

// four horizontally adjacent outputs per thread (Wo % 4 == 0): one row decode and one y-axis interpolation per 16-byte store, 32-bit
// index math.  Same blend as the scalar kernel (both written as the plain expression in this file).
__global__ __launch_bounds__(256) void upsample_to_chw4_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                               int Hi, int Wi, int Ho, int Wo, int C, unsigned total4) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const unsigned wq = (unsigned)Wo / 4;
    const int ox0 = (int)(i % wq) * 4;
    unsigned t = i / wq;
    const int oy = (int)(t % (unsigned)Ho);
    t /= (unsigned)Ho;
    const int c = (int)(t % (unsigned)C);
    const int b = (int)(t / (unsigned)C);
    int y0, y1;
    float wy0, wy1;
    bilin_axis(oy, Hi, Ho, y0, y1, wy0, wy1);
    const float* xb = x + (long)b * Hi * Wi * ldx + c;
    const float* r0 = xb + (long)y0 * Wi * ldx;
    const float* r1 = xb + (long)y1 * Wi * ldx;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int x0, x1;
        float wx0, wx1;
        bilin_axis(ox0 + k, Wi, Wo, x0, x1, wx0, wx1);
        const float v00 = r0[(long)x0 * ldx], v01 = r0[(long)x1 * ldx], v10 = r1[(long)x0 * ldx], v11 = r1[(long)x1 * ldx];
        o[k] = upsample_blend(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
    }
    *reinterpret_cast<f32x4*>(y + (long)i * 4) = o;
}

// ----------------------------------------------------------------------------- LayerNorm: one wave per row
template <int PER>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, long rows,
                                                        float eps) {
    constexpr int D = PER * 64;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * D;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v[k] = xr[k * 64 + lane];
        s += v[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const float d = v[k] - mean;
        q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    float* yr = y + row * D;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = k * 64 + lane;
        yr[c] = (v[k] - mean) * rstd * gamma[c] + beta[c];
    }
}

__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ t, float* __restrict__ y, int G, int P,
                                                         int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // index into NHWC output [B, G*P, G*P, C]
    if (i >= total) return;
    const int c = (int)(i % C);
    long r = i / C;
    const int xx = (int)(r % (G * P));
    r /= (G * P);
    const int yy = (int)(r % (G * P));
    const int b = (int)(r / (G * P));
    const int gh = yy / P, p1 = yy % P, gw = xx / P, p2 = xx % P;
    y[i] = t[((long)b * G * G + gh * G + gw) * (P * P * C) + (p1 * P + p2) * C + c];
}

}  // namespace

LM_API int lm_gn_stats(void* stream, const float* x, double* workspace, float* stats, int B, int HW, int C, float eps) {
    LM_REQUIRE(x && workspace && stats, "gn_stats: null pointer");
    LM_REQUIRE(C <= 256 && 256 % C == 0, "gn_stats: C=%d must divide 256", C);
    const int nchunk = lm_cdiv(HW, GN_CHUNK);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, workspace, HW, C, nchunk);
    LM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_final_kernel, dim3(lm_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, workspace, stats, HW, C, nchunk, eps, 1);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// second pass for partials produced by lm_conv2d_nhwc_mfma_f32_gnstats ([B][nchunk][C][2] doubles)
LM_API int lm_gn_finalize(void* stream, const double* partial, float* stats, int B, int HW, int C, int nchunk, float eps) {
    LM_REQUIRE(partial && stats && nchunk >= 1, "gn_finalize: bad args");
    hipLaunchKernelGGL(gn_final_kernel, dim3(lm_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, partial, stats, HW, C, nchunk, eps, 1);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// the same with the statistics laid out per channel group: stats [split][B][C / split][2] (C % split == 0) - same values
LM_API int lm_gn_finalize_split(void* stream, const double* partial, float* stats, int B, int HW, int C, int nchunk, float eps, int split) {
    LM_REQUIRE(partial && stats && nchunk >= 1 && split >= 1 && C % split == 0, "gn_finalize_split: bad args (C=%d split=%d)", C, split);
    hipLaunchKernelGGL(gn_final_kernel, dim3(lm_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, partial, stats, HW, C, nchunk, eps, split);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API long lm_gn_stats_workspace_bytes(int B, int HW, int C) {
    return (long)B * lm_cdiv(HW, GN_CHUNK) * C * 2 * (long)sizeof(double);
}

LM_API int lm_gn_relu_upsample(void* stream, const float* x, const float* stats, const float* gamma, const float* beta,
                               float* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int accumulate) {
    LM_REQUIRE(x && stats && gamma && beta && y && C % 4 == 0, "gn_relu_upsample: bad args");
    const long total4 = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(gn_relu_upsample_kernel, dim3(lm_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, stats, gamma, beta, y, Hi, Wi, Ho, Wo, C, accumulate, total4);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

namespace {
// Source rows (columns) the tiles of `tile` output rows need at most: last second tap - first first tap + 1, from the SAME float
// arithmetic as lm_bilin_axis (IEEE single precision on both sides, no contraction), so the kernel's LDS block is exactly large enough.
int up_block_extent(int in, int out, int tile) {
#pragma clang fp contract(off)
    const float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    auto tap0 = [&](int o) {
        int i0 = (int)(scale * (float)o);
        return i0 > in - 1 ? in - 1 : i0;
    };
    int ext = 1;
    for (int o0 = 0; o0 < out; o0 += tile) {
        const int ol = (o0 + tile - 1 < out ? o0 + tile - 1 : out - 1);
        const int last0 = tap0(ol), last1 = last0 + (last0 < in - 1 ? 1 : 0);
        const int e = last1 - tap0(o0) + 1;
        if (e > ext) ext = e;
    }
    return ext;
}

int launch_gn_sum(void* stream, int n, const float* const* x, const float* const* stats, const int* Hi, const int* Wi, const int* ldx,
                  const float* gamma, const float* beta, float* y, int B, int Ho, int Wo, int C, Proj1x1 Q) {
    LM_REQUIRE(n >= 1 && n <= 3 && x && stats && Hi && Wi && gamma && beta && C > 0 && C % 4 == 0, "gn_relu_upsample_sum: bad args");
    const long total4 = (long)B * Ho * Wo * (C / 4);
    LM_REQUIRE(total4 > 0 && total4 < (1L << 31), "gn_relu_upsample_sum: %ld output quads do not fit 32-bit indices", total4);
    GnSum P;
    P.n = n;
    for (int k = 0; k < 3; ++k) {
        const int q = k < n ? k : 0;
        LM_REQUIRE(x[q] && stats[q] && Hi[q] > 0 && Wi[q] > 0, "gn_relu_upsample_sum: bad term %d", q);
        const int ld = ldx ? ldx[q] : C;
        LM_REQUIRE(ld >= C && ld % 4 == 0, "gn_relu_upsample_sum: bad leading dimension %d of term %d", ld, q);
        LM_REQUIRE((long)Hi[q] * Wi[q] * ld < (1L << 30), "gn_relu_upsample_sum: term %d: an image of %ld elements does not fit 32-bit byte offsets", q,
                   (long)Hi[q] * Wi[q] * ld);
        P.t[k] = GnTerm{x[q], stats[q], Hi[q], Wi[q], ld, Ho > 1 ? (float)(Hi[q] - 1) / (float)(Ho - 1) : 0.f,
                        Wo > 1 ? (float)(Wi[q] - 1) / (float)(Wo - 1) : 0.f};
    }
    const int c4n = C / 4;
    LM_REQUIRE(256 % c4n == 0 || !Q.w, "gn_relu_upsample_sum: the fused 1x1 projection needs C/4 = %d to divide 256", c4n);
    LM_REQUIRE(Ho <= 65535 && B <= 65535, "gn_relu_upsample_sum: grid too large");
    int c4shift = -1;
    for (int sft = 0; sft < 16; ++sft)
        if ((1 << sft) == c4n) c4shift = sft;
    int same = 0;
    for (int k = 0; k < n; ++k)
        if (Hi[k] == Ho && Wi[k] == Wo) same |= 1 << k;
    if (n == 1 && same == 0 && !Q.w && y && 256 % c4n == 0 && Hi[0] > 1 && Wi[0] > 1 && Ho >= 2 * Hi[0] - 1 && Wo >= 2 * Wi[0] - 1) {
        // one up-sampling term (scale <= 1/2): source block staged in LDS
        static const bool lds_up = [] { const char* e = getenv("LM_GN_UP_LDS"); return !e || atoi(e) != 0; }();
        const int SR = up_block_extent(Hi[0], Ho, UT_R), SC = up_block_extent(Wi[0], Wo, UT_C);
        const size_t lds = (size_t)SR * SC * C * sizeof(float);
        if (lds_up && lds <= 72 * 1024) {
            if (int e = lm_ensure_dynamic_lds((const void*)gn_relu_up_lds_kernel, lds)) return e;
            hipLaunchKernelGGL(gn_relu_up_lds_kernel, dim3((unsigned)lm_cdiv(Wo, UT_C), (unsigned)lm_cdiv(Ho, UT_R), (unsigned)B), dim3(256), lds,
                               (hipStream_t)stream, P.t[0], gamma, beta, y, Ho, Wo, C, SR, SC);
            LM_LAUNCH_CHECK();
            return LM_OK;
        }
    }
    if (n == 3 && same == 5 && Q.w && c4n >= 8 && c4n <= 64 && (c4n & (c4n - 1)) == 0 && Hi[1] > 1 && Wi[1] > 1 && Ho >= 2 * Hi[1] - 1 &&
        Wo >= 2 * Wi[1] - 1) {
        // s2 + s3 + s4 (+ the branch's 1x1 output layer): the up-sampled middle term through LDS, affines and weights once per thread
        static const bool lds_sum = [] { const char* e = getenv("LM_GN_SUM_LDS"); return !e || atoi(e) != 0; }();
        const int SR = up_block_extent(Hi[1], Ho, UT_R), SC = up_block_extent(Wi[1], Wo, UT_C);
        const size_t lds = (size_t)SR * SC * C * sizeof(float);
        if (lds_sum && lds <= 64 * 1024) {
            if (int e = lm_ensure_dynamic_lds((const void*)gn_sum3_lds_kernel, lds)) return e;
            hipLaunchKernelGGL(gn_sum3_lds_kernel, dim3((unsigned)lm_cdiv(Wo, UT_C), (unsigned)lm_cdiv(Ho, UT_R), (unsigned)B), dim3(256), lds,
                               (hipStream_t)stream, P, gamma, beta, y, Ho, Wo, C, SR, SC, Q);
            LM_LAUNCH_CHECK();
            return LM_OK;
        }
    }
    const dim3 grid((unsigned)lm_cdiv((long)Wo * c4n, 256), (unsigned)Ho, (unsigned)B);
#define LM_GNS(NN, SS)                                                                                                           \
    case (NN) * 8 + (SS):                                                                                                        \
        hipLaunchKernelGGL((gn_relu_upsample_sum_kernel<NN, SS>), grid, dim3(256), 0, (hipStream_t)stream, P, gamma, beta, y, Ho, Wo, C, \
                           c4shift, Q);                                                                                         \
        break;
    switch (n * 8 + same) {
        LM_GNS(1, 0) LM_GNS(1, 1)
        LM_GNS(2, 0) LM_GNS(2, 1) LM_GNS(2, 2) LM_GNS(2, 3)
        LM_GNS(3, 0) LM_GNS(3, 1) LM_GNS(3, 2) LM_GNS(3, 3) LM_GNS(3, 4) LM_GNS(3, 5) LM_GNS(3, 6) LM_GNS(3, 7)
    }
#undef LM_GNS
    LM_LAUNCH_CHECK();
    return LM_OK;
}
}  // namespace

LM_API int lm_gn_relu_upsample_sum(void* stream, int n, const float* const* x, const float* const* stats, const int* Hi, const int* Wi,
                                   const int* ldx, const float* gamma, const float* beta, float* y, int B, int Ho, int Wo, int C) {
    LM_REQUIRE(y, "gn_relu_upsample_sum: null output");
    return launch_gn_sum(stream, n, x, stats, Hi, Wi, ldx, gamma, beta, y, B, Ho, Wo, C, Proj1x1{nullptr, nullptr, nullptr, 0, 0});
}

// Same sum followed by a 1x1 convolution y1 = sum @ w + bias (cout <= 8, w in the [C][16] layout of lm_conv2d_nhwc_small) computed from
// registers; y may be NULL, in which case the C-channel sum is never written (feature_layer / output_layer_endp read nothing else).
LM_API int lm_gn_relu_upsample_sum_conv1x1(void* stream, int n, const float* const* x, const float* const* stats, const int* Hi,
                                           const int* Wi, const int* ldx, const float* gamma, const float* beta, float* y, int B, int Ho,
                                           int Wo, int C, const float* w_c16, const float* bias, int cout, float* y1, int ldy1) {
    LM_REQUIRE(w_c16 && y1 && cout >= 1 && cout <= 8 && ldy1 >= cout, "gn_relu_upsample_sum_conv1x1: bad projection (cout=%d)", cout);
    const int c4n = C / 4;
    LM_REQUIRE(C % 4 == 0 && c4n >= 8 && c4n <= 64 && (c4n & (c4n - 1)) == 0,
               "gn_relu_upsample_sum_conv1x1: C=%d: C/4 must be a power of two in [8, 64] (one pixel per wave segment)", C);
    return launch_gn_sum(stream, n, x, stats, Hi, Wi, ldx, gamma, beta, y, B, Ho, Wo, C, Proj1x1{w_c16, bias, y1, cout, ldy1});
}

LM_API int lm_upsample_bilinear_nhwc(void* stream, const float* x, int ldx, const float* add, int lda, float* y, int ldy,
                                     int B, int Hi, int Wi, int Ho, int Wo, int C) {
    LM_REQUIRE(x && y && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!add || lda % 4 == 0), "upsample_nhwc: bad args");
    const long total4 = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(upsample_nhwc_kernel, dim3(lm_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, add, lda, y, ldy, Hi, Wi, Ho, Wo, C, total4);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_upsample_bilinear_to_chw(void* stream, const float* x, int ldx, float* y, int B, int Hi, int Wi,
                                       int Ho, int Wo, int C) {
    LM_REQUIRE(x && y && C >= 1, "upsample_to_chw: bad args");
    const long total = (long)B * C * Ho * Wo;
    if (Wo % 4 == 0 && total / 4 < (1L << 31) && ((uintptr_t)y & 15) == 0) {
        hipLaunchKernelGGL(upsample_to_chw4_kernel, dim3(lm_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           x, ldx, y, Hi, Wi, Ho, Wo, C, (unsigned)(total / 4));
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    hipLaunchKernelGGL(upsample_to_chw_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, y, Hi, Wi, Ho, Wo, C, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_layernorm_rows(void* stream, const float* x, const float* gamma, const float* beta, float* y,
                             long rows, int D, float eps) {
    LM_REQUIRE(x && gamma && beta && y && (D == 512 || D == 1024), "layernorm: D=%d must be 512 or 1024", D);
    if (D == 512)
        hipLaunchKernelGGL(layernorm_kernel<8>, dim3(lm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<16>, dim3(lm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, eps);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_unpatchify(void* stream, const float* tokens, float* y_nhwc, int B, int G, int P, int C) {
    LM_REQUIRE(tokens && y_nhwc, "unpatchify: null pointer");
    const long total = (long)B * G * P * G * P * C;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, tokens, y_nhwc, G, P, C, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

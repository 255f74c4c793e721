// Winograd F(2x2, 3x3) convolution on the matrix cores, fp32: 3x3 / stride 1 / pad == dilation layers of the FPN
// (baseline/models/pcencoder/postprojector.py:322-338 BasicBlock convs, :597-599 smooth*, :615-647 conv2/conv3/
// semantic_branch*) need 16 instead of 36 multiplies per 2x2 output block and (cin, cout) pair.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        (Lavin & Gray; transform matrices below)
//
// fp32 error of F(2,3) stays at the level of a re-ordered direct sum (transforms use only 0, +-1, 1/2): measured through the
// whole network 1e-6 .. 3e-6 of the tensor scale, the same as the direct kernel against the reference.
//
// Two kernels:
//   wino_input_kernel   V[xi][tile][c] = (B^T d B)[xi] for every 4x4 input patch (stride 2, zero padded); memory bound,
//                       writes 4x the input.  A dilated layer is d x d interleaved plain convolutions: the tile index runs
//                       over (image, phase, ty, tx) and the patch samples the input with pitch d.
//   wino_gemm_kernel    one workgroup = 128 tiles (= 512 output pixels) x BN output channels.  For xi = 0..15 it runs the K
//                       loop over the input channels (same global->LDS double-buffered slab pipeline and 32x32x2 MFMA
//                       fragment scheme as conv_mfma.hip) into a temporary accumulator and folds it with the +-1
//                       coefficients of A^T . A into FOUR output accumulators (the 2x2 pixels of every tile), so the
//                       transformed products M[xi] never go to memory.  Epilogue = BN scale/shift, residual, ReLU, NHWC.
// U = G g G^T is computed at weight-packing time ([16][CoutP][Cin], lanemapping_amd/ops.py pack_wino).
#include "common.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;            // channel granularity of the path (Cin % 32 == 0); the GEMM's K slab is KB = 32 or 64 floats

__device__ __attribute__((aligned(16))) float g_wino_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct WinoGeom {
    int B, H, W, dil, Ty, Tx;   // Ty x Tx tiles per (image, phase); dil*dil phases
    int Timg, Tpad;             // real tiles per image, and that count rounded up to 128 (a workgroup never straddles images)
    long T;                     // B * Tpad rows of V
};

// row m of V -> tile; false for the padding rows at the end of every image
__device__ __forceinline__ bool tile_decode(const WinoGeom& g, long m, int& b, int& pa, int& pb, int& ty, int& tx) {
    b = (int)((unsigned)m / (unsigned)g.Tpad);                  // T < 2^31 (checked at launch): 32-bit divisions only
    int t = (int)((unsigned)m - (unsigned)b * (unsigned)g.Tpad);
    if (t >= g.Timg) return false;
    tx = t % g.Tx;
    t /= g.Tx;
    ty = t % g.Ty;
    const int ph = t / g.Ty;
    pa = ph / g.dil;
    pb = ph % g.dil;
    return true;
}

// B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] (rows first, then columns) of one 4x4 patch x 4 channels -> V[xi][m][c4..]
__device__ __forceinline__ void transform_store(const f32x4 (&d)[4][4], float* __restrict__ V, long T, long m, int C, int c4) {
    f32x4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j];
        r[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v0 = r[i][0] - r[i][2], v1 = r[i][1] + r[i][2], v2 = r[i][2] - r[i][1], v3 = r[i][1] - r[i][3];
        float* o = V + ((long)(4 * i) * T + m) * C + c4;
        *reinterpret_cast<f32x4*>(o) = v0;
        *reinterpret_cast<f32x4*>(o + T * C) = v1;
        *reinterpret_cast<f32x4*>(o + 2 * T * C) = v2;
        *reinterpret_cast<f32x4*>(o + 3 * T * C) = v3;
    }
}

// grid: ceil(T * C/4 / 256)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, WinoGeom g, int C, float* __restrict__ V) {
    const int c4n = C / 4;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; give every XCD one contiguous run of tiles so that the
    // 4-fold overlap of neighbouring 4x4 patches is served by ITS L2 (otherwise each input pixel reaches the fabric ~2.6 times)
    unsigned bid = blockIdx.x;
    {
        const unsigned nb = gridDim.x, per = nb / 8, full = per * 8;
        if (bid < full) bid = (bid % 8) * per + bid / 8;
    }
    const long e = (long)bid * 256 + threadIdx.x;
    if (e >= g.T * c4n) return;
    const int c4 = (int)(e % c4n) * 4;
    const long m = e / c4n;
    int b, pa, pb, ty, tx;
    if (!tile_decode(g, m, b, pa, pb, ty, tx)) return;      // padding row: never read (the GEMM substitutes zeros)
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = (2 * ty + i - 1) * g.dil + pa;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = (2 * tx + j - 1) * g.dil + pb;
            d[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)g.H && (unsigned)xx < (unsigned)g.W && (2 * ty + i - 1) >= 0 && (2 * tx + j - 1) >= 0)
                d[i][j] = *reinterpret_cast<const f32x4*>(x + (((long)b * g.H + y) * g.W + xx) * ldx + c4);
        }
    }
    transform_store(d, V, g.T, m, C, c4);
}

// ---- fused producer: V of  bilinear_x2(relu(gn(t)))  without materialising the upsampled tensor ----------------------------------
// The FPN's s4 = _upsample(relu(gn(conv(p4)))) (postprojector.py:615-618) is consumed by exactly one 3x3 convolution; this kernel
// writes that convolution's Winograd input straight from the low-resolution tensor t [B][Hi][Wi][C]: the GN + ReLU'd source pixels a
// group of tiles needs are staged in LDS once, every thread blends the 4x4 patch of its (tile, 4 channels) from them with the same
// fixed-order arithmetic as lm_gn_relu_upsample (common.h helpers: identical bits) and applies B^T d B.  Saves the 0.68 GB write
// and re-read of the upsampled tensor per call (B = 8).  Exact x2 geometry only (Ho = 2 Hi, Wo = 2 Wi, dilation 1).
// block = 256 threads = TXB tiles of one tile row x C/4 channel quads; grid = B * Ty * ceil(Tx / TXB)
template <int C4N>
__global__ __launch_bounds__(256) void wino_input_gn_up2_kernel(const float* __restrict__ t, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int Hi, int Wi, int ldt, WinoGeom g, float* __restrict__ V) {
    constexpr int TXB = 256 / C4N, RC = TXB + 3, C = C4N * 4;    // region: 4 source rows x (TXB + 3) source columns (scale < 1/2)
    __shared__ __attribute__((aligned(16))) float S[4 * RC * C];
    unsigned bid = blockIdx.x;
    {
        const unsigned nb = gridDim.x, per = nb / 8, full = per * 8;   // XCD-contiguous order, as in wino_input_kernel
        if (bid < full) bid = (bid % 8) * per + bid / 8;
    }
    const int gx = (g.Tx + TXB - 1) / TXB;
    const int txg = (int)(bid % (unsigned)gx);
    const int ty = (int)((bid / (unsigned)gx) % (unsigned)g.Ty);
    const int b = (int)(bid / ((unsigned)gx * (unsigned)g.Ty));
    const int tx0 = txg * TXB;
    const int tid = threadIdx.x;
    // region origin = first source tap of the first output row / column this block touches
    int yb, xb, i1_, dummy;
    float w0_, w1_;
    lm_bilin_axis(max(2 * ty - 1, 0), Hi, g.H, yb, i1_, w0_, w1_);
    lm_bilin_axis(max(2 * tx0 - 1, 0), Wi, g.W, xb, dummy, w0_, w1_);
    {
        const int c4 = (tid % C4N) * 4;
        f32x4 a, gg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ae, ge;
            lm_gn_affine(stats[((long)b * C + c4 + e) * 2], stats[((long)b * C + c4 + e) * 2 + 1], gamma[c4 + e], beta[c4 + e], ae, ge);
            a[e] = ae;
            gg[e] = ge;
        }
        for (int p = tid / C4N; p < 4 * RC; p += TXB) {          // (a thread keeps its channel quad: one affine for all its pixels)
            const int ry = p / RC, rx = p % RC;
            const int sy = min(yb + ry, Hi - 1), sx = min(xb + rx, Wi - 1);
            f32x4 v = *reinterpret_cast<const f32x4*>(t + (((long)b * Hi + sy) * Wi + sx) * ldt + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lm_gn_relu(v[e], a[e], gg[e]);
            *reinterpret_cast<f32x4*>(S + (ry * RC + rx) * C + c4) = v;
        }
    }
    __syncthreads();
    const int tx = tx0 + tid / C4N, c4 = (tid % C4N) * 4;
    if (tx >= g.Tx) return;
    f32x4 d[4][4];
    int ry0[4], ry1[4], rx0[4], rx1[4];
    float wy0[4], wy1[4], wx0[4], wx1[4];
    bool oky[4], okx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = 2 * ty + i - 1, ox = 2 * tx + i - 1;
        oky[i] = (unsigned)oy < (unsigned)g.H;
        okx[i] = (unsigned)ox < (unsigned)g.W;
        int y0, y1, x0, x1;
        lm_bilin_axis(oky[i] ? oy : 0, Hi, g.H, y0, y1, wy0[i], wy1[i]);
        lm_bilin_axis(okx[i] ? ox : 0, Wi, g.W, x0, x1, wx0[i], wx1[i]);
        ry0[i] = y0 - yb; ry1[i] = y1 - yb; rx0[i] = x0 - xb; rx1[i] = x1 - xb;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (oky[i] && okx[j]) {
                const f32x4 v00 = *reinterpret_cast<const f32x4*>(S + (ry0[i] * RC + rx0[j]) * C + c4);
                const f32x4 v01 = *reinterpret_cast<const f32x4*>(S + (ry0[i] * RC + rx1[j]) * C + c4);
                const f32x4 v10 = *reinterpret_cast<const f32x4*>(S + (ry1[i] * RC + rx0[j]) * C + c4);
                const f32x4 v11 = *reinterpret_cast<const f32x4*>(S + (ry1[i] * RC + rx1[j]) * C + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[i][j][e] = lm_bilerp(v00[e], v01[e], v10[e], v11[e], wy0[i], wy1[i], wx0[j], wx1[j]);
            }
        }
    const long m = (long)b * g.Tpad + (long)ty * g.Tx + tx;      // dilation 1: one phase, row-major tiles (tile_decode)
    transform_store(d, V, g.T, m, C, c4);
}

struct WinoParams {
    const float* V; const float* U; const float* scale; const float* shift; const float* res; float* y; const float* zero;
    int ldr, ldy, C, Cout, CoutP, act;
    double* gn_part;   // optional [B][Tpad/32][Cout][2]: per (image, 32-tile chunk, channel) sum / sum of squares of the outputs
    WinoGeom g;
};

template <int BM, int BN, int WM, int WN, int KB = 32, int NBUF = 2>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void wino_gemm_kernel(WinoParams p) {
    constexpr int LDS_LD = KB;                        // unpadded, lane-linear rows (required by global_load_lds)
    constexpr int LPRW = KB / 4;                      // lanes (16-byte chunks) per row: 8 or 16
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int RPP = NT / LPRW;                    // rows covered by one load pass
    constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
    static_assert(RPP % 16 == 0 && BM % RPP == 0 && BN % RPP == 0, "a load pass covers whole swizzle periods (16 rows)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + NBUF * BM * LDS_LD;     // As [NBUF][BM][KB], Bs [NBUF][BN][KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps the LDS destinations of the loads in SGPRs
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int n_tiles = (p.Cout + BN - 1) / BN;
    unsigned bid = blockIdx.x;
    {
        const unsigned nb = gridDim.x, per = nb / 8, full = per * 8;   // XCD-aware order (see conv_mfma.hip)
        if (bid < full) bid = (bid % 8) * per + bid / 8;
    }
    const long m0 = (long)(bid / n_tiles) * BM;
    const int n0 = (bid % n_tiles) * BN;
    // XOR swizzle of the 16-byte chunks, applied to the SOURCE address (the LDS image is lane-linear) and to the fragment
    // reads: 128-byte rows (KB 32) repeat banks every 2 rows -> key (row >> 1) & 7; 256-byte rows (KB 64) every row -> row & 7
    auto swz = [](int row) { return KB == 32 ? (row >> 1) & 7 : row & 7; };
    const int lrow = tid / LPRW;
    const int lc4 = ((tid % LPRW) ^ swz(lrow)) * 4;
    const long T = p.g.T;
    // Address generation: ONE per-lane pointer per operand; everything else (load pass, xi, channel slab) is a wave-uniform
    // byte offset kept in scalar registers, so a load costs one 64-bit add.  Every row m0 .. m0+BM-1 exists in V (T is a
    // multiple of BM and the allocation covers the padding rows of each image): padding rows are read as they are (never
    // written, arbitrary bits) - MFMA rows are independent and the epilogue discards those rows, so no zero substitution.
    const float* const pa = p.V + ((m0 + lrow) * p.C + lc4);
    const float* const pb = p.U + ((long)(n0 + lrow) * p.C + lc4);
    const int cslabs = p.C / KB;
    const long xi_stride_a = T * p.C;
    const int xi_stride_b = p.CoutP * p.C;
    const long pass_stride_a = (long)RPP * p.C, pass_stride_b = (long)RPP * p.C;
    int cur_cs = 0, cur_xi = 0;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    auto gload = [&](int buf) {
#ifdef LM_ABL_LOADSAME                                          // timing ablation: every slab re-reads the first one (cache resident)
        const long adelta = 0, bdelta = 0;
#else
        const long adelta = cur_xi * xi_stride_a + cur_cs * KB;
        const long bdelta = (long)cur_xi * xi_stride_b + cur_cs * KB;
#endif
#ifndef LM_ABL_NOALOAD
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(pa + (adelta + i * pass_stride_a)),
                                             (lptr_t*)(As + (buf * BM + i * RPP + wave * (64 / LPRW)) * LDS_LD), 16, 0, 0);
#endif
#ifndef LM_ABL_NOBLOAD
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(pb + (bdelta + i * pass_stride_b)),
                                             (lptr_t*)(Bs + (buf * BN + i * RPP + wave * (64 / LPRW)) * LDS_LD), 16, 0, 0);
#endif
        if (++cur_cs == cslabs) {
            cur_cs = 0;
            ++cur_xi;
        }
    };

    f32x16 out[2][2][TM][TN];      // [a][b]: output pixel (2 ty + a, 2 tx + b) of every tile
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[a][b][i][j][r] = 0.f;

    // NBUF == 3: the loads run TWO slabs ahead (a 32-MFMA slab is only ~0.9 us of matrix work, less than an L2 / Infinity
    // Cache round trip under load), so the end-of-slab wait leaves the youngest slab's loads in flight: explicit
    // s_waitcnt vmcnt(loads per slab) + s_barrier instead of __syncthreads() (which always waits for vmcnt(0)).
    constexpr int LOADS = A_LOADS + B_LOADS;
    static_assert(NBUF == 2 || (NBUF == 3 && LOADS == 6), "the vmcnt immediates below assume 6 loads per slab");
    const int KT = 16 * cslabs;
    gload(0);
    if (NBUF == 3) {
        if (KT > 1) {
            gload(1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __syncthreads();
    }
    const int frow = lane & 31, fswz = swz(frow), fhalf = lane >> 5;
    int kt = 0;
#ifndef LM_ABL_XI
#define LM_ABL_XI 16
#endif
    for (int xi = 0; xi < LM_ABL_XI; ++xi) {      // (timing ablations shorten the loop)
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int cs = 0; cs < cslabs; ++cs, ++kt) {
            const int buf = NBUF == 2 ? (kt & 1) : kt % 3;
#ifndef LM_ABL_NOLOAD
            if (NBUF == 2) {
                if (kt + 1 < KT) gload(buf ^ 1);
            } else if (kt + 2 < KT) {
                gload((kt + 2) % 3);
            }
#endif
            const float* Ab = As + (buf * BM + wm0 + frow) * LDS_LD;
            const float* Bb = Bs + (buf * BN + wn0 + frow) * LDS_LD;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 8) {
                const int fo = (((kk >> 2) + fhalf) ^ fswz) * 4;
                f32x4 af[TM], bf[TN];
#pragma unroll
#ifdef LM_ABL_NOLDS
                for (int i = 0; i < TM; ++i) af[i] = f32x4{(float)kk, 1.f, (float)fo, (float)lane};
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = f32x4{1.f, (float)kt, 3.f, (float)lane};
#else
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_LD + fo);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_LD + fo);
#endif
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
            }
            if (NBUF == 3) {
                if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // slab kt+1 has landed, kt+2 may be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
#ifndef LM_ABL_NOBAR
                __syncthreads();
#endif
            }
        }
#ifndef LM_ABL_NOFOLD
        // fold M[xi] into the 2x2 outputs: Y[a][b] += AT[a][i] * AT[b][j] * M[i][j],  A^T = [1 1 1 0; 0 1 -1 -1].
        // The coefficients are 0 / +1 / -1, so the multiply-adds below are exact additions (or no-ops).
        const int wi = xi >> 2, wj = xi & 3;
        const float ca[2] = {wi < 3 ? 1.f : 0.f, wi == 0 ? 0.f : (wi == 1 ? 1.f : -1.f)};
        const float cb[2] = {wj < 3 ? 1.f : 0.f, wj == 0 ? 0.f : (wj == 1 ? 1.f : -1.f)};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float c = ca[a] * cb[b];
                if (c == 0.f) continue;                        // (wave-uniform) 28 of the 64 (xi, position) pairs
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) out[a][b][i][j][r] = fmaf(acc[i][j][r], c, out[a][b][i][j][r]);
            }
#else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[0][0][i][j][r] += acc[i][j][r];     // ablation: keep the accumulator live, skip the fold
#endif
    }

#ifdef LM_ABL_NOEPI
    {   // timing ablation: no output transform / stores, one value per thread keeps the accumulators live
        float sum = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += out[a][b][i][j][r];
        p.y[(long)bid * 256 + tid] = sum;
        return;
    }
#endif
    // --- epilogue: per output position (a, b) transpose the wave tile through LDS (16-byte coalesced channel vectors)
    constexpr int ELD = WN + 4;
    float* stage = smem + wave * (WM * ELD);
    const int half = lane >> 5;
    constexpr int LPR = WN / 4, RPI = 64 / LPR;
    const int c4 = (lane % LPR) * 4;
    const int n = n0 + wn0 + c4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) {
                if (p.scale) sc[e] = p.scale[n + e];
                if (p.shift) sh[e] = p.shift[n + e];
            }
    }
    const bool vec = (n + 3 < p.Cout) && ((p.ldy & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
    f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};   // GroupNorm(C,C) statistics of this wave tile (all 4 positions)
    // Output pixel of every row this lane stores, decoded ONCE (the four positions (a, b) of a tile differ by constant pixel
    // offsets): one division-based decode for the lane's first row, then carries - the rows of a lane are RPI tiles apart.
    // (Decoding per position and row cost ~250 VALU instructions x 32 per lane = 6 % of the kernel.)
    constexpr int NP = WM / RPI;
    int pix0[NP];
    unsigned vmask = 0;                                        // 3 bits per row: tile exists | a = 1 inside | b = 1 inside
    {
        const int bi = (int)(m0 / p.g.Tpad);                   // a workgroup tile never straddles images (Tpad % BM == 0)
        int t = (int)(m0 - (long)bi * p.g.Tpad) + wm0 + lane / LPR;
        int tx = t % p.g.Tx, q = t / p.g.Tx;
        int ty = q % p.g.Ty, ph = q / p.g.Ty;
        int pa = ph / p.g.dil, pb = ph % p.g.dil;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int oy = 2 * ty * p.g.dil + pa, ox = 2 * tx * p.g.dil + pb;
            pix0[pass] = (bi * p.g.H + oy) * p.g.W + ox;
            if (t < p.g.Timg && oy < p.g.H && ox < p.g.W)
                vmask |= (1u | (oy + p.g.dil < p.g.H ? 2u : 0u) | (ox + p.g.dil < p.g.W ? 4u : 0u)) << (3 * pass);
            t += RPI;
            tx += RPI;
            while (tx >= p.g.Tx) {
                tx -= p.g.Tx;
                if (++ty >= p.g.Ty) {
                    ty = 0;
                    ++ph;
                    pa = ph / p.g.dil;
                    pb = ph % p.g.dil;
                }
            }
        }
    }
    const int step_a = p.g.dil * p.g.W, step_b = p.g.dil;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ELD + j * 32 + frow] = out[a][b][i][j][r];
            __syncthreads();
            if (n >= p.Cout) continue;
#pragma unroll
            for (int pass = 0; pass < WM / RPI; ++pass) {
                const int row = pass * RPI + lane / LPR;
                const unsigned vm = vmask >> (3 * pass);
                if (!(vm & 1u) || (a && !(vm & 2u)) || (b && !(vm & 4u))) continue;
                const long pix = pix0[pass] + a * step_a + b * step_b;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * ELD + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = p.scale ? v[e] * sc[e] + sh[e] : v[e] + sh[e];
                if (p.gn_part) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gs[e] += v[e];
                        gq[e] = fmaf(v[e], v[e], gq[e]);
                    }
                }
                if (vec) {
                    if (p.res) {
                        const f32x4 rr = *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rr[e];
                    }
                    if (p.act == LM_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + n) = v;
                } else {
                    for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                        float u = v[e];
                        if (p.res) u += p.res[pix * p.ldr + n + e];
                        if (p.act == LM_ACT_RELU) u = fmaxf(u, 0.f);
                        p.y[pix * p.ldy + n + e] = u;
                    }
                }
            }
        }
    if (p.gn_part && n < p.Cout) {   // fixed-order reduction over the RPI lanes that share a channel quad, then one writer lane
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gs[e] += __shfl_xor(gs[e], o);
                gq[e] += __shfl_xor(gq[e], o);
            }
        if (lane < LPR) {
            const long mt = m0 + wm0;                          // first row of this wave tile: one image (Tpad % 128 == 0)
            const long bi = mt / p.g.Tpad;
            const long chunk = (mt - bi * p.g.Tpad) / WM;
            double* o = p.gn_part + ((bi * (p.g.Tpad / WM) + chunk) * p.Cout + n) * 2;
            for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                o[2 * e] = (double)gs[e];
                o[2 * e + 1] = (double)gq[e];
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int KB = 32, int NBUF = 2>
int launch_wino(const WinoParams& p, hipStream_t stream) {
    LM_REQUIRE(p.C % KB == 0, "conv_wino: Cin=%d must be a multiple of the %d-float K slab", p.C, KB);
    const size_t kloop = (size_t)NBUF * (BM + BN) * KB, stage = (size_t)(BM / WM) * (BN / WN) * WM * (WN + 4);
    const size_t lds = (kloop > stage ? kloop : stage) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        LM_HIP(hipFuncSetAttribute((const void*)wino_gemm_kernel<BM, BN, WM, WN, KB, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const long blocks = ((p.g.T + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
    LM_REQUIRE(blocks > 0 && blocks < (1L << 31), "conv_wino: bad grid %ld", blocks);
    LM_REQUIRE(p.g.T % BM == 0, "conv_wino: %ld rows of V are not a multiple of the %d-row tile", p.g.T, BM);   // loads are unguarded
    LM_REQUIRE(p.g.Tpad % BM == 0, "conv_wino: %d tiles per image are not a multiple of the %d-row tile", p.g.Tpad, BM);   // epilogue decode
    LM_REQUIRE((long)p.g.B * p.g.H * p.g.W < (1L << 31) && p.g.T < (1L << 31), "conv_wino: output too large for 32-bit pixel indices");
    hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN, KB, NBUF>), dim3((unsigned)blocks), dim3((BM / WM) * (BN / WN) * 64), lds, stream, p);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

int wino_force() {   // LM_WINO_TILE: tile-variant experiments only
    static const int force = [] { const char* e = getenv("LM_WINO_TILE"); return e ? atoi(e) : 0; }();
    return force;
}

WinoGeom geom(int B, int H, int W, int dil) {
    WinoGeom g;
    g.B = B; g.H = H; g.W = W; g.dil = dil;
    g.Ty = ((H + dil - 1) / dil + 1) / 2;
    g.Tx = ((W + dil - 1) / dil + 1) / 2;
    g.Timg = dil * dil * g.Ty * g.Tx;
    const int unit = wino_force() == 8 ? 384 : 128;      // rows per workgroup (192-row variant: lcm with the 128 of the others)
    g.Tpad = (g.Timg + unit - 1) / unit * unit;
    g.T = (long)B * g.Tpad;
    return g;
}

}  // namespace

LM_API long lm_conv3x3_winograd_workspace_bytes(int B, int H, int W, int Cin, int dil) {
    if (dil < 1) return 0;
    return 16 * geom(B, H, W, dil).T * (long)Cin * (long)sizeof(float);
}

// 32-tile chunks per image of the GroupNorm partial sums written by lm_winograd_gemm_f32 (-> lm_gn_finalize's nchunk)
LM_API int lm_winograd_gn_chunks(int H, int W, int dil) { return dil < 1 ? 0 : geom(1, H, W, dil).Tpad / 32; }

// V = B^T d B of every 4x4 patch: [16][B * Tpad][Cin] (lm_conv3x3_winograd_workspace_bytes).  Several convolutions that
// read the same tensor (the two semantic branches of the FPN) share one transform.
LM_API int lm_winograd_input_transform_f32(void* stream, const float* x, int ldx, int B, int H, int W, int Cin, int dil, void* V,
                                           long V_bytes) {
    LM_REQUIRE(x && V, "wino_input: null pointer");
    LM_REQUIRE(Cin > 0 && Cin % BK == 0 && dil >= 1 && B > 0 && H > 0 && W > 0, "wino_input: bad shape (Cin=%d must be a multiple of %d)", Cin, BK);
    LM_REQUIRE(ldx >= Cin && ldx % 4 == 0, "wino_input: bad leading dim ldx=%d", ldx);
    LM_REQUIRE(lm_conv3x3_winograd_workspace_bytes(B, H, W, Cin, dil) <= V_bytes, "wino_input: V buffer too small");
    const WinoGeom g = geom(B, H, W, dil);
    LM_REQUIRE(g.T < (1L << 31), "wino_input: too many tiles");
    const long in_threads = g.T * (Cin / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((in_threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, g, Cin,
                       (float*)V);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// V of the tensor  bilinear_align_corners_x2(relu(gn(t; stats, gamma, beta)))  [B][2 Hi][2 Wi][C] that is never materialised
// (lm_gn_relu_upsample followed by lm_winograd_input_transform_f32, bit-identical to that pair).  C = 128 or 256, dilation 1;
// ldt = floats between pixels of t (a channel slice of a wider tensor is fine).
LM_API int lm_winograd_input_transform_gn_up2_f32(void* stream, const float* t, int ldt, const float* stats, const float* gamma,
                                                  const float* beta, int B, int Hi, int Wi, int C, void* V, long V_bytes) {
    LM_REQUIRE(t && stats && gamma && beta && V, "wino_input_gn_up2: null pointer");
    LM_REQUIRE(ldt >= C && ldt % 4 == 0, "wino_input_gn_up2: bad leading dimension ldt=%d", ldt);
    LM_REQUIRE((C == 128 || C == 256) && B > 0 && Hi > 1 && Wi > 1, "wino_input_gn_up2: C=%d must be 128 or 256, source at least 2x2", C);
    const int H = 2 * Hi, W = 2 * Wi;
    LM_REQUIRE(lm_conv3x3_winograd_workspace_bytes(B, H, W, C, 1) <= V_bytes, "wino_input_gn_up2: V buffer too small");
    const WinoGeom g = geom(B, H, W, 1);
    LM_REQUIRE(g.T < (1L << 31), "wino_input_gn_up2: too many tiles");
    const int txb = 256 / (C / 4);
    const long blocks = (long)B * g.Ty * ((g.Tx + txb - 1) / txb);
    LM_REQUIRE(blocks < (1L << 31), "wino_input_gn_up2: bad grid");
    if (C == 256)
        hipLaunchKernelGGL(wino_input_gn_up2_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, stats, gamma, beta, Hi, Wi, ldt, g, (float*)V);
    else
        hipLaunchKernelGGL(wino_input_gn_up2_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, stats, gamma, beta, Hi, Wi, ldt, g, (float*)V);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// y = act((sum_xi V[xi] U[xi]^T folded by A^T . A) * scale + shift + res), NHWC.  wu: [16][CoutP][Cin] = (G g G^T)[xi = 4i + j].
// gn_partial (optional, needs res == NULL and act == none): [B][lm_winograd_gn_chunks][Cout][2] doubles, sum / sum of squares
// of the outputs per (image, chunk, channel) -> lm_gn_finalize (first pass of GroupNorm(C,C) without re-reading y).
LM_API int lm_winograd_gemm_f32(void* stream, const void* V, const float* wu, int CoutP, const float* scale, const float* shift,
                                const float* res, int ldr, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int dil, int act,
                                double* gn_partial) {
    LM_REQUIRE(V && wu && y, "conv_wino: null pointer");
    LM_REQUIRE(Cin > 0 && Cin % BK == 0 && dil >= 1 && B > 0 && H > 0 && W > 0, "conv_wino: bad shape (Cin=%d must be a multiple of %d)", Cin, BK);
    LM_REQUIRE(CoutP >= Cout && CoutP % 128 == 0, "conv_wino: CoutP=%d must be Cout=%d rounded up to 128", CoutP, Cout);
    LM_REQUIRE(ldy >= Cout, "conv_wino: bad leading dim ldy=%d", ldy);
    LM_REQUIRE(act == LM_ACT_NONE || act == LM_ACT_RELU, "conv_wino: activation %d not supported", act);
    LM_REQUIRE((long)16 * CoutP * Cin < (1L << 31), "conv_wino: weights too large");
    LM_REQUIRE(!gn_partial || (res == nullptr && act == LM_ACT_NONE && Cout % 4 == 0), "conv_wino(gn stats): no residual / activation");
    WinoParams p;
    p.g = geom(B, H, W, dil);
    p.V = (const float*)V; p.U = wu; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.ldr = ldr; p.ldy = ldy; p.C = Cin; p.Cout = Cout; p.CoutP = CoutP; p.act = act;
    p.gn_part = gn_partial;
    static const float* zero = nullptr;
    if (!zero) {
        void* sym = nullptr;
        LM_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_wino_zero)));
        zero = (const float*)sym;
    }
    p.zero = zero;
    hipStream_t s = (hipStream_t)stream;
    const int force = wino_force();
    if (force == 8) return launch_wino<192, 64, 96, 32>(p, s);         // 4 waves of 96x32: three accumulator tiles (chains) per wave
    if (force == 9) return launch_wino<128, 64, 32, 64, 32, 3>(p, s);   // loads two slabs ahead (72 KB LDS, still 2 workgroups/CU)
    if (force == 11) return launch_wino<128, 64, 32, 32>(p, s);    // 8 waves of 32x32: half the loads per wave, 4 waves per SIMD if <= 128 registers
    if (force == 10) return launch_wino<64, 64, 32, 32>(p, s);     // 4 waves of 32x32: ~120 registers, 4 workgroups/CU
    if (force == 1 && !gn_partial) return launch_wino<128, 128, 64, 64>(p, s);
    if (force == 3 && !gn_partial) return launch_wino<64, 128, 32, 64>(p, s);
    if (force == 4) return launch_wino<256, 64, 32, 64>(p, s);
    if (force == 5) return launch_wino<128, 128, 64, 32>(p, s);
    if (force == 6 && Cin % 64 == 0) return launch_wino<128, 128, 64, 32, 64>(p, s);   // 8 waves, 64-float K slabs, 128 KB LDS
    if (force == 7 && Cin % 64 == 0) return launch_wino<128, 64, 32, 64, 64>(p, s);    // 4 waves, 64-float K slabs, 96 KB LDS
    // 128 x 64 tiles: 4 output + 1 temporary accumulator sets = 246 registers -> two workgroups per CU.  The 128 x 128 tile
    // (471 registers, one workgroup per CU) measured 1.32x over the direct kernel on 256->256@288^2, this one 1.52x.
    return launch_wino<128, 64, 32, 64>(p, s);
}

// Transform + GEMM in one call (workspace = V).
LM_API int lm_conv3x3_winograd_f32(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                                   const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W, int Cin,
                                   int Cout, int dil, int act, void* workspace, long workspace_bytes) {
    if (int e = lm_winograd_input_transform_f32(stream, x, ldx, B, H, W, Cin, dil, workspace, workspace_bytes)) return e;
    return lm_winograd_gemm_f32(stream, workspace, wu, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, nullptr);
}

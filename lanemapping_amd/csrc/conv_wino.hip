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
    if (int e = lm_ensure_dynamic_lds((const void*)wino_gemm_kernel<BM, BN, WM, WN, KB, NBUF>, lds)) return e;
    const long blocks = ((p.g.T + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
    LM_REQUIRE(blocks > 0 && blocks < (1L << 31), "conv_wino: bad grid %ld", blocks);
    LM_REQUIRE(p.g.T % BM == 0, "conv_wino: %ld rows of V are not a multiple of the %d-row tile", p.g.T, BM);   // loads are unguarded
    LM_REQUIRE(p.g.Tpad % BM == 0, "conv_wino: %d tiles per image are not a multiple of the %d-row tile", p.g.Tpad, BM);   // epilogue decode
    LM_REQUIRE((long)p.g.B * p.g.H * p.g.W < (1L << 31) && p.g.T < (1L << 31), "conv_wino: output too large for 32-bit pixel indices");
    hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN, KB, NBUF>), dim3((unsigned)blocks), dim3((BM / WM) * (BN / WN) * 64), lds, stream, p);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

WinoGeom geom(int B, int H, int W, int dil) {
    WinoGeom g;
    g.B = B; g.H = H; g.W = W; g.dil = dil;
    g.Ty = ((H + dil - 1) / dil + 1) / 2;
    g.Tx = ((W + dil - 1) / dil + 1) / 2;
    g.Timg = dil * dil * g.Ty * g.Tx;
    g.Tpad = (g.Timg + 127) / 128 * 128;                 // 128 rows of V per workgroup of the GEMM: a workgroup never straddles two images
    g.T = (long)B * g.Tpad;
    return g;
}


// =====================================================================================================================================
// Implicit-transform Winograd: no V tensor in HBM.  Three kernels share the scheme described here (it is the round-2 kernel's, a
// 64 tiles x 64 channels workgroup of four 32 x 32 waves, whose code is gone): wino_dual_kernel (fp32, 32 x 64, Cout <= 64),
// wino_pipe_kernel (fp32, 32 x 128, the transform spread over the MFMA steps) and wino_rows_split_kernel (bf16 x 3, 64 x 64, one
// transform row per wave, transform in registers) - their own headers say what differs.
// The loop order is (input-channel slab) outer, xi inner: all SIXTEEN transformed products M[xi] of the wave tile live in registers
// (16 accumulators x 16 = 256 AGPRs, hence one wave per SIMD).  Per 16-channel slab:
//   1. the RAW input patch of the 64 tiles (4 patch rows x up to 136 column slots x 16 channels, 34 KB) arrives in LDS through
//      global_load_lds gathers straight from the NHWC tensor (zero block for padding), issued during the previous slab's MFMA phase;
//   2. TRANSFORM phase: the 256 threads turn it into the slab's V = B^T d B for all 16 xi ONCE (thread = one tile x 4 channels: 16
//      ds_read_b128, 32 vector adds in the order of transform_store - rows first, then columns - 16 ds_write_b128) into a 64 KB LDS
//      buffer V[xi][tile][16 ch].  f32 MFMA executes on the SIMD's vector ALUs (it runs at the VALU rate and, measured, VALU time of
//      the same wave ADDS to MFMA time), so the transform must not be repeated per output-channel tile or per wave: the first version
//      of this kernel transformed in the A-fragment path (3 adds per MFMA, redundantly in both N waves) and lost 21 % to it;
//   3. MFMA phase: for xi = 0..15: two ds_read_b128 A fragments, 8 MFMAs into accumulator xi; B (U = G g G^T, repacked per wave
//      fragment: [xi][slab][32-channel tile][kk][lane][4]) goes global -> registers seven xi steps ahead (ring of 8 register sets).
// V has the bits of the materialising kernel and M[xi] accumulates over the same K order; the fold with A^T . A happens once at
// the end in ascending xi like the streaming kernel: y is bit-identical to wino_input_kernel + wino_gemm_kernel.
// Raw patch layout in LDS: cell (r, q) = patch row r (0..3), column slot q; the 64 tiles of a workgroup are consecutive in the
// linear (phase, ty, tx) order, i.e. up to INSEG runs of horizontally adjacent tiles; inside a run neighbouring tiles share two
// columns (slot of tile tl, patch column c: q = 2 tl + 2 run + c).  A cell is 16 floats = 4 chunks of 16 bytes; cell q sits at
// position rot(q) (low three bits rotated so that the cells of consecutive TILES differ in their low two position bits): the 16
// lanes of a ds_read_b128 group (4 tiles x 4 channel quads) then cover all 64 banks.  V rows are XOR-swizzled by (tile >> 2) & 3.
constexpr int IBM = 64, INSEG = 4;               // the largest tile block of the geometries below and the runs of adjacent tiles it can touch
__device__ __attribute__((aligned(16))) float g_wino_zeros[1024 + 32];   // zero source for padding cells, any channel slab (Cin <= 1024)

struct WinoImpParams {
    const float* x; const float* U; const float* scale; const float* shift; const float* res; float* y; const float* zeros;
    int ldx, ldr, ldy, C, Cout, NT, act;      // NT = CoutP / 32 channel tiles in U
    int n_inner;                               // (ROWS) workgroup order: N tile inner
    double* gn_part;
    WinoGeom g;
};

__device__ __forceinline__ int rot3(int q) { return (q & ~7) | ((q >> 1) & 3) | ((q & 1) << 2); }          // cell -> LDS position
__device__ __forceinline__ int unrot3(int p) { return (p & ~7) | ((p & 3) << 1) | ((p >> 2) & 1); }

// B fragment loads bypass the compiler's wait-count bookkeeping (it would drain the in-flight global_load_lds queue at every use):
// explicit s_waitcnt vmcnt(N) below, tied to the destination registers through "+v" operands.
__device__ __forceinline__ void bload2(f32x4 (&b)[2], unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %2, %3\n\t"
                 "global_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(b[0]), "=&v"(b[1]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void bwait(f32x4 (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}

// ---- split-precision variant (SPLIT = true): same kernel, same fp32 V slab in LDS; the GEMM runs on the bf16 matrix cores with fp32
// operands split into three bf16 pieces each, v = v1 + v2 + v3 EXACTLY (truncation split with bit masks: 8 + 8 + 8 significant bits;
// A fragments in registers when they are read, U at weight-packing time), and six of the nine piece
// products, v1u1 + v1u2 + v2u1 + v1u3 + v2u2 + v3u1 (smallest first), accumulated in the fp32 accumulators: relative error of a product
// <= 2^-23, the class of an fp32 rounding (profiles/r2_split_precision_study.txt: max error vs fp64 1.2e-6 on a 256-channel layer
// against 3.3e-6 for the fp32 Winograd, 0 decision flips outside the reference margin on the G10 tile).  Six 32x32x16 bf16 MFMAs of
// 32 cycles replace eight 32x32x2 f32 MFMAs of 64: 2.67x less matrix time, and the bf16 matrix cores do not share the vector ALUs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bload3(f32x4 (&b)[3], unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %3, %4\n\t"
                 "global_load_dwordx4 %1, %3, %4 offset:1024\n\t"
                 "global_load_dwordx4 %2, %3, %4 offset:2048"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void bwait3(f32x4 (&b)[3]) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
}

// Three bf16 pieces of the 8 fp32 values of an A fragment (a0 = channels 4g..4g+3, a1 = channels 8+4g..8+4g+3 of the slab for lane
// half g), split by truncation in registers: v & 0xFFFF0000 is the leading bf16 piece AS a float, v - piece is exact, twice; a piece
// pair is packed with one v_perm_b32 (the high halves of two registers).  4 VALU per value + 12 packs = 44 VALU per xi step; the bf16
// matrix cores do not share the vector ALUs, so this runs under the six MFMAs of the step.
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);      // {hi[31:16], lo[31:16]}
}
__device__ __forceinline__ void split3_frag(const f32x4 a0, const f32x4 a1, f32x4 (&piece)[3]) {
    float r[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        u32x4v pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack_hi(r[2 * e], r[2 * e + 1]);
        piece[k] = __builtin_bit_cast(f32x4, pk);
        if (k < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = r[e] - __uint_as_float(__float_as_uint(r[e]) & 0xFFFF0000u);      // exact
        }
    }
}

// TRANSFORM phase, one thread = (tile tl, channel quad qd): V[xi][tl][4 qd ..] = (B^T d B)[xi] with the arithmetic of transform_store
// Packed fp32 adds for the transform phases (no MFMA is in flight there): v_pk_add_f32 does two lanes of a sum per issue slot; the
// compiler selects it for a + b but turns a - b (and a + (-b)) into four scalar v_sub_f32, so the subtraction is spelled out with
// the negate modifiers.  IEEE fp32 adds either way: same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pk_sub4(const f32x4 a, const f32x4 b) {
    const f32x2 alo = __builtin_shufflevector(a, a, 0, 1), ahi = __builtin_shufflevector(a, a, 2, 3);
    const f32x2 blo = __builtin_shufflevector(b, b, 0, 1), bhi = __builtin_shufflevector(b, b, 2, 3);
    f32x2 rlo, rhi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rlo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rhi) : "v"(ahi), "v"(bhi));
    return __builtin_shufflevector(rlo, rhi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_add4(const f32x4 a, const f32x4 b) {
    const f32x2 alo = __builtin_shufflevector(a, a, 0, 1), ahi = __builtin_shufflevector(a, a, 2, 3);
    const f32x2 blo = __builtin_shufflevector(b, b, 0, 1), bhi = __builtin_shufflevector(b, b, 2, 3);
    f32x2 rlo, rhi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(rlo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(rhi) : "v"(ahi), "v"(bhi));
    return __builtin_shufflevector(rlo, rhi, 0, 1, 2, 3);
}


// One xi step of a slab's MFMA phase.  All VMEM traffic of a wave shares ONE in-order counter (vmcnt): waiting for this step's B
// fragments also waits for every patch load (global_load_lds) issued before them, and those come from HBM (~2 us under load) while
// B comes from L2.  So B runs BD = 7 steps (~1.5 us of matrix work) ahead in a ring of 8 register sets, the next slab's ILPW patch
// loads are issued in the first five steps (G = 2,2,2,2,1), and the end-of-slab wait leaves the seven youngest B sets in flight.
// NWAIT = loads allowed to stay outstanding when B of this step is needed = 2 * BD + the patch loads of the last BD + 1 steps.
// a_cur holds the A fragments of this xi (read behind the previous step's first MFMA); the reads of xi + 1 are pinned behind this
// step's first MFMA (the compiler waits with lgkmcnt(0) at the first use of LDS data whenever LDS-DMA is in flight, so the only
// reads outstanding at a wait must be the ones it needs).
constexpr int BD = 7;
template <int SLOT, int G, int NWAIT, bool NEXT, int NL>
__device__ __forceinline__ void wino_imp_step(f32x16& acc, f32x4 (&bq)[8][2], unsigned bvoff, const float* bpre_base, const float* anext0,
                                              const float* anext1, f32x4 (&a_cur)[2], f32x4 (&a_nxt)[2],
                                              const float* const (&gsrc)[NL], long goff, float* rawbuf, int wave, int& gnext) {
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
#ifndef LM_IABL_NOB                       // (timing ablations: tools/build_variant.sh)
    bload2(bq[(SLOT + BD) & 7], bvoff, bpre_base);
#endif
#ifndef LM_IABL_NOGLDS
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int s_ = gnext + g;
        __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[s_] + goff), (lptr_t*)(rawbuf + (s_ * 4 + wave) * 256), 16, 0, 0);
    }
#endif
    gnext += G;
    f32x4 (&bcur)[2] = bq[SLOT & 7];
#if !defined(LM_IABL_NOB) && !defined(LM_IABL_NOGLDS)
    bwait<NWAIT>(bcur);
#else
    bwait<0>(bcur);
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0][0], bcur[0][0], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) {
        a_nxt[0] = *reinterpret_cast<const f32x4*>(anext0);
        a_nxt[1] = *reinterpret_cast<const f32x4*>(anext1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0][t], bcur[0][t], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1][t], bcur[1][t], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef LM_IPROF                             // (tools/build_variant.sh probe: per-phase shader-clock cycles summed over waves)
constexpr int IPROF_WG = 16384;
__device__ unsigned long long g_iprof[IPROF_WG][12];   // [workgroup % IPROF_WG][phase], wave 0 only (plain stores: same-address atomics serialise)   // prologue, transform, barrier 1, MFMA phase, end-of-slab wait, barrier 2, epilogue, waves
#define LM_TICK(slot)                                        \
    {                                                        \
        const long long t_now = clock64();                   \
        iprof[slot] += t_now - t_last;                       \
        t_last = t_now;                                      \
    }
#else
#define LM_TICK(slot)
#endif
#ifdef LM_IPROF_EPI                         // (finer epilogue phases: costs registers, perturbs the main loop)
#define LM_TICKE(slot) LM_TICK(slot)
#else
#define LM_TICKE(slot)
#endif

// =====================================================================================================================================
// DUAL geometry (round 2, third version): TWO workgroups per CU.  With sixteen accumulators per wave (512 registers) a CU holds one
// workgroup, and nothing runs under its transform phases, its prologue and its epilogue (26 % of the wide kernel's cycles, see the
// phase table in DESIGN.md 3.1c).  Here a wave keeps EIGHT xi (128 AGPRs + <= 128 VGPRs = two waves per SIMD): a workgroup is
// 32 tiles x 64 output channels, its four waves are (32-channel half nh) x (xi half xh); 16-channel slabs, 20 KB raw + 32 KB V of LDS.
// While one workgroup transforms, waits at a barrier or stores its outputs, the other one's MFMAs own the SIMDs.
// The fold needs all sixteen products: the xi < 8 waves fold their eight products (the PREFIX of the ascending-xi sum), hand the four
// partial outputs over through LDS, and the xi >= 8 waves continue the same sum with theirs and store - the same sequence of
// additions as everywhere else, bit-identical outputs.
constexpr int DBM = 32, DBN = 64, DKS = 16, DLPW = 5;
constexpr int DNCOL = 2 * DBM + 2 * INSEG;        // 72 column slots
constexpr int DRAW = DLPW * 4 * 256;              // floats of the raw buffer (20 KB; cells 288.. are zero-source padding)
constexpr int DVBUF = 16 * DBM * DKS;             // floats of the V slab (32 KB)
static_assert(4 * DNCOL * DKS <= DRAW && DNCOL % 8 == 0, "dual geometry: patch loads cover the slab");

// this thread's half of the slab transform: tile x channel quad x row pair ih (xi = 8 ih .. 8 ih + 7); arithmetic of wino_slab_transform
__device__ __forceinline__ void wino_slab_transform_half(const float* raw, float* V, const int (&roff)[4], int voff, int ih) {
    constexpr int ROWF = DNCOL * DKS;
    f32x4 ra[4], rb[4];
    if (ih == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(raw + roff[c]);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(raw + ROWF + roff[c]);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(raw + 2 * ROWF + roff[c]);
            ra[c] = pk_sub4(d0, d2);
            rb[c] = pk_add4(d1, d2);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(raw + ROWF + roff[c]);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(raw + 2 * ROWF + roff[c]);
            const f32x4 d3 = *reinterpret_cast<const f32x4*>(raw + 3 * ROWF + roff[c]);
            ra[c] = pk_sub4(d2, d1);
            rb[c] = pk_sub4(d1, d3);
        }
    }
    float* o = V + (8 * ih) * (DBM * DKS) + voff;
    *reinterpret_cast<f32x4*>(o) = pk_sub4(ra[0], ra[2]);
    *reinterpret_cast<f32x4*>(o + DBM * DKS) = pk_add4(ra[1], ra[2]);
    *reinterpret_cast<f32x4*>(o + 2 * DBM * DKS) = pk_sub4(ra[2], ra[1]);
    *reinterpret_cast<f32x4*>(o + 3 * DBM * DKS) = pk_sub4(ra[1], ra[3]);
    *reinterpret_cast<f32x4*>(o + 4 * DBM * DKS) = pk_sub4(rb[0], rb[2]);
    *reinterpret_cast<f32x4*>(o + 5 * DBM * DKS) = pk_add4(rb[1], rb[2]);
    *reinterpret_cast<f32x4*>(o + 6 * DBM * DKS) = pk_sub4(rb[2], rb[1]);
    *reinterpret_cast<f32x4*>(o + 7 * DBM * DKS) = pk_sub4(rb[1], rb[3]);
}

__device__ __forceinline__ float wino_fold_coef(int ab, int xi) {      // (A^T)[a][wi] (A^T)[b][wj], xi = 4 wi + wj
    const int a = ab >> 1, b = ab & 1, wi = xi >> 2, wj = xi & 3;
    const float ca = a == 0 ? (wi < 3 ? 1.f : 0.f) : (wi == 0 ? 0.f : (wi == 1 ? 1.f : -1.f));
    const float cb = b == 0 ? (wj < 3 ? 1.f : 0.f) : (wj == 0 ? 0.f : (wj == 1 ? 1.f : -1.f));
    return ca * cb;
}

__global__ __launch_bounds__(256, 2) void wino_dual_kernel(WinoImpParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // raw patch slab [DRAW] | V slab [DVBUF]
    float* const rawbuf = smem;
    float* const Vbuf = smem + DRAW;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wave & 1, xh = wave >> 1;            // 32-channel half, xi half (waves 0,1: xi 0..7; waves 2,3: xi 8..15)
    const int wn0 = nh * 32;
    const int n_tiles = (p.Cout + DBN - 1) / DBN;
    unsigned mblk, ntile;
    {   // XCD-aware order, N tile outer (see wino_implicit_kernel)
        const unsigned bid = blockIdx.x, mb = gridDim.x / (unsigned)n_tiles, mbx = mb / 8, full = mbx * 8 * (unsigned)n_tiles;
        if (bid < full) {
            const unsigned xcd = bid % 8, idx = bid / 8;
            ntile = idx / mbx;
            mblk = xcd * mbx + idx % mbx;
        } else {
            const unsigned r = bid - full;
            mblk = 8 * mbx + r / (unsigned)n_tiles;
            ntile = r % (unsigned)n_tiles;
        }
    }
    const long m0 = (long)mblk * DBM;
    const int n0 = (int)ntile * DBN;
    const WinoGeom& g = p.g;
    const int bi = (int)(m0 / g.Tpad);
    const int t0 = (int)(m0 - (long)bi * g.Tpad);
    int ts[INSEG + 1], sn[INSEG], iy0[INSEG], ix0[INSEG], oy0[INSEG], ox0[INSEG];      // run table (see wino_implicit_kernel)
    {
        int at = 0, t = t0;
        int tx = t0 % g.Tx, rest = t0 / g.Tx;
        int ty = rest % g.Ty, ph = rest / g.Ty;
        int pa = ph / g.dil, pb = ph - pa * g.dil;
#pragma unroll
        for (int s_ = 0; s_ < INSEG; ++s_) {
            ts[s_] = at;
            const bool real = t < g.Timg && at < DBM;
            const int n = at < DBM ? min(DBM - at, g.Tx - tx) : 0;
            sn[s_] = real ? n : 0;
            iy0[s_] = (2 * ty - 1) * g.dil + pa;
            ix0[s_] = (2 * tx - 1) * g.dil + pb;
            oy0[s_] = 2 * ty * g.dil + pa;
            ox0[s_] = 2 * tx * g.dil + pb;
            at += n;
            t += n;
            tx += n;
            if (tx >= g.Tx) {
                tx = 0;
                if (++ty >= g.Ty) {
                    ty = 0;
                    if (++pb >= g.dil) {
                        pb = 0;
                        ++pa;
                    }
                }
            }
        }
        ts[INSEG] = at;
    }
    const float* gsrc[DLPW];
    const int img_pix0 = bi * g.H * g.W;
#pragma unroll
    for (int s_ = 0; s_ < DLPW; ++s_) {
        const int pos = (s_ * 4 + wave) * 16 + (lane >> 2);           // LDS cell position (16 cells of 64 B per wave load)
        const int r = pos / DNCOL;
        const int q = unrot3(pos - r * DNCOL);
        const int ch = lane & 3;
        int n = sn[0], yb = iy0[0], xb = ix0[0], q0 = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k)
            if (q >= 2 * ts[k] + 2 * k) {
                n = sn[k]; yb = iy0[k]; xb = ix0[k]; q0 = 2 * ts[k] + 2 * k;
            }
        const int lc = q - q0;
        const int yy = yb + r * g.dil, xx = xb + lc * g.dil;
        const bool ok = r < 4 && lc < 2 * n + 2 && n > 0 && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
        gsrc[s_] = ok ? p.x + (long)(img_pix0 + yy * g.W + xx) * p.ldx + ch * 4 : p.zeros + ch * 4;
    }
    // transform task: tile (tid & 127) / 4, channel quad tid & 3, row pair tid / 128 (= xh: a wave transforms the xi it will multiply)
    int roff[4], tvoff;
    {
        const int tl = (tid & 127) >> 2, qd = tid & 3;
        int sg = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k) sg += (ts[k] < DBM && tl >= ts[k]) ? 1 : 0;
        const int cb = 2 * tl + 2 * sg;
#pragma unroll
        for (int c = 0; c < 4; ++c) roff[c] = (rot3(cb + c) * 4 + qd) * 4;
        tvoff = (tl * 4 + (qd ^ ((tl >> 2) & 3))) * 4;
    }
    const int frow = lane & 31, fhalf = lane >> 5;
    int aoff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) aoff[kk] = (frow * 4 + ((2 * kk + fhalf) ^ ((frow >> 2) & 3))) * 4;
    const float* const Vx = Vbuf + (8 * xh) * (DBM * DKS);             // this wave's eight planes
    const int cslabs = p.C / DKS;
    const unsigned bvoff = (unsigned)lane * 16u;
    const long bstep = (long)p.NT * 512;
    const long bxi = (long)cslabs * bstep;
    const float* const bbase = p.U + (long)((n0 + wn0) >> 5) * 512 + (long)(8 * xh) * bxi;

    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f32x4 bq[8][2];
#pragma unroll
    for (int s_ = 0; s_ < DLPW; ++s_)
        __builtin_amdgcn_global_load_lds((gptr_t*)gsrc[s_], (lptr_t*)(rawbuf + (s_ * 4 + wave) * 256), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < BD; ++k) bload2(bq[k], bvoff, bbase + (long)k * bxi);
#pragma unroll
    for (int k = 0; k < BD; ++k) bwait<0>(bq[k]);
    __builtin_amdgcn_s_barrier();
    for (int cs = 0; cs < cslabs; ++cs) {
        wino_slab_transform_half(rawbuf, Vbuf, roff, tvoff, xh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = cs + 1 < cslabs;
        const long goff = more ? (long)(cs + 1) * DKS : 0;
        const float* const bs = bbase + (long)cs * bstep;
        const float* const bs_next = bbase + (long)(more ? cs + 1 : 0) * bstep;
        int gnext = 0;
        f32x4 a0[2], a1[2];
        a0[0] = *reinterpret_cast<const f32x4*>(Vx + aoff[0]);
        a0[1] = *reinterpret_cast<const f32x4*>(Vx + aoff[1]);
        // eight steps (one per xi of this wave) per slab, B seven steps ahead in the ring of 8: a window of eight consecutive steps is
        // one slab period, so the loads younger than a step's B fragments are always 14 B loads + the 5 patch loads: NWAIT = 19
#define LM_DSTEP(K, AC, AN) \
        wino_imp_step<K, ((K) < DLPW ? 1 : 0), 19, ((K) < 7)>(acc[K], bq, bvoff, (K) + BD < 8 ? bs + (long)((K) + BD) * bxi : bs_next + (long)((K) + BD - 8) * bxi, \
                                                             Vx + ((K) + 1) * (DBM * DKS) + aoff[0], Vx + ((K) + 1) * (DBM * DKS) + aoff[1], AC, AN, gsrc, goff, \
                                                             rawbuf, wave, gnext)
        LM_DSTEP(0, a0, a1); LM_DSTEP(1, a1, a0); LM_DSTEP(2, a0, a1); LM_DSTEP(3, a1, a0);
        LM_DSTEP(4, a0, a1); LM_DSTEP(5, a1, a0); LM_DSTEP(6, a0, a1); LM_DSTEP(7, a1, a0);
#undef LM_DSTEP
        bwait<6>(bq[0]);                       // the patch loads (last one in step 4) have landed: only steps 5..7's B loads are younger
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) bwait<0>(bq[k]);

    // --- epilogue.  xi < 8 waves: prefix of the fold -> LDS; xi >= 8 waves: rest of the fold, transposes, stores.
    float* const xchg = smem + nh * 4096;                              // [ab 4][r4 4][lane 64][4] floats per channel half
    constexpr int ELD = 32 + 4;
    float* const stage = smem + 8192 + nh * (32 * ELD);
    static_assert(8192 + 2 * 32 * ELD <= DRAW + DVBUF, "dual epilogue buffers fit");
    __syncthreads();                                   // every wave is done with the patch / V buffers
    if (xh == 0) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) {
                const float c = wino_fold_coef(ab, xi);
                if (c == 0.f) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = fmaf(acc[xi][r], c, o[r]);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[4 * r4], o[4 * r4 + 1], o[4 * r4 + 2], o[4 * r4 + 3]};
                *reinterpret_cast<f32x4*>(xchg + ((ab * 4 + r4) * 64 + lane) * 4) = v;
            }
        }
    }
    __syncthreads();
    if (xh == 0) return;
    constexpr int LPR = 8, RPI = 8, NP = 4;
    const int c4 = (lane & 7) * 4;
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
    f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};
    int pix0[NP];
    unsigned vmask = 0;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int tl = pass * RPI + lane / LPR;
        int nn = sn[0], oy = oy0[0], oxb = ox0[0], tb = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k)
            if (tl >= ts[k]) {
                nn = sn[k]; oy = oy0[k]; oxb = ox0[k]; tb = ts[k];
            }
        const int ox = oxb + 2 * (tl - tb) * g.dil;
        pix0[pass] = img_pix0 + oy * g.W + ox;
        if (nn > 0 && oy < g.H && ox < g.W)
            vmask |= (1u | (oy + g.dil < g.H ? 2u : 0u) | (ox + g.dil < g.W ? 4u : 0u)) << (3 * pass);
    }
    const int step_a = g.dil * g.W, step_b = g.dil;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ab = 2 * a + b;
            // (residual loads of this output position before its fold: see wino_pipe_kernel)
            f32x4 rpre[NP];
            if (vec && p.res && n < p.Cout) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
                    const unsigned vm = vmask >> (3 * pass);
                    const bool ok = (vm & 1u) && (!a || (vm & 2u)) && (!b || (vm & 4u));
                    const long pix = pix0[pass] + a * step_a + b * step_b;
                    rpre[pass] = ok ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            f32x16 o;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xchg + ((ab * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * r4 + e] = v[e];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float c = wino_fold_coef(ab, 8 + k);
                if (c == 0.f) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = fmaf(acc[k][r], c, o[r]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * ELD + frow] = o[r];
            __builtin_amdgcn_wave_barrier();
            if (n >= p.Cout) continue;
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
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
                        const f32x4 rr = rpre[pass];
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
    if (p.gn_part && n < p.Cout) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gs[e] += __shfl_xor(gs[e], o);
                gq[e] += __shfl_xor(gq[e], o);
            }
        if (lane < LPR) {
            const long chunk = t0 / 32;
            double* o = p.gn_part + (((long)bi * (g.Tpad / 32) + chunk) * p.Cout + n) * 2;
            for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                o[2 * e] = (double)gs[e];
                o[2 * e + 1] = (double)gq[e];
            }
        }
    }
}

// =====================================================================================================================================
// PIPE geometry (round 3): the WIDE workgroup (32 tiles x 128 output channels, four waves side by side in N, sixteen xi per wave) with
// the slab transform taken OUT of its own phase and spread over the MFMA steps.  Measured with tools/probes/coissue_probe*.hip
// (profiles/r3_coissue_probe.txt) on v_mfma_f32_32x32x2_f32, one wave per SIMD: an LDS instruction behind an MFMA costs nothing (up
// to one per MFMA; four ds_write_b128 in a row do cost), a VALU instruction 8 cycles, a VMEM instruction 8-9 - and a second wave on
// the SIMD does not change any of that (two MFMA-streaming waves do not interleave, the f32 MFMA holds the SIMD's VALU issue for its
// 64 cycles).  The wide kernel's transform phase is 2.0 k cycles per 32-channel slab, most of it the 64 KB of ds_write_b128 traffic
// (13 cycles of data transfer per wave instruction) and LDS round trips with nothing to overlap; as single instructions between MFMAs
// only its 64 packed adds remain visible.
// Time is cut into slots of one 16-channel half-slab h: 16 steps (one per xi) of 8 MFMAs on V[h & 1], and between them the pieces of
// this thread's share of the transform of half-slab h + 1 (raw[(h + 1) & 1] -> V[(h + 1) & 1]: 12 ds_read_b128, 16 packed FMAs / adds
// x 2, 8 ds_write_b128) and the five patch loads of half-slab h + 2.  One barrier per slot.  Per accumulator the channels still
// arrive in ascending order (half-slab h = channels 16 h ..), V has the bits of the materialising kernel: bit-identical outputs.
// LDS: raw[2] x 20 KB + V[2] x 32 KB = 104 KB; layouts of the DUAL geometry (16-float cells, rot3 columns, V rows XOR-swizzled).
constexpr int PBM = 32, PBN = 128, PKS = 16, PLPW = 5;
constexpr int PNCOL = 2 * PBM + 2 * INSEG;        // 72 column slots
constexpr int PRAWH = PLPW * 4 * 256;             // floats of one raw half-slab buffer (20 KB; cells 288.. are zero-source padding)
constexpr int PVH = 16 * PBM * PKS;               // floats of one V half-slab buffer (32 KB)
static_assert(4 * PNCOL * PKS <= PRAWH && PNCOL % 8 == 0, "pipe geometry: patch loads cover the half-slab");

typedef float f32x2v __attribute__((ext_vector_type(2)));
// r = p + s q on four lanes of a channel quad, s = +-1 (IEEE sum / difference through the FMA unit: the product is exact)
__device__ __forceinline__ f32x4 pk_fma4(const f32x4 q, const f32x2v s, const f32x4 p) {
    const f32x2v qlo = __builtin_shufflevector(q, q, 0, 1), qhi = __builtin_shufflevector(q, q, 2, 3);
    const f32x2v plo = __builtin_shufflevector(p, p, 0, 1), phi = __builtin_shufflevector(p, p, 2, 3);
    f32x2v rlo, rhi;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(rlo) : "v"(qlo), "v"(s), "v"(plo));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(rhi) : "v"(qhi), "v"(s), "v"(phi));
    return __builtin_shufflevector(rlo, rhi, 0, 1, 2, 3);
}

// This thread's share of a half-slab transform = (tile, channel quad, row pair ih): rows 2 ih and 2 ih + 1 of B^T d for its four
// patch columns, then the column pass -> planes xi = 8 ih .. 8 ih + 7.  Row pair 0: ra = d0 - d2, rb = d1 + d2; row pair 1: ra = d2 - d1,
// rb = d1 - d3: three patch rows each, the per-wave row offsets and the sign of rb's second operand spell that out (wino_pipe_kernel).
// PIECE 0..3: read column c; 4..7: row pass of column c - 4; 8..15: compute output plane PIECE - 8; 16..23: store plane PIECE - 16.
struct PipeXf {
    f32x4 d[2][3];         // two columns in flight (the reads of column c + 1 are issued before the row pass of column c): patch rows o0, o1, o2
    f32x4 ra[4], rb[4];
    f32x4 o;               // the output plane computed in the previous step, stored in this one
};
template <int PIECE>
__device__ __forceinline__ void wino_pipe_xf(PipeXf& t, const float* raw, float* V, const int (&roff)[4], int o0, int o1, int o2,
                                             f32x2v sb) {
    const f32x2v neg = {-1.f, -1.f};
    if constexpr (PIECE < 4) {
#ifndef LM_PABL_NOXLDS
        t.d[PIECE & 1][0] = *reinterpret_cast<const f32x4*>(raw + o0 + roff[PIECE]);
        t.d[PIECE & 1][1] = *reinterpret_cast<const f32x4*>(raw + o1 + roff[PIECE]);
        t.d[PIECE & 1][2] = *reinterpret_cast<const f32x4*>(raw + o2 + roff[PIECE]);
#endif
    } else if constexpr (PIECE < 8) {
        // row pair 0: (o0, o1, o2) = rows (0, 2, 1): ra = d0 - d2, rb = d1 + d2 = d[2] + (+1) d[1]
        // row pair 1: (o0, o1, o2) = rows (2, 1, 3): ra = d2 - d1, rb = d1 - d3 = d[1] + (-1) d[2]  -> the operands of rb swap with ih (o1 / o2 below)
        const f32x4 (&d)[3] = t.d[PIECE & 1];
#ifdef LM_PABL_NOXVALU
        t.ra[PIECE - 4] = d[0];
        t.rb[PIECE - 4] = d[2];
#else
        t.ra[PIECE - 4] = pk_fma4(d[1], neg, d[0]);
        t.rb[PIECE - 4] = pk_fma4(d[2], sb, d[1]);
#endif
    } else if constexpr (PIECE < 16) {
        constexpr int k = PIECE - 8, j = k & 3;
        const f32x4 (&r)[4] = k < 4 ? t.ra : t.rb;
#ifdef LM_PABL_NOXVALU
        t.o = r[j];
#else
        t.o = j == 0 ? pk_sub4(r[0], r[2]) : j == 1 ? pk_add4(r[1], r[2]) : j == 2 ? pk_sub4(r[2], r[1]) : pk_sub4(r[1], r[3]);
#endif
    } else {               // PIECE 16..23: store output plane PIECE - 16 (computed one step earlier: the store must not wait for this step's VALU work)
#ifndef LM_PABL_NOXLDS
        *reinterpret_cast<f32x4*>(V + (PIECE - 16) * (PBM * PKS)) = t.o;
#else
        asm volatile("" :: "v"(t.o));
#endif
    }
}

// One step (xi = K) of a slot: 8 MFMAs; B ring of 8 register sets, BD = 7 steps ahead; the slot's PLPW patch loads go out in steps
// 0 .. PLPW-1; NWAIT = 14 B loads + the patch loads of the last eight steps.  XW / XR / XV = transform pieces issued behind the
// first MFMA (-1: none), in this order: XW the ds_write of the plane computed a step ago, the A-fragment reads of the next step, XR
// the ds_reads of the next patch column, XV the VALU piece on data read a step ago - LAST, because VALU work of this wave only starts
// when the MFMA before it has left the pipe, and every LDS instruction behind it would wait as well (measured: store right behind
// its adds 906 cycles per slot, adds and LDS traffic separately 175 + 225).
template <int K, int NWAIT, int XW, int XR, int XV>
__device__ __forceinline__ void wino_pipe_step(f32x16& acc, f32x4 (&bq)[8][2], unsigned bvoff, const float* bpre, const float* anext0,
                                               const float* anext1, f32x4 (&a_cur)[2], f32x4 (&a_nxt)[2], const float* const (&gsrc)[PLPW],
                                               long goff, float* rawld, int wave, PipeXf& xf, const float* xraw, float* xV,
                                               const int (&roff)[4], int o0, int o1, int o2, f32x2v sb) {
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
#ifndef LM_IABL_NOB
    bload2(bq[(K + BD) & 7], bvoff, bpre);
#endif
#ifndef LM_IABL_NOGLDS
    if constexpr (K < PLPW)
        __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[K] + goff), (lptr_t*)(rawld + (K * 4 + wave) * 256), 16, 0, 0);
#endif
    f32x4 (&bcur)[2] = bq[K & 7];
#if !defined(LM_IABL_NOB) && !defined(LM_IABL_NOGLDS)
    bwait<NWAIT>(bcur);
#else
    bwait<0>(bcur);
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0][0], bcur[0][0], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#ifndef LM_PABL_NOT
    if constexpr (XW >= 16) wino_pipe_xf<XW>(xf, xraw, xV, roff, o0, o1, o2, sb);
#endif
    if constexpr (K < 15) {
        a_nxt[0] = *reinterpret_cast<const f32x4*>(anext0);
        a_nxt[1] = *reinterpret_cast<const f32x4*>(anext1);
    }
#ifndef LM_PABL_NOT
    if constexpr (XR >= 0 && XR < 4) wino_pipe_xf<XR>(xf, xraw, xV, roff, o0, o1, o2, sb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XV >= 4 && XV < 16) wino_pipe_xf<XV>(xf, xraw, xV, roff, o0, o1, o2, sb);
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0][t], bcur[0][t], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1][t], bcur[1][t], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(256) void wino_pipe_kernel(WinoImpParams p) {
#ifdef LM_IPROF
    long long iprof[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];      // raw[2][PRAWH] | V[2][PVH]
    float* const raw0 = smem;
    float* const V0 = smem + 2 * PRAWH;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn0 = wave * 32;
    const int n_tiles = (p.Cout + PBN - 1) / PBN;
    unsigned mblk, ntile;
    {   // XCD-aware order, N tile outer (see wino_implicit_kernel)
        const unsigned bid = blockIdx.x, mb = gridDim.x / (unsigned)n_tiles, mbx = mb / 8, full = mbx * 8 * (unsigned)n_tiles;
        if (bid < full) {
            const unsigned xcd = bid % 8, idx = bid / 8;
            ntile = idx / mbx;
            mblk = xcd * mbx + idx % mbx;
        } else {
            const unsigned r = bid - full;
            mblk = 8 * mbx + r / (unsigned)n_tiles;
            ntile = r % (unsigned)n_tiles;
        }
    }
    const long m0 = (long)mblk * PBM;
    const int n0 = (int)ntile * PBN;
    const WinoGeom& g = p.g;
    const int bi = (int)(m0 / g.Tpad);
    const int t0 = (int)(m0 - (long)bi * g.Tpad);
    int ts[INSEG + 1], sn[INSEG], iy0[INSEG], ix0[INSEG], oy0[INSEG], ox0[INSEG];      // run table (see wino_implicit_kernel)
    {
        int at = 0, t = t0;
        int tx = t0 % g.Tx, rest = t0 / g.Tx;
        int ty = rest % g.Ty, ph = rest / g.Ty;
        int pa = ph / g.dil, pb = ph - pa * g.dil;
#pragma unroll
        for (int s_ = 0; s_ < INSEG; ++s_) {
            ts[s_] = at;
            const bool real = t < g.Timg && at < PBM;
            const int n = at < PBM ? min(PBM - at, g.Tx - tx) : 0;
            sn[s_] = real ? n : 0;
            iy0[s_] = (2 * ty - 1) * g.dil + pa;
            ix0[s_] = (2 * tx - 1) * g.dil + pb;
            oy0[s_] = 2 * ty * g.dil + pa;
            ox0[s_] = 2 * tx * g.dil + pb;
            at += n;
            t += n;
            tx += n;
            if (tx >= g.Tx) {
                tx = 0;
                if (++ty >= g.Ty) {
                    ty = 0;
                    if (++pb >= g.dil) {
                        pb = 0;
                        ++pa;
                    }
                }
            }
        }
        ts[INSEG] = at;
    }
    const float* gsrc[PLPW];
    const int img_pix0 = bi * g.H * g.W;
#pragma unroll
    for (int s_ = 0; s_ < PLPW; ++s_) {
        const int pos = (s_ * 4 + wave) * 16 + (lane >> 2);           // LDS cell position (16 cells of 64 B per wave load)
        const int r = pos / PNCOL;
        const int q = unrot3(pos - r * PNCOL);
        const int ch = lane & 3;
        int n = sn[0], yb = iy0[0], xb = ix0[0], q0 = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k)
            if (q >= 2 * ts[k] + 2 * k) {
                n = sn[k]; yb = iy0[k]; xb = ix0[k]; q0 = 2 * ts[k] + 2 * k;
            }
        const int lc = q - q0;
        const int yy = yb + r * g.dil, xx = xb + lc * g.dil;
        const bool ok = r < 4 && lc < 2 * n + 2 && n > 0 && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
        gsrc[s_] = ok ? p.x + (long)(img_pix0 + yy * g.W + xx) * p.ldx + ch * 4 : p.zeros + ch * 4;
    }
    // transform share: tile (tid & 127) >> 2, channel quad tid & 3, row pair ih = wave >> 1 (uniform per wave: the row offsets and the
    // sign below are scalars)
    int roff[4], tvoff, xo0, xo1, xo2;
    f32x2v xsb;
    {
        const int ih = wave >> 1;
        const int tl = (tid & 127) >> 2, qd = tid & 3;
        int sg = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k) sg += (ts[k] < PBM && tl >= ts[k]) ? 1 : 0;
        const int cb = 2 * tl + 2 * sg;
#pragma unroll
        for (int c = 0; c < 4; ++c) roff[c] = (rot3(cb + c) * 4 + qd) * 4;
        tvoff = (8 * ih) * (PBM * PKS) + (tl * 4 + (qd ^ ((tl >> 2) & 3))) * 4;
        constexpr int ROWF = PNCOL * PKS;
        // ih = 0: d[] = rows (0, 2, 1): ra = d[0] - d[1] = d0 - d2, rb = d[1] + d[2]... see wino_pipe_xf: ra = d[0] - d[1], rb = d[1] + sb d[2]
        //         rb = d1 + d2 is commutative: rows (0, 2, 1) give d[1] + d[2] = d2 + d1 - NOT the same bits as d1 + d2?  They are: IEEE
        //         addition is commutative.  ih = 1: d[] = rows (2, 1, 3): ra = d2 - d1, rb = d1 - d3.
        xo0 = (ih == 0 ? 0 : 2) * ROWF;
        xo1 = (ih == 0 ? 2 : 1) * ROWF;
        xo2 = (ih == 0 ? 1 : 3) * ROWF;
        const float sbv = ih == 0 ? 1.f : -1.f;
        xsb = f32x2v{sbv, sbv};
    }
    const int frow = lane & 31, fhalf = lane >> 5;
    int aoff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) aoff[kk] = (frow * 4 + ((2 * kk + fhalf) ^ ((frow >> 2) & 3))) * 4;
    const int H = p.C / PKS;                           // half-slabs (slots)
    const unsigned bvoff = (unsigned)lane * 16u;
    const long bstep = (long)p.NT * 512;
    const long bxi = (long)H * bstep;
    const float* const bbase = p.U + (long)((n0 + wn0) >> 5) * 512;

    f32x16 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    LM_TICK(7)
    f32x4 bq[8][2];
    PipeXf xf;
    // prologue: raw half-slabs 0 and 1, B of steps 0 .. BD-1 of slot 0, V of half-slab 0
#pragma unroll
    for (int s_ = 0; s_ < PLPW; ++s_)
        __builtin_amdgcn_global_load_lds((gptr_t*)gsrc[s_], (lptr_t*)(raw0 + (s_ * 4 + wave) * 256), 16, 0, 0);
#pragma unroll
    for (int s_ = 0; s_ < PLPW; ++s_)
        __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[s_] + PKS), (lptr_t*)(raw0 + PRAWH + (s_ * 4 + wave) * 256), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < BD; ++k) bload2(bq[k], bvoff, bbase + (long)k * bxi);
    // only raw half-slab 0 has to be there for T(0): its PLPW loads are the oldest of the 2 PLPW + 2 BD in flight.  Raw half-slab 1 is
    // waited for behind T(0) (slot 0 reads it), the B fragments by the steps that use them.
    bwait<PLPW + 2 * BD>(bq[0]);
    __builtin_amdgcn_s_barrier();
    LM_TICK(0)
#define LM_PXF(P) wino_pipe_xf<P>(xf, raw0, V0 + tvoff, roff, xo0, xo1, xo2, xsb)
    LM_PXF(0); LM_PXF(4); LM_PXF(1); LM_PXF(5); LM_PXF(2); LM_PXF(6); LM_PXF(3); LM_PXF(7);
    LM_PXF(8); LM_PXF(16); LM_PXF(9); LM_PXF(17); LM_PXF(10); LM_PXF(18); LM_PXF(11); LM_PXF(19);
    LM_PXF(12); LM_PXF(20); LM_PXF(13); LM_PXF(21); LM_PXF(14); LM_PXF(22); LM_PXF(15); LM_PXF(23);
#undef LM_PXF
    bwait<2 * BD>(bq[0]);                      // raw half-slab 1 has landed (this wave's part; the barrier below collects all parts)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LM_TICK(1)
    __builtin_amdgcn_s_barrier();
    LM_TICK(2)
    // one slot, PAR = h & 1 a compile-time constant (every LDS address of the slot is then a register + immediate): H is even
#define LM_PSTEP(PAR, K, NW, XW, XR, XV, AC, AN) \
        wino_pipe_step<K, NW, XW, XR, XV>(acc[K], bq, bvoff, (K) + BD < 16 ? bs + (long)((K) + BD) * bxi : bs_next + (long)((K) + BD - 16) * bxi, \
                                          V0 + (PAR) * PVH + ((K) + 1) * (PBM * PKS) + aoff[0], V0 + (PAR) * PVH + ((K) + 1) * (PBM * PKS) + aoff[1], AC, AN, \
                                          gsrc, goff, raw0 + (PAR) * PRAWH, wave, xf, raw0 + ((PAR) ^ 1) * PRAWH, V0 + ((PAR) ^ 1) * PVH + tvoff, roff, xo0, xo1, xo2, xsb)
    // NWAIT(K) = 2 BD + |{K-7 .. K} (mod 16) intersected with the patch-load steps {0 .. 4}|.  Transform pieces: column c read in step
    // c, its row pass in step c + 1 (pieces 4..7), output plane k computed in step 5 + k (8..15) and stored in step 6 + k (16..23)
#define LM_PSLOT(PAR, HH)                                                                                                        \
    {                                                                                                                            \
        const int hh = (HH);                                                                                                     \
        const long goff = hh + 2 < H ? (long)(hh + 2) * PKS : 0;            /* (nothing left to fetch: harmless re-read) */       \
        const float* const bs = bbase + (long)hh * bstep;                                                                        \
        const float* const bs_next = bbase + (long)(hh + 1 < H ? hh + 1 : 0) * bstep;                                            \
        f32x4 a0[2], a1[2];                                                                                                      \
        a0[0] = *reinterpret_cast<const f32x4*>(V0 + (PAR) * PVH + aoff[0]);                                                     \
        a0[1] = *reinterpret_cast<const f32x4*>(V0 + (PAR) * PVH + aoff[1]);                                                     \
        LM_PSTEP(PAR, 0, 15, -1, 0, -1, a0, a1);  LM_PSTEP(PAR, 1, 16, -1, 1, 4, a1, a0);   LM_PSTEP(PAR, 2, 17, -1, 2, 5, a0, a1);   \
        LM_PSTEP(PAR, 3, 18, -1, 3, 6, a1, a0);   LM_PSTEP(PAR, 4, 19, -1, -1, 7, a0, a1);  LM_PSTEP(PAR, 5, 19, -1, -1, 8, a1, a0);  \
        LM_PSTEP(PAR, 6, 19, 16, -1, 9, a0, a1);  LM_PSTEP(PAR, 7, 19, 17, -1, 10, a1, a0); LM_PSTEP(PAR, 8, 18, 18, -1, 11, a0, a1); \
        LM_PSTEP(PAR, 9, 17, 19, -1, 12, a1, a0); LM_PSTEP(PAR, 10, 16, 20, -1, 13, a0, a1); LM_PSTEP(PAR, 11, 15, 21, -1, 14, a1, a0); \
        LM_PSTEP(PAR, 12, 14, 22, -1, 15, a0, a1); LM_PSTEP(PAR, 13, 14, 23, -1, -1, a1, a0); LM_PSTEP(PAR, 14, 14, -1, -1, -1, a0, a1); \
        LM_PSTEP(PAR, 15, 14, -1, -1, -1, a1, a0);                                                                               \
        LM_TICK(3)                                                                                                               \
        bwait<14>(bq[0]);                      /* this wave's patch loads (steps 0 .. 4) have landed */                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        LM_TICK(4)                                                                                                               \
        __builtin_amdgcn_s_barrier();          /* V(h + 1) complete, V(h) and raw(h + 1) free, raw(h + 2) landed */              \
        LM_TICK(5)                                                                                                               \
    }
    static_assert(PLPW == 5 && BD == 7, "NWAIT table above");
    for (int h = 0; h < H; h += 2) {
        LM_PSLOT(0, h)
        LM_PSLOT(1, h + 1)
    }
#undef LM_PSLOT
#undef LM_PSTEP
#pragma unroll
    for (int k = 0; k < 8; ++k) bwait<0>(bq[k]);

#include "wino_pipe_epilogue.h"
#ifdef LM_IPROF
    LM_TICK(6)
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 11; ++k) g_iprof[blockIdx.x % IPROF_WG][k] = (unsigned long long)iprof[k];
        g_iprof[blockIdx.x % IPROF_WG][11] = 1ull;
    }
#endif
}

// =====================================================================================================================================
// ROWS geometry of the split-precision GEMM: 64 tiles x 64 channels per workgroup, wave w owns the four Winograd products of ROW w of the
// 4 x 4 transform (xi = 4w .. 4w + 3) for the whole block - 4 xi x 2 tile halves x 2 channel halves = 16 accumulator blocks.
// Why: the same GEMM in the PIPE geometry (32 x 128, V pieces in LDS; measured and removed) pulled 192 KB of B pieces per slot and CU
// = 62 B/clk at full matrix speed; the vector memory path sustains ~35-40 of its 64 B/clk with every CU pulling, so it ran at the speed
// of its B traffic (2.43 ms on 256->256@288^2 B = 8, matrix time 0.95; this kernel: 2.13, the fp32 PIPE kernel: 2.97).
// Here every B fragment is loaded by exactly ONE wave and used for both tile halves (32 B/clk), and the transform of a xi row
// needs nothing from the other waves: V[w][j] comes from two patch rows (row pass), a column pass and the split, all in the registers of
// the lane that feeds it to the MFMA (lane (r, g) = tile r, channels 4g..4g+3 and 8+4g..8+4g+3) - no V in LDS, no barrier but the one that
// publishes the patch buffer of the slot after next.  The 480 VALU instructions per slot and wave ride behind the 96 MFMAs (four to five
// plain fp32 VALU per bf16 MFMA issue for free - tools/probes/split_probe.hip).  Patch layout in LDS: 8 planes (patch row, column
// parity) of 68 entries of 64 bytes (one pixel's 16-channel slab, fetched by four adjacent lanes of a global_load_lds; quad q at
// position q ^ ((entry >> 2) & 1)): the 32 lanes of a tile half read consecutive entries, two-way bank conflicts at most.  Four
// patch buffers in rotation (slot h fills the buffer of slot h + 3; the end-of-slot wait is for the loads of ONE SLOT AGO).
// The 2 x 2 outputs need all four rows: the epilogue folds a wave's four xi (column pass), exchanges the row terms through LDS and the
// wave that owns a (tile half, channel half) block combines them in ascending row order and stores (wino_rows_epilogue.h).  DESIGN 3.1e.
constexpr int RBM = 64, RBN = 64, RKS = 16, RLPW = 9;
constexpr int RPL = RBM + INSEG;                  // 64-byte entries of one patch plane (the columns of one parity)
constexpr int RRAWH = RLPW * 4 * 256;             // floats of one patch buffer (36 KB)
constexpr int RELD = 32 + 4;
constexpr int RXCH = 16 * 32 * RELD;              // floats of ONE output column's terms in the epilogue's exchange area (72 KB; two columns)
static_assert(8 * RPL * 4 <= RLPW * 4 * 64, "rows geometry: patch loads cover the planes");

struct RowsR {
    float v[4][8];                                // row-pass result of one tile: [patch column][channel of the lane's 8]
};

// row pass of ROW w for one tile half: r[c] = d[first row][c] + sgn * d[second row][c], 16 ds_read_b128 (two-way bank conflicts: the
// swizzle of the quad position covers half of the 16-lane groups' 64-byte stride)
__device__ __forceinline__ void wino_rows_rowpass(const float* raw, const int (&b1)[2], const int (&b2)[2], float sgn, RowsR& r) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const int off = (c & 1) * (RPL * 16) + qs * 8;
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(raw + b1[c >> 1] + off);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(raw + b2[c >> 1] + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) r.v[c][qs * 4 + e] = fmaf(d2[e], sgn, d1[e]);
        }
}

// column pass J of the row + split into the three A pieces (52 VALU) - the prologue's version
template <int J>
__device__ __forceinline__ void wino_rows_make_a(const RowsR& r, f32x4 (&a)[3]) {
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        lo[e] = J == 0 ? r.v[0][e] - r.v[2][e] : J == 1 ? r.v[1][e] + r.v[2][e] : J == 2 ? r.v[2][e] - r.v[1][e] : r.v[1][e] - r.v[3][e];
        hi[e] = J == 0 ? r.v[0][4 + e] - r.v[2][4 + e] : J == 1 ? r.v[1][4 + e] + r.v[2][4 + e] : J == 2 ? r.v[2][4 + e] - r.v[1][4 + e]
                                                                                                   : r.v[1][4 + e] - r.v[3][4 + e];
    }
    split3_frag(lo, hi, a);
}

// The same work cut into the twelve pieces that ride behind the twelve MFMAs of a sub-step: piece K makes part K % 3 of dword K / 3
// (values 2d, 2d + 1 of the lane's eight) - column pass + leading piece (5 VALU), second piece (5), third piece (3).
struct RowsSplitSt {
    float x, y;
};
__device__ __forceinline__ float wino_low16(float v) { return v - __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); }      // exact
template <int J, int K>
__device__ __forceinline__ void wino_rows_chunk(const RowsR& r, f32x4 (&a)[3], RowsSplitSt& st) {
    constexpr int d = K / 3, part = K % 3, e0 = 2 * d, e1 = 2 * d + 1;
    if constexpr (part == 0) {
        const float v0 = J == 0 ? r.v[0][e0] - r.v[2][e0] : J == 1 ? r.v[1][e0] + r.v[2][e0] : J == 2 ? r.v[2][e0] - r.v[1][e0] : r.v[1][e0] - r.v[3][e0];
        const float v1 = J == 0 ? r.v[0][e1] - r.v[2][e1] : J == 1 ? r.v[1][e1] + r.v[2][e1] : J == 2 ? r.v[2][e1] - r.v[1][e1] : r.v[1][e1] - r.v[3][e1];
        a[0][d] = __uint_as_float(pack_hi(v0, v1));
        st.x = wino_low16(v0);
        st.y = v1;
    } else if constexpr (part == 1) {
        const float r1 = wino_low16(st.y);
        a[1][d] = __uint_as_float(pack_hi(st.x, r1));
        st.x = wino_low16(st.x);
        st.y = r1;
    } else {
        a[2][d] = __uint_as_float(pack_hi(st.x, wino_low16(st.y)));
    }
}

// One patch column of the row pass (r[C] = d[first row][C] + sgn * d[second row][C], 8 channels), spread over a sub-step: the four
// ds_read_b128 behind MFMAs 2 and 5 (short split pieces), the eight fma behind MFMAs 8 .. 11 - with the loads in the first half of the
// sub-step every MFMA has at most six instructions behind it.
struct RowsColT {
    f32x4 d1[2], d2[2];
};
template <int K, int C>
__device__ __forceinline__ void wino_rows_col(const float* raw, int b1, int b2, float sgn, RowsR& r, RowsColT& t) {
    constexpr int off = (C & 1) * (RPL * 16);
    if constexpr (K == 2) {
        t.d1[0] = *reinterpret_cast<const f32x4*>(raw + b1 + off);
        t.d2[0] = *reinterpret_cast<const f32x4*>(raw + b2 + off);
    } else if constexpr (K == 5) {
        t.d1[1] = *reinterpret_cast<const f32x4*>(raw + b1 + off + 8);
        t.d2[1] = *reinterpret_cast<const f32x4*>(raw + b2 + off + 8);
    } else if constexpr (K == 8) {
#pragma unroll
        for (int e = 0; e < 3; ++e) r.v[C][e] = fmaf(t.d2[0][e], sgn, t.d1[0][e]);
    } else if constexpr (K == 9) {
        r.v[C][3] = fmaf(t.d2[0][3], sgn, t.d1[0][3]);
    } else if constexpr (K == 10) {
        r.v[C][4] = fmaf(t.d2[1][0], sgn, t.d1[1][0]);
    } else if constexpr (K == 11) {
#pragma unroll
        for (int e = 1; e < 4; ++e) r.v[C][4 + e] = fmaf(t.d2[1][e], sgn, t.d1[1][e]);
    }
}

template <int OFF>
__device__ __forceinline__ void bload1(f32x4& b, unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(b) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void bwait6(f32x4 (&b)[2][3]) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]) : "n"(N) : "memory");
}

// MFMA K of a sub-step: piece product K / 2 (smallest first: v3 u1, v2 u2, v1 u3, v2 u1, v1 u2, v1 u1), channel half K % 2
template <int K>
__device__ __forceinline__ void wino_rows_mfma(f32x16& acc0, f32x16& acc1, const f32x4 (&a)[3], const f32x4 (&b)[2][3]) {
    constexpr int PP = K >> 1, NS = K & 1;
    constexpr int AI = PP == 0 ? 2 : (PP == 1 || PP == 3) ? 1 : 0;
    constexpr int BI = (PP == 0 || PP == 3 || PP == 5) ? 0 : (PP == 1 || PP == 4) ? 1 : 2;
    if constexpr (NS == 0)
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[AI]), __builtin_bit_cast(bf16x8, b[0][BI]), acc0, 0, 0, 0);
    else
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[AI]), __builtin_bit_cast(bf16x8, b[1][BI]), acc1, 0, 0, 0);
}

// Sub-step (J, MS): the twelve MFMAs of (xi 4w + J, tile half MS) and, one piece behind each: the A pieces of the NEXT sub-step
// ((J, 1) after (J, 0); (J + 1, 0) of this or the next slot after (J, 1)), one patch column of the row pass, and at most one vector
// memory instruction (a run of them holds the wave's issue for 16+ cycles each - the MFMA pipe drains): the six B loads of step J + 3
// behind MFMAs 0 .. 5 of (J, 0), three patch loads behind MFMAs 0, 3, 6 of (J, 1), J < 3.
// Row-pass columns: (0, ms) column 0, (2, ms) column 2, (3, ms) column 1 of the NEXT slot's patches, (1, ms) column 3 of THIS slot's -
// each right after the last column pass that read the old value (J = 0 reads columns 0 2, J = 1: 1 2, J = 2: 2 1, J = 3: 1 3).
template <int J, int MS, int K>
__device__ __forceinline__ void wino_rows_gap(f32x16 (&acc)[16], f32x4 (&bq)[4][2][3], const f32x4 (&acur)[3], f32x4 (&anxt)[3], RowsR (&rr)[2],
                                              RowsSplitSt& st, RowsColT& ct, const float* rawcol, int b1, int b2, float rsgn, unsigned bvoff,
                                              const float* bnext, const float* const (&gsrc)[RLPW], long goff, float* rawld, int wave) {
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    constexpr int JN = MS == 0 ? J : (J + 1) & 3;
    constexpr int C = J == 0 ? 0 : J == 1 ? 3 : J == 2 ? 2 : 1;
    wino_rows_mfma<K>(acc[4 * J + 2 * MS], acc[4 * J + 2 * MS + 1], acur, bq[J & 3]);
    wino_rows_chunk<JN, K>(rr[MS ^ 1], anxt, st);
    wino_rows_col<K, C>(rawcol, b1, b2, rsgn, rr[MS], ct);
    if constexpr (MS == 0 && K < 6) {
        constexpr int ns = K / 3, piece = K % 3;
        bload1<piece * 1024>(bq[(J + 3) & 3][ns][piece], bvoff, bnext + ns * 768);
    }
    if constexpr (MS == 1 && J < 3 && (K == 0 || K == 3 || K == 6)) {
        constexpr int s_ = 3 * J + K / 3;
        __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[s_] + goff), (lptr_t*)(rawld + (s_ * 4 + wave) * 256), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int J, int MS>
__device__ __forceinline__ void wino_rows_substep(f32x16 (&acc)[16], f32x4 (&bq)[4][2][3], const f32x4 (&acur)[3], f32x4 (&anxt)[3], RowsR (&rr)[2],
                                                  const float* rawcol, const int (&rb1)[2][2], const int (&rb2)[2][2], float rsgn, unsigned bvoff,
                                                  const float* bnext, const float* const (&gsrc)[RLPW], long goff, float* rawld, int wave) {
    constexpr int C = J == 0 ? 0 : J == 1 ? 3 : J == 2 ? 2 : 1;
    const int b1 = rb1[MS][C >> 1], b2 = rb2[MS][C >> 1];
    RowsSplitSt st;
    RowsColT ct;
#define LM_G(K) wino_rows_gap<J, MS, K>(acc, bq, acur, anxt, rr, st, ct, rawcol, b1, b2, rsgn, bvoff, bnext, gsrc, goff, rawld, wave);
    LM_G(0) LM_G(1) LM_G(2) LM_G(3) LM_G(4) LM_G(5) LM_G(6) LM_G(7) LM_G(8) LM_G(9) LM_G(10) LM_G(11)
#undef LM_G
}

__global__ __launch_bounds__(256) void wino_rows_split_kernel(WinoImpParams p) {
#ifdef LM_IPROF
    long long iprof[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];      // patches[4][RRAWH]; the epilogue's exchange area afterwards
    float* const raw0 = smem;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (p.Cout + RBN - 1) / RBN;
    unsigned mblk, ntile;
    {   // XCD-aware order, N tile outer (see wino_pipe_kernel)
        const unsigned bid = blockIdx.x, mb = gridDim.x / (unsigned)n_tiles, mbx = mb / 8, full = mbx * 8 * (unsigned)n_tiles;
        if (bid < full) {
            const unsigned xcd = bid % 8, idx = bid / 8;
            if (p.n_inner) {        // the N tiles of a tile block back to back on one XCD: its patches are fetched from HBM once
                ntile = idx % (unsigned)n_tiles;
                mblk = xcd * mbx + idx / (unsigned)n_tiles;
            } else {                // N tile outer: one N tile's B pieces stay in the XCD's L2
                ntile = idx / mbx;
                mblk = xcd * mbx + idx % mbx;
            }
        } else {
            const unsigned r = bid - full;
            mblk = 8 * mbx + r / (unsigned)n_tiles;
            ntile = r % (unsigned)n_tiles;
        }
    }
    const long m0 = (long)mblk * RBM;
    const int n0 = (int)ntile * RBN;
    const WinoGeom& g = p.g;
    const int bi = (int)(m0 / g.Tpad);
    const int t0 = (int)(m0 - (long)bi * g.Tpad);
    int ts[INSEG + 1], sn[INSEG], iy0[INSEG], ix0[INSEG], oy0[INSEG], ox0[INSEG];      // run table (see wino_implicit_kernel)
    {
        int at = 0, t = t0;
        int tx = t0 % g.Tx, rest = t0 / g.Tx;
        int ty = rest % g.Ty, ph = rest / g.Ty;
        int pa = ph / g.dil, pb = ph - pa * g.dil;
#pragma unroll
        for (int s_ = 0; s_ < INSEG; ++s_) {
            ts[s_] = at;
            const bool real = t < g.Timg && at < RBM;
            const int n = at < RBM ? min(RBM - at, g.Tx - tx) : 0;
            sn[s_] = real ? n : 0;
            iy0[s_] = (2 * ty - 1) * g.dil + pa;
            ix0[s_] = (2 * tx - 1) * g.dil + pb;
            oy0[s_] = 2 * ty * g.dil + pa;
            ox0[s_] = 2 * tx * g.dil + pb;
            at += n;
            t += n;
            tx += n;
            if (tx >= g.Tx) {
                tx = 0;
                if (++ty >= g.Ty) {
                    ty = 0;
                    if (++pb >= g.dil) {
                        pb = 0;
                        ++pa;
                    }
                }
            }
        }
        ts[INSEG] = at;
    }
    // patch loads: LDS 16-byte unit (load * 4 + wave) * 64 + lane = entry * 4 + position, entry = plane * RPL + idx, plane = row * 2 + column
    // parity; the four lanes of an entry fetch the 64 contiguous bytes of one pixel's slab, quad q at position q ^ ((idx >> 2) & 1)
    const float* gsrc[RLPW];
    const int img_pix0 = bi * g.H * g.W;
#pragma unroll
    for (int s_ = 0; s_ < RLPW; ++s_) {
        const int ent4 = (s_ * 4 + wave) * 64 + lane;
        const int ent = ent4 >> 2;
        const int plane = ent / RPL, idx = ent - plane * RPL;
        const int r = plane >> 1, ch = (ent4 & 3) ^ ((idx >> 2) & 1);
        const int q = 2 * idx + (plane & 1);
        int n = sn[0], yb = iy0[0], xb = ix0[0], q0 = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k)
            if (q >= 2 * ts[k] + 2 * k) {
                n = sn[k]; yb = iy0[k]; xb = ix0[k]; q0 = 2 * ts[k] + 2 * k;
            }
        const int lc = q - q0;
        const int yy = yb + r * g.dil, xx = xb + lc * g.dil;
        const bool ok = plane < 8 && lc < 2 * n + 2 && n > 0 && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
        gsrc[s_] = ok ? p.x + (long)(img_pix0 + yy * g.W + xx) * p.ldx + ch * 4 : p.zeros + ch * 4;
    }
    // row-pass reads of this lane: tile ms * 32 + (lane & 31), quads g and g + 2 (g = lane >> 5); rows of ROW w: first + sgn * second
    const int frow = lane & 31, fhalf = lane >> 5;
    const int row1 = wave == 0 ? 0 : wave == 2 ? 2 : 1;
    const int row2 = wave == 2 ? 1 : wave == 3 ? 3 : 2;
    const float rsgn = wave == 1 ? 1.f : -1.f;
    int rb1[2][2], rb2[2][2];                          // [tile half][patch column pair]
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int tl = ms * 32 + frow;
        int sg = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k) sg += (ts[k] < RBM && tl >= ts[k]) ? 1 : 0;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int idx = tl + sg + cp;                               // entry idx = (2 tl + 2 sg + c) / 2, c = 2 cp + parity
            const int e0 = idx * 16 + (fhalf ^ ((idx >> 2) & 1)) * 4;
            rb1[ms][cp] = row1 * 2 * (RPL * 16) + e0;
            rb2[ms][cp] = row2 * 2 * (RPL * 16) + e0;
        }
    }
    const int H = p.C / RKS;                           // slots (16 input channels each)
    const unsigned bvoff = (unsigned)lane * 16u;
    const long bstep = (long)p.NT * 768;               // floats between consecutive 16-channel slabs of one xi (3 pieces x 64 lanes x 16 B)
    const long bxi = (long)H * bstep;
    const float* const bbase = p.U + (long)(n0 >> 5) * 768 + (long)(4 * wave) * bxi;      // xi = 4 * wave, channel half 0 (half 1: + 768)

    f32x16 acc[16];                                    // [j][tile half][channel half]
    f32x4 bq[4][2][3];                                 // B ring: xi step j in set j & 3, requested three steps ahead
    LM_TICK(7)
    // prologue: patches of slots 0 .. 2 (four buffers in rotation), B of steps 0 .. 2; columns 0 .. 2 of slot 0's row pass, A(0, half 0)
#pragma unroll
    for (int s_ = 0; s_ < RLPW; ++s_)
        __builtin_amdgcn_global_load_lds((gptr_t*)gsrc[s_], (lptr_t*)(raw0 + (s_ * 4 + wave) * 256), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bload3(bq[k][0], bvoff, bbase + (long)k * bxi);
        bload3(bq[k][1], bvoff, bbase + (long)k * bxi + 768);
    }
#pragma unroll
    for (int s_ = 0; s_ < RLPW; ++s_)
        __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[s_] + RKS), (lptr_t*)(raw0 + RRAWH + (s_ * 4 + wave) * 256), 16, 0, 0);
    {
        const long g2 = H > 2 ? 2 * RKS : 0;
#pragma unroll
        for (int s_ = 0; s_ < RLPW; ++s_)
            __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[s_] + g2), (lptr_t*)(raw0 + 2 * RRAWH + (s_ * 4 + wave) * 256), 16, 0, 0);
    }
    // the 256 accumulator writes go out HERE, under the cold loads (the compiler sinks them to the loop's doorstep otherwise: 2.7 k cycles
    // in front of the first MFMA of every workgroup)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        asm volatile("" : "+a"(acc[k]));
    }
    asm volatile("s_waitcnt vmcnt(36)" ::: "memory");      // the patches of slot 0 (18 B loads and 18 patch loads are younger)
    __builtin_amdgcn_s_barrier();
    LM_TICK(8)
    RowsR rr[2];
    f32x4 a0[3], a1[3];
    wino_rows_rowpass(raw0, rb1[0], rb2[0], rsgn, rr[0]);  // (column 3 is made again in step 1 of slot 0)
    wino_rows_rowpass(raw0, rb1[1], rb2[1], rsgn, rr[1]);
    wino_rows_make_a<0>(rr[0], a0);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");       // the patches of slot 1: column passes read them from step 0 on
    __builtin_amdgcn_s_barrier();
    LM_TICK(0)
    for (int h = 0; h < H; ++h) {
        const float* const rawC = raw0 + (h & 3) * RRAWH;                                  // patches of this slot (column 3 in step 1)
        const float* const rawN = raw0 + ((h + 1) & 3) * RRAWH;                            // patches of slot h + 1 (columns 0, 2, 1)
        float* const rawld = raw0 + ((h + 3) & 3) * RRAWH;                                 // patches of slot h + 3 replace those of slot h - 1
        const long goff = h + 3 < H ? (long)(h + 3) * RKS : 0;                             // (nothing left to fetch: harmless re-read)
        const float* const bs = bbase + (long)h * bstep;
        const float* const bs_next = bbase + (long)(h + 1 < H ? h + 1 : 0) * bstep;
#define LM_BPRE(X) ((X) < 4 ? bs + (long)(X) * bxi : bs_next + (long)((X) - 4) * bxi)
        // NW = loads younger than B(J) when step J starts: 12 B loads and the patch loads of the three steps before (3, 3, 3, 0 per step)
#define LM_RSTEP(J, NW, RAWCOL)                                                                                                        \
        bwait6<NW>(bq[(J) & 3]);                                                                                                      \
        LM_TICK((J) == 0 ? 1 : 3)                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
        wino_rows_substep<J, 0>(acc, bq, a0, a1, rr, RAWCOL, rb1, rb2, rsgn, bvoff, LM_BPRE((J) + 3), gsrc, goff, rawld, wave);       \
        wino_rows_substep<J, 1>(acc, bq, a1, a0, rr, RAWCOL, rb1, rb2, rsgn, bvoff, LM_BPRE((J) + 3), gsrc, goff, rawld, wave);       \
        LM_TICK((J) == 3 ? 2 : 4)
        static_assert(RLPW == 9, "NW table: patch loads per step 3, 3, 3, 0");
        LM_RSTEP(0, 18, rawN)
        LM_RSTEP(1, 18, rawC)
        LM_RSTEP(2, 18, rawN)
        LM_RSTEP(3, 21, rawN)
#undef LM_RSTEP
#undef LM_BPRE
        asm volatile("s_waitcnt vmcnt(33)" ::: "memory");     // this wave's patch loads of the PREVIOUS slot have landed (33 loads per slot)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // patches of slot h + 2 visible; every wave is done with those of slot h
        LM_TICK(5)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bwait3<0>(bq[k][0]);
        bwait3<0>(bq[k][1]);
    }

#include "wino_rows_epilogue.h"
#ifdef LM_IPROF
    LM_TICK(6)
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 11; ++k) g_iprof[blockIdx.x % IPROF_WG][k] = (unsigned long long)iprof[k];
        g_iprof[blockIdx.x % IPROF_WG][11] = 1ull;
    }
#endif
}

// runs of adjacent tiles a 64-tile block can touch: floor((IBM - 2) / Tx) + 2
bool wino_implicit_ok(const WinoGeom& g) { return (IBM - 2) / g.Tx + 2 <= INSEG; }

}  // namespace

#ifdef LM_IPROF
extern "C" __attribute__((visibility("default"))) int lm_iprof_read(unsigned long long* out, int reset) {
    static unsigned long long host[IPROF_WG][12];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_iprof), sizeof(host)) != hipSuccess) return 1;
    for (int k = 0; k < 12; ++k) out[k] = 0;
    for (int w = 0; w < IPROF_WG; ++w)
        for (int k = 0; k < 12; ++k) out[k] += host[w][k];
    if (reset) {
        static unsigned long long zero[IPROF_WG][12];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_iprof), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

namespace {
// Address of a zero-source symbol ON THE CURRENT DEVICE (a __device__ variable has one instance per device: a process that switches
// devices must not hand the first device's pointer to a kernel on another one).  which = 0: g_wino_zero, 1: g_wino_zeros.
int wino_symbol_per_device(int which, const float** out) {
    static const float* cache[2][64] = {{nullptr}, {nullptr}};
    int dev = 0;
    LM_HIP(hipGetDevice(&dev));
    LM_REQUIRE(dev >= 0 && dev < 64, "conv_wino: device index %d", dev);
    if (!cache[which][dev]) {
        void* sym = nullptr;
        if (which == 0) LM_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_wino_zero)));
        else LM_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_wino_zeros)));
        cache[which][dev] = (const float*)sym;
    }
    *out = cache[which][dev];
    return LM_OK;
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
    if (int e = wino_symbol_per_device(0, &p.zero)) return e;
    hipStream_t s = (hipStream_t)stream;
    // 128 x 64 tiles: 4 output + 1 temporary accumulator sets = 246 registers -> two workgroups per CU (the tile / slab / buffering
    // variants that were measured against it in round 1 are listed in profiles/README.md; their code is gone)
    return launch_wino<128, 64, 32, 64>(p, s);
}

// Transform + GEMM in one call (workspace = V).
LM_API int lm_conv3x3_winograd_f32(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                                   const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W, int Cin,
                                   int Cout, int dil, int act, void* workspace, long workspace_bytes) {
    if (int e = lm_winograd_input_transform_f32(stream, x, ldx, B, H, W, Cin, dil, workspace, workspace_bytes)) return e;
    return lm_winograd_gemm_f32(stream, workspace, wu, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, nullptr);
}

// 1 if lm_conv3x3_winograd_implicit_f32 covers the shape (wide enough tile rows: at most INSEG runs of adjacent tiles per block)
LM_API int lm_winograd_implicit_supported(int H, int W, int Cin, int dil) {
    if (dil < 1 || H < 1 || W < 1 || Cin < 32 || Cin % 32 != 0 || Cin > 1024) return 0;      // (Cin % 32: shares pack_wino's layout rules)
    return wino_implicit_ok(geom(1, H, W, dil)) ? 1 : 0;
}

// Same result as lm_conv3x3_winograd_f32 without the transformed-input tensor (wino_implicit_kernel): x NHWC (ldx floats between
// pixels), wu_frag = U = G g G^T repacked per wave fragment, [16][Cin/16][CoutP/32][2][64][4] floats:
//   wu_frag[xi][cs][nt][kk][lane][e] = U[xi][nt*32 + (lane & 31)][cs*16 + kk*8 + (lane >> 5)*4 + e]     (ops.pack_wino_fragments)
namespace {
int wino_implicit_launch(int mode, void* stream, const float* x, int ldx, const float* wu_frag, int CoutP, const float* scale,
                         const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                         int Cin, int Cout, int dil, int act, double* gn_partial) {
    LM_REQUIRE(x && wu_frag && y, "conv_wino_implicit: null pointer");
    LM_REQUIRE(lm_winograd_implicit_supported(H, W, Cin, dil) && B > 0, "conv_wino_implicit: unsupported shape (H=%d W=%d Cin=%d dil=%d)", H, W, Cin, dil);
    LM_REQUIRE(CoutP >= Cout && CoutP % 128 == 0, "conv_wino_implicit: CoutP=%d must be Cout=%d rounded up to 128", CoutP, Cout);
    LM_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldy >= Cout, "conv_wino_implicit: bad leading dimension");
    LM_REQUIRE(act == LM_ACT_NONE || act == LM_ACT_RELU, "conv_wino_implicit: activation %d not supported", act);
    LM_REQUIRE(!gn_partial || (res == nullptr && act == LM_ACT_NONE && Cout % 4 == 0), "conv_wino_implicit(gn stats): no residual / activation");
    WinoImpParams p;
    p.g = geom(B, H, W, dil);
    LM_REQUIRE((long)B * H * W * ldx < (1L << 40) && (long)B * H * W < (1L << 31) && p.g.T < (1L << 31), "conv_wino_implicit: tensor too large");
    p.x = x; p.U = wu_frag; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.ldx = ldx; p.ldr = ldr; p.ldy = ldy; p.C = Cin; p.Cout = Cout; p.NT = CoutP / 32; p.act = act;
    p.gn_part = gn_partial;
    if (int e = wino_symbol_per_device(1, &p.zeros)) return e;
    p.n_inner = 0;
    if (mode == 1) {        // split-precision GEMM, ROWS geometry (wino_rows_split_kernel)
        const size_t rlds = (size_t)(4 * RRAWH > 2 * RXCH ? 4 * RRAWH : 2 * RXCH) * sizeof(float);
        const long rblocks = (p.g.T / RBM) * ((Cout + RBN - 1) / RBN);
        // an input larger than the 256 MB Infinity Cache is re-read from HBM once per N tile in the N-outer order (256->256@288^2 B = 8:
        // 680 MB x 4): N inner there (2.33 -> 2.21 ms); smaller inputs keep N outer (256->256 d2@144^2: 0.581 vs 0.591 ms)
        p.n_inner = (long)B * H * W * Cin * 4 > (256L << 20) ? 1 : 0;
        LM_REQUIRE(rblocks > 0 && rblocks < (1L << 31) && p.g.T % RBM == 0, "conv_wino_implicit: bad grid %ld", rblocks);
        if (int e = lm_ensure_dynamic_lds((const void*)wino_rows_split_kernel, rlds)) return e;
        hipLaunchKernelGGL(wino_rows_split_kernel, dim3((unsigned)rblocks), dim3(256), rlds, (hipStream_t)stream, p);
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    // fp32, two geometries (bit-identical to each other and to the materialising pair):
    //   Cout <= 64: DUAL (32 tiles x 64 channels, 8 xi per wave, two workgroups per CU) - what the per-workgroup fixed costs of the
    //               thin layers want (64->64@288^2 B = 8: 0.324 ms with a 64 x 64 sixteen-xi tile, 0.290 with this one);
    //   Cout  > 64: PIPE (32 tiles x 128 channels, 16 xi per wave, the slab transform spread over the MFMA steps) - half the transform
    //               work and half the input re-reads per matrix operation (256->256@288^2: DUAL 3.39 ms, round-2 WIDE 3.25, PIPE 3.18).
    // LANEMAP_WINO_DUAL=1 forces DUAL everywhere (test_conv_winograd_geometries_bit_identical).
    static const bool dual_all = getenv("LANEMAP_WINO_DUAL") && atoi(getenv("LANEMAP_WINO_DUAL")) == 1;
    if (dual_all || Cout <= DBN) {
        const size_t dlds = (size_t)(DRAW + DVBUF) * sizeof(float);
        const long dblocks = (p.g.T / DBM) * ((Cout + DBN - 1) / DBN);
        LM_REQUIRE(dblocks > 0 && dblocks < (1L << 31) && p.g.T % DBM == 0, "conv_wino_implicit: bad grid %ld", dblocks);
        if (int e = lm_ensure_dynamic_lds((const void*)wino_dual_kernel, dlds)) return e;      // (52 KB: below the default limit, set anyway)
        hipLaunchKernelGGL(wino_dual_kernel, dim3((unsigned)dblocks), dim3(256), dlds, (hipStream_t)stream, p);
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    const size_t plds = (size_t)(2 * PRAWH + 2 * PVH) * sizeof(float);
    const long pblocks = (p.g.T / PBM) * ((Cout + PBN - 1) / PBN);
    LM_REQUIRE(pblocks > 0 && pblocks < (1L << 31) && p.g.T % PBM == 0 && Cin % (2 * PKS) == 0, "conv_wino_implicit: bad grid %ld", pblocks);
    if (int e = lm_ensure_dynamic_lds((const void*)wino_pipe_kernel, plds)) return e;
    hipLaunchKernelGGL(wino_pipe_kernel, dim3((unsigned)pblocks), dim3(256), plds, (hipStream_t)stream, p);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

}  // namespace

LM_API int lm_conv3x3_winograd_implicit_f32(void* stream, const float* x, int ldx, const float* wu_frag, int CoutP, const float* scale,
                                            const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                            int Cin, int Cout, int dil, int act, double* gn_partial) {
    return wino_implicit_launch(0, stream, x, ldx, wu_frag, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, gn_partial);
}

// The same convolution with the GEMM on the bf16 matrix cores: every fp32 operand split exactly into three bf16 pieces, six piece
// products per multiply, fp32 accumulation (error of the class of an fp32 rounding, NOT bit-identical to the fp32 kernels).
// wu_frag3 = U split the same way and repacked per wave fragment, [16][Cin/16][CoutP/32][3 pieces][64 lanes][8 bf16]:
//   piece k of U[xi][nt*32 + (lane & 31)][cs*16 + 4*(lane >> 5) + (e & 3) + 8*(e >> 2)], e = 0..7     (ops.pack_wino_fragments_bf16x3;
//   the k order of a lane half = the channels of its two fp32 A fragments)
LM_API int lm_conv3x3_winograd_implicit_bf16x3(void* stream, const float* x, int ldx, const void* wu_frag3, int CoutP, const float* scale,
                                               const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                               int Cin, int Cout, int dil, int act, double* gn_partial) {
    return wino_implicit_launch(1, stream, x, ldx, (const float*)wu_frag3, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act,
                                gn_partial);
}

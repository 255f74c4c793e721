// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores: the 3x3 / stride 1 / pad == dilation layers of the FPN
// (baseline/models/pcencoder/postprojector.py:322-338 BasicBlock convs, :597-599 smooth*, :615-647 conv2/conv3/semantic_branch*)
// with 36 instead of 144 multiplies per 4x4 output block and (cin, cout) pair - 2.25 per output against 4 for F(2x2, 3x3)
// (the kernels of rounds 1-3, removed in round 5) and 9 for the direct sum: 0.5625x the matrix work of F(2x2), still exact fp32 MFMA.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      interpolation points 0, +-1, +-2, inf (Lavin & Gray)
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]   (U = G g G^T in fp64 at pack time, ops.pack_wino44)
//
// Numerics were priced before the kernel was written (tests/study_winograd_f44.py, profiles/r3_f44_numerics_study.txt): through the
// whole config-2 network the raw outputs stay within 2.5e-5 .. 7.6e-5 of the reference on scales 6.5 .. 43 (F(2x2): 1.4e-5 .. 5.7e-5),
// no thresholded decision outside the reference's own margin.  The results are NOT bit-identical to the F(2x2) family; they ARE
// bit-identical between the two implementations in this file, which share every arithmetic helper:
//   * the materialising twin (wino44_input_kernel -> V in HBM, wino44_gemm_kernel -> M in HBM, wino44_output_kernel): three plain
//     kernels, test infrastructure for the fused one (lm_conv3x3_winograd44_twin_f32);
//   * wino44_kernel: one workgroup = 32 tiles (512 output pixels) x 64 output channels, no V / M tensor in HBM.
//
// wino44_kernel.  36 accumulators of a 32 x 32 block are 576 registers, so the 36 xi are split over the four waves of the workgroup by
// QUADRANT of the 6 x 6 transform: wave (qa, qb) owns xi = (i, j) with i in 3 qa .. 3 qa + 2, j in 3 qb .. 3 qb + 2 for all 32 tiles and
// all 64 channels: 9 xi x 2 channel blocks x 16 = 288 accumulator registers, one wave per SIMD.  Every wave reads only its own nine
// planes of V (LDS) and its own nine planes of U (global -> registers, ring of 6 register sets 5 steps ahead): no operand is fetched
// twice inside a workgroup.  Per 16-channel unit of the input:
//   1. the RAW 6 x 6 patches of the 32 tiles (6 patch rows x 144 cells of 16 channels; horizontally adjacent tiles share two columns)
//      arrive in LDS through global_load_lds gathers straight from the NHWC tensor (zero block for padding), requested two units ahead,
//      double-buffered;
//   2. per 8-channel half (a SLOT of nine steps, one per xi of the wave: one A fragment from LDS, two B fragments, 8 MFMAs = 4 k steps
//      x 2 channel blocks): the 256 threads compute V = B^T d B of the NEXT half once (thread = tile x channel pair x half of the first
//      pass: 30 ds_read_b64, 72 packed FMAs / adds, 18 ds_write_b64 into V[xi][channel pair][tile][2], 36 KB, single-buffered) in pieces
//      that ride behind the first MFMA of each step - see W44Xf for the early / late plane split and the barrier in the middle of a slot.
//   f32 MFMA shares the SIMD's vector ALUs, so the transform's arithmetic is NOT hidden wherever it stands - it costs 72
//   packed VALU per 72 MFMAs - but it is paid once per 64 output channels, and 36 products replace 4 x 16.
// Epilogue: the 36 products of an output block sit in four different waves, so they meet in LDS: per 32-channel block every wave stores
// its nine planes straight from the accumulator registers (M[xi][tile][32 channels], 144 KB), then every thread takes one (tile, channel
// quad): 36 ds_read_b128, A^T M A (100 operations per element), the tail (BN scale / shift, residual, ReLU, GroupNorm partial sums) and
// sixteen 16-byte stores.  (The first version folded each wave's quadrant into partial outputs in registers and exchanged those: 2.8 k
// VALU per lane plus as many accumulator reads, spills, 114 k cycles per workgroup against 17 k for this one.)
#include "common.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

// ---- split-precision second line (round 6, LANEMAP_WINO_SPLIT=1; never the headline) ----------------------------------------------
// The Winograd-domain products on the fp16 matrix pipe with fp32-accurate products: every fp32 operand is split into two fp16 terms,
// x = hi + lo exactly to 22 bits (hi = x truncated to fp16, lo = (x - hi) truncated; x - hi is exact in fp32), and
//     v u ~= v_hi u_hi + v_hi u_lo + v_lo u_hi                       (the dropped v_lo u_lo is 2^-22 of the product)
// is three v_mfma_f32_32x32x8_f16 (32 cycles each, exact fp16 x fp16 products, fp32 accumulation) in place of four
// v_mfma_f32_32x32x2_f32 (64 cycles each) per 8 channels: 96 instead of 256 cycles of matrix time.  U is scaled by a power of two per
// layer on the host (max |U| ~ 2^13: its low terms stay normal fp16 numbers) and split once on the device (lm_wino44_split_fragments);
// V is split by the input transform right before it is stored (5 VALU instructions per channel pair and plane); the epilogue's scale
// undoes the power of two exactly.  Range: |V| <= 100 max|x| must stay below 65504, i.e. activations below ~650 (the truncating
// conversion saturates instead of producing infinities; nothing on the path checks it - one of the reasons this is a second line).
// One 32-bit word = the fp16 terms of a channel pair; (hi word, lo word) of a pair of fp32 values:
// (plain builtins, no inline asm: the compiler must see these VALU instructions to keep the wait states between them and an MFMA that reads
// their results - the twin's GEMM kernel consumes them straight from registers)
__device__ __forceinline__ u32x2 w44_split2(f32x2 t) {
#pragma clang fp contract(off)
    typedef __fp16 f16x2r __attribute__((ext_vector_type(2)));            // what __builtin_amdgcn_cvt_pkrtz returns
    const f16x2r h = __builtin_amdgcn_cvt_pkrtz(t[0], t[1]);
    const float h0 = (float)h[0], h1 = (float)h[1];
    const f16x2r l = __builtin_amdgcn_cvt_pkrtz(t[0] - h0, t[1] - h1);     // t - hi is exact: hi is t with its low mantissa bits cleared
    return u32x2{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};       // (whole-vector casts)
}

constexpr int QBM = 32, QBN = 64, QSEG = 4;
constexpr int QNCELL = 144;                     // cells (one pixel x 16 channels = 64 B) per patch row: 36 tile slots x 4 columns
constexpr int QLPW = 14;                        // patch loads per wave and unit (1 KB each)
constexpr int QGRP = 256 + 8;                   // floats between the 16-cell groups (one patch-load instruction each) of a raw buffer: 32 pad
                                                // bytes move consecutive groups by 8 banks - see roff in wino44_kernel
constexpr int QRAWF = QLPW * 4 * QGRP;          // floats of one raw buffer (57.75 KB; cells 864.. are zero-source padding)
constexpr int QVF = 36 * 256;                   // floats of the V buffer: 36 planes x [4 channel pairs][32 tiles][2]
#ifndef LM_QBD
#define LM_QBD 5
#define LM_QRING 6
#endif
constexpr int QBD = LM_QBD, QRING = LM_QRING;   // B fragments run QBD steps ahead in a ring of QRING register sets
static_assert(6 * (QNCELL / 16) * QGRP <= QRAWF && QNCELL % 16 == 0, "patch loads cover the unit");

__device__ __attribute__((aligned(16))) float g_w44_zeros[1024 + 32];   // zero source for padding cells, any channel unit (Cin <= 1024)

struct W44Geom {
    int B, H, W, dil, Ty, Tx;   // Ty x Tx tiles of 4 x 4 outputs per (image, phase); dil * dil phases
    int Timg, Tpad;             // real tiles per image, and that count rounded up to 32 (a workgroup never straddles images)
    long T;
    LmFastDiv dTx, dTy, ddil, dTpad;   // the divisions of wino44_kernel's set-up as multiply-high + shift (every dividend < 2^31)
};

W44Geom geom44(int B, int H, int W, int dil) {
    W44Geom g;
    g.B = B; g.H = H; g.W = W; g.dil = dil;
    g.Ty = ((H + dil - 1) / dil + 3) / 4;
    g.Tx = ((W + dil - 1) / dil + 3) / 4;
    g.Timg = dil * dil * g.Ty * g.Tx;
    g.Tpad = (g.Timg + QBM - 1) / QBM * QBM;
    g.T = (long)B * g.Tpad;
    g.dTx = lm_fastdiv_make((unsigned)g.Tx); g.dTy = lm_fastdiv_make((unsigned)g.Ty);
    g.ddil = lm_fastdiv_make((unsigned)dil); g.dTpad = lm_fastdiv_make((unsigned)g.Tpad);
    return g;
}

// tile t of an image -> phase and tile coordinates (linear order: phase, ty, tx)
__device__ __forceinline__ void tile44_decode(const W44Geom& g, int t, int& pa, int& pb, int& ty, int& tx) {
    tx = t % g.Tx;
    t /= g.Tx;
    ty = t % g.Ty;
    const int ph = t / g.Ty;
    pa = ph / g.dil;
    pb = ph - pa * g.dil;
}

// ---- the arithmetic both implementations share -------------------------------------------------------------------------------------
// 1-D input transform t = B^T d, fixed association (the fused kernel issues the same operations as packed instructions:
// fmaf(-5, d2, d4) = fma(-d2, 5, d4) bit for bit)
__device__ __forceinline__ void w44_bt(const float (&d)[6], float (&t)[6]) {
#pragma clang fp contract(off)
    t[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
    t[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
    const float p = __builtin_fmaf(-4.f, d[2], d[4]), q = __builtin_fmaf(-4.f, d[1], d[3]);
    t[1] = p + q;
    t[2] = p - q;
    const float r = d[4] - d[2], s = d[3] - d[1];
    t[3] = __builtin_fmaf(2.f, s, r);
    t[4] = __builtin_fmaf(-2.f, s, r);
}

// 1-D output transform y = A^T m, fixed association (T = float in the twin, four channels of a lane in the fused kernel)
__device__ __forceinline__ float w44_fma(float k, float a, float c) { return __builtin_fmaf(k, a, c); }
__device__ __forceinline__ f32x4 w44_fma(float k, f32x4 a, f32x4 c) { return __builtin_elementwise_fma(f32x4{k, k, k, k}, a, c); }
template <typename T>
__device__ __forceinline__ void w44_at(const T (&m)[6], T (&y)[4]) {
#pragma clang fp contract(off)
    const T s1 = m[1] + m[2], d1 = m[1] - m[2], s3 = m[3] + m[4], d3 = m[3] - m[4];
    y[0] = (m[0] + s1) + s3;
    y[1] = w44_fma(2.f, d3, d1);
    y[2] = w44_fma(4.f, s3, s1);
    y[3] = w44_fma(8.f, d3, d1) + m[5];
}

// ===================================================================================================================================
// Materialising twin (test infrastructure: plain kernels, one thread per element)
// V[xi][m][c] = (B^T d B)[xi] of the 6 x 6 patch of tile m (zero padded), rows first, then columns
__global__ __launch_bounds__(256) void wino44_input_kernel(const float* __restrict__ x, int ldx, W44Geom g, int C, float* __restrict__ V) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.T * C) return;
    const int c = (int)(e % C);
    const long m = e / C;
    const int b = (int)(m / g.Tpad), t = (int)(m - (long)b * g.Tpad);
    float d[6][6];
    int pa = 0, pb = 0, ty = 0, tx = 0;
    const bool real = t < g.Timg;
    if (real) tile44_decode(g, t, pa, pb, ty, tx);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int yy = (4 * ty + i - 1) * g.dil + pa, xx = (4 * tx + j - 1) * g.dil + pb;
            const bool ok = real && (4 * ty + i - 1) >= 0 && (4 * tx + j - 1) >= 0 && yy < g.H && xx < g.W;
            d[i][j] = ok ? x[(((long)b * g.H + yy) * g.W + xx) * ldx + c] : 0.f;
        }
    float w[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {                       // first pass: over the patch rows, per column
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float t6[6];
        w44_bt(col, t6);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = t6[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {                       // second pass: over the columns, per transformed row
        float t6[6];
        w44_bt(w[i], t6);
#pragma unroll
        for (int j = 0; j < 6; ++j) V[((long)(6 * i + j) * g.T + m) * C + c] = t6[j];
    }
}

// M[xi][m][n] = sum_c V[xi][m][c] U[xi][n][c]: one wave per (32 tiles, 32 channels, xi); the MFMA sequence of wino44_kernel (8-channel
// units in ascending order, k step e pairs channel 8 u + e with 8 u + 4 + e)
// (SPLIT: the operands are split on the fly by the helper the fused kernel and lm_wino44_split_fragments use, same three products in the same order)
template <bool SPLIT>
__global__ __launch_bounds__(64) void wino44_gemm_kernel(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M, long T,
                                                         int C, int CoutP) {
    const int lane = threadIdx.x;
    const long m0 = (long)blockIdx.x * 32;
    const int n0 = blockIdx.y * 32, xi = blockIdx.z;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* a = V + ((long)xi * T + m0 + (lane & 31)) * C + 4 * (lane >> 5);
    const float* b = U + ((long)xi * CoutP + n0 + (lane & 31)) * C + 4 * (lane >> 5);
    for (int u = 0; u < C / 8; ++u) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + 8 * u), bv = *reinterpret_cast<const f32x4*>(b + 8 * u);
        if constexpr (SPLIT) {
            const u32x2 a01 = w44_split2(f32x2{av[0], av[1]}), a23 = w44_split2(f32x2{av[2], av[3]});
            const u32x2 b01 = w44_split2(f32x2{bv[0], bv[1]}), b23 = w44_split2(f32x2{bv[2], bv[3]});
            const f16x4 ah = __builtin_bit_cast(f16x4, u32x2{a01[0], a23[0]}), al = __builtin_bit_cast(f16x4, u32x2{a01[1], a23[1]});
            const f16x4 bh = __builtin_bit_cast(f16x4, u32x2{b01[0], b23[0]}), bl = __builtin_bit_cast(f16x4, u32x2{b01[1], b23[1]});
            acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x8f16(al, bh, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        M[((long)xi * T + m0 + i) * CoutP + n0 + (lane & 31)] = acc[r];
    }
}

struct W44Epi {
    const float* scale; const float* shift; const float* res; float* y;
    int ldr, ldy, Cout, act;
    float post;            // split twin: 1 / (power-of-two scale of U), folded into the scale like the fused kernel does; 0 = exact twin
};

__device__ __forceinline__ float w44_tail(float v, int n, const W44Epi& e, long pix) {
#pragma clang fp contract(off)
    const float sh = e.shift ? e.shift[n] : 0.f;
    if (e.post != 0.f) v = v * ((e.scale ? e.scale[n] : 1.f) * e.post) + sh;
    else v = e.scale ? v * e.scale[n] + sh : v + sh;
    if (e.res) v += e.res[pix * e.ldr + n];
    if (e.act == LM_ACT_RELU) v = fmaxf(v, 0.f);
    return v;
}

// thread = (tile m, channel n): the 36 products -> A^T M A (rows first, then columns), tail, 16 guarded stores
__global__ __launch_bounds__(256) void wino44_output_kernel(const float* __restrict__ M, W44Geom g, int CoutP, W44Epi e) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.T * e.Cout) return;
    const int n = (int)(idx % e.Cout);
    const long m = idx / e.Cout;
    const int b = (int)(m / g.Tpad), t = (int)(m - (long)b * g.Tpad);
    if (t >= g.Timg) return;
    int pa, pb, ty, tx;
    tile44_decode(g, t, pa, pb, ty, tx);
    float z[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float col[6], y4[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = M[((long)(6 * i + j) * g.T + m) * CoutP + n];
        w44_at(col, y4);
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) z[yy][j] = y4[yy];
    }
#pragma unroll
    for (int yy = 0; yy < 4; ++yy) {
        float o[4];
        w44_at(z[yy], o);
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
            const int oy = (4 * ty + yy) * g.dil + pa, ox = (4 * tx + xx) * g.dil + pb;
            if (oy >= g.H || ox >= g.W) continue;
            const long pix = ((long)b * g.H + oy) * g.W + ox;
            e.y[pix * e.ldy + n] = w44_tail(o[xx], n, e, pix);
        }
    }
}

// U fragments (already scaled by the layer's power of two) -> fp16 term words, in place order: {c0 c1 | c2 c3 | c4 ..} of a lane's four
// channels become {hi 01, hi 23, lo 01, lo 23}
__global__ __launch_bounds__(256) void wino44_split_frag_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f32x4 v = in[i];
    const u32x2 a = w44_split2(f32x2{v[0], v[1]}), b = w44_split2(f32x2{v[2], v[3]});
    // (whole-vector bit cast: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0 whatever the index - hipcc 7.2)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    out[i] = __builtin_bit_cast(f32x4, u32x4{a[0], b[0], a[1], b[1]});
}

// ===================================================================================================================================
// The fused kernel
struct W44Params {
    const float* x; const float* U; const float* scale; const float* shift; const float* res; float* y; const float* zeros;
    int ldx, ldr, ldy, C, Cout, NT, act;      // NT = CoutP / 32 channel blocks in U
    int n_inner;                               // workgroup order: N tile inner
    LmFastDiv dnt;                             // / number of N tiles
    double* gn_part;
    W44Geom g;
    float post;                                // SPLIT: 1 / (power-of-two scale of U), folded into the epilogue's scale (exact); else 1.  (Last:
};                                             // the exact kernel's argument offsets - and with them its code - are what they were)

// packed fp32 pairs (two channels of a lane).  Plain asm (not volatile): the scheduler may move them, the arithmetic is fixed.
// (the multiplier b is one of three wave-uniform constant pairs: an SGPR-pair operand - with a "v" constraint every use cost a v_mov_b64)
#ifdef LM_W44_SCALAR_XF
// experiment: the same operations as plain (unpacked) f32 VALU instructions, two per pair (MI355X_MICROARCH.md: packed f32 VALU beside MFMAs
// costs ~11-13 cycles beyond its issue slot, plain v_fma_f32 / v_add_f32 are hidden fillers)
__device__ __forceinline__ float sc_fma(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float sc_fnma(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float sc_add(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sc_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(const f32x2 a, const f32x2 b, const f32x2 c) { return f32x2{sc_fma(a[0], b[0], c[0]), sc_fma(a[1], b[0], c[1])}; }
__device__ __forceinline__ f32x2 pk_fnma(const f32x2 a, const f32x2 b, const f32x2 c) { return f32x2{sc_fnma(a[0], b[0], c[0]), sc_fnma(a[1], b[0], c[1])}; }
__device__ __forceinline__ f32x2 pk_add(const f32x2 a, const f32x2 b) { return f32x2{sc_add(a[0], b[0]), sc_add(a[1], b[1])}; }
__device__ __forceinline__ f32x2 pk_sub(const f32x2 a, const f32x2 b) { return f32x2{sc_sub(a[0], b[0]), sc_sub(a[1], b[1])}; }
#else
__device__ __forceinline__ f32x2 pk_fma(const f32x2 a, const f32x2 b, const f32x2 c) {          // a b + c
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 pk_fnma(const f32x2 a, const f32x2 b, const f32x2 c) {         // (-a) b + c
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 pk_add(const f32x2 a, const f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(const f32x2 a, const f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif

struct W44K {
    f32x2 c2, c4, c5;
};

// w44_bt on channel pairs.  Source order = issue order (the scheduler leaves opaque asm statements where they stand and puts an s_nop
// between two that depend on each other): the six first-level operations, then the six that consume them - same arithmetic
__device__ __forceinline__ void pk_bt(const f32x2 (&d)[6], f32x2 (&t)[6], const W44K& k) {
    const f32x2 a = pk_fnma(d[2], k.c5, d[4]), b = pk_fnma(d[3], k.c5, d[5]);
    const f32x2 p = pk_fnma(d[2], k.c4, d[4]), q = pk_fnma(d[1], k.c4, d[3]);
    const f32x2 r = pk_sub(d[4], d[2]), s = pk_sub(d[3], d[1]);
    t[0] = pk_fma(d[0], k.c4, a);
    t[5] = pk_fma(d[1], k.c4, b);
    t[1] = pk_add(p, q);
    t[2] = pk_sub(p, q);
    t[3] = pk_fma(s, k.c2, r);
    t[4] = pk_fnma(s, k.c2, r);
}

// Transform of one 8-channel half (a SLOT of nine MFMA steps), this thread = (tile, channel pair, LOWER): first pass over the patch rows
// for its three transformed rows (LOWER = false: i = 0, 1, 2 from patch rows 0..4; true: i = 5, 3, 4 from patch rows 1..5), second pass
// over the columns, 18 planes.  Round 4, third version: NOTHING of it has a phase of its own any more - the pieces ride behind the
// MFMAs of the slot BEFORE the one that multiplies their result (an LDS instruction behind an f32 MFMA is free, a VALU instruction costs
// ~8 cycles of matrix time wherever it stands: DESIGN 3.3), which removes the phase's latency parts (LDS store drain, barrier, first
// A-fragment read with nothing to overlap: ~450 of its 1,040 cycles).  V stays SINGLE-buffered (a second 36 KB buffer does not fit
// beside two 57.75 KB patch buffers): a plane of V(s+1) may be overwritten once its owner wave has read V(s)'s - so the slot has a
// barrier in the middle, the nine planes of every quadrant are split into EARLY ones (local index k <= 4, read in steps 0..4: their
// successors are stored in steps 5..7, behind the mid barrier) and LATE ones (k >= 5, read in steps 5..8: their successors stay in
// their registers over the end of the slot and are stored in steps 0..3 of the next one, visible behind ITS mid barrier).
// Rows by class: A = the row with i % 3 == 0 (planes k = 0..2: early), B = i % 3 == 1 (k = 3, 4 early; k = 5 late), C = i % 3 == 2 (late).
// Step k of a slot:   0: read cols 0, 1 | store late 0, 1        1: pass 1 of cols 0, 1 | read cols 2, 3 | store late 2, 3
//   2: pass 1 of cols 2, 3 | read cols 4, 5 | store late 4, 5     3: pass 1 of cols 4, 5 | store late 6, 7      4: pass 2 of row A
//   -- mid barrier --   5: pass 2 of row B | store A 0..2         6: pass 2 of row C | store A 3..5             7: store B early (4)
// rawrow = raw buffer + 8 * half + (LOWER ? one patch row : 0); roff[c] = float offset of patch column c of the tile (+ 2 * channel pair).
struct W44Xf {
    f32x2 e[6][5];           // [patch column][patch row 0..4 (LOWER: 1..5)] - two columns live at a time
    f32x2 w[3][6];           // first-pass results by class: [A, B, C][column]
    f32x2 tA[6], tB[6], tC[6];     // second-pass results; the late planes - (B, j = 2), (B, 5), (C, 0..5) - stay here over the end of the slot
};                                 // and are stored in steps 0..3 of the next one, before steps 5 / 6 compute their successors
template <int C>
__device__ __forceinline__ void w44_preread(W44Xf& d, const float* rawrow, const int (&roff)[6]) {
    constexpr int ROWF = (QNCELL / 16) * QGRP;
    const float* s = rawrow + roff[C];
#pragma unroll
    for (int r = 0; r < 5; ++r) d.e[C][r] = *reinterpret_cast<const f32x2*>(s + r * ROWF);
}
// first pass of patch columns C and C + 1 (first-level operations of both, then the second level: no dependent neighbours):
// (upper) t0, t1, t2 -> classes A, B, C; (lower) t5, t3, t4 -> classes C, A, B
template <bool LOWER, int C>
__device__ __forceinline__ void w44_pass1x2(W44Xf& d, const W44K& k) {
    const f32x2 (&e)[5] = d.e[C];
    const f32x2 (&f)[5] = d.e[C + 1];
    if constexpr (!LOWER) {           // e = d0 .. d4
        const f32x2 a0 = pk_fnma(e[2], k.c5, e[4]), a1 = pk_fnma(f[2], k.c5, f[4]);
        const f32x2 p0 = pk_fnma(e[2], k.c4, e[4]), p1 = pk_fnma(f[2], k.c4, f[4]);
        const f32x2 q0 = pk_fnma(e[1], k.c4, e[3]), q1 = pk_fnma(f[1], k.c4, f[3]);
        d.w[0][C] = pk_fma(e[0], k.c4, a0);
        d.w[0][C + 1] = pk_fma(f[0], k.c4, a1);
        d.w[1][C] = pk_add(p0, q0);
        d.w[1][C + 1] = pk_add(p1, q1);
        d.w[2][C] = pk_sub(p0, q0);
        d.w[2][C + 1] = pk_sub(p1, q1);
    } else {                          // e = d1 .. d5
        const f32x2 a0 = pk_fnma(e[2], k.c5, e[4]), a1 = pk_fnma(f[2], k.c5, f[4]);
        const f32x2 r0 = pk_sub(e[3], e[1]), r1 = pk_sub(f[3], f[1]);            // r = d4 - d2
        const f32x2 s0 = pk_sub(e[2], e[0]), s1 = pk_sub(f[2], f[0]);            // s = d3 - d1
        d.w[2][C] = pk_fma(e[0], k.c4, a0);                                      // t5 = 4 d1 - 5 d3 + d5  (i = 5: class C)
        d.w[2][C + 1] = pk_fma(f[0], k.c4, a1);
        d.w[0][C] = pk_fma(s0, k.c2, r0);                                        // t3 (i = 3: class A)
        d.w[0][C + 1] = pk_fma(s1, k.c2, r1);
        d.w[1][C] = pk_fnma(s0, k.c2, r0);                                       // t4 (i = 4: class B)
        d.w[1][C + 1] = pk_fnma(s1, k.c2, r1);
    }
}
// the LDS part of step K's transform share.  vA / vB / vC = V + plane offset of the thread's class-A / B / C row + its element offset
// (SPLIT: a V plane is [hi / lo][4 channel pairs][32 tiles] words - the store is two ds_write_b32 128 words apart, v = plane + pair * 32 + tile)
template <bool SPLIT>
__device__ __forceinline__ void w44_vstore(float* v, int j, const f32x2 t) {
    if constexpr (SPLIT) {
        const u32x2 w = w44_split2(t);
        reinterpret_cast<unsigned*>(v)[j * 256] = w[0];
        reinterpret_cast<unsigned*>(v)[j * 256 + 128] = w[1];
    } else {
        *reinterpret_cast<f32x2*>(v + j * 256) = t;
    }
}
template <int K, bool SPLIT>
__device__ __forceinline__ void w44_xf_lds(W44Xf& d, const float* rawrow, const int (&roff)[6], float* vA, float* vB, float* vC) {
#ifdef LM_QABL_NOVST                         // (PMC ablation: the transform without its V stores - whose LDS bank conflicts are they?)
    auto st = [](float* v, int j, const f32x2 t) { asm volatile("" :: "v"(v), "v"(t)); };
#else
    auto st = [](float* v, int j, const f32x2 t) { w44_vstore<SPLIT>(v, j, t); };
#endif
    if constexpr (K == 0) {
        st(vB, 2, d.tB[2]); st(vB, 5, d.tB[5]);
        w44_preread<0>(d, rawrow, roff); w44_preread<1>(d, rawrow, roff);
    } else if constexpr (K == 1) {
        st(vC, 0, d.tC[0]); st(vC, 1, d.tC[1]);
        w44_preread<2>(d, rawrow, roff); w44_preread<3>(d, rawrow, roff);
    } else if constexpr (K == 2) {
        st(vC, 2, d.tC[2]); st(vC, 3, d.tC[3]);
        w44_preread<4>(d, rawrow, roff); w44_preread<5>(d, rawrow, roff);
    } else if constexpr (K == 3) {
        st(vC, 4, d.tC[4]); st(vC, 5, d.tC[5]);
    } else if constexpr (K == 5) {
        st(vA, 0, d.tA[0]); st(vA, 1, d.tA[1]); st(vA, 2, d.tA[2]);
    } else if constexpr (K == 6) {
        st(vA, 3, d.tA[3]); st(vA, 4, d.tA[4]); st(vA, 5, d.tA[5]);
    } else if constexpr (K == 7) {
        st(vB, 0, d.tB[0]); st(vB, 1, d.tB[1]); st(vB, 3, d.tB[3]); st(vB, 4, d.tB[4]);
    }
}
// the VALU part of step K's transform share (issued LAST behind the step's first MFMA: everything queued behind a VALU instruction
// waits for the MFMA in front of it to leave the pipe)
template <int K>
__device__ __forceinline__ void w44_xf_valu(W44Xf& d, bool lower, const W44K& k) {
    if constexpr (K >= 1 && K <= 3) {
        if (lower) w44_pass1x2<true, 2 * K - 2>(d, k);
        else w44_pass1x2<false, 2 * K - 2>(d, k);
    } else if constexpr (K == 4) {
        pk_bt(d.w[0], d.tA, k);
    } else if constexpr (K == 5) {
        pk_bt(d.w[1], d.tB, k);
    } else if constexpr (K == 6) {
        pk_bt(d.w[2], d.tC, k);
    }
}
// the whole transform of one slot at once (prologue: V(0) has nobody to hide behind): every plane stored; the first slot stores the late
// ones again in its steps 0..3 (the same bits)
template <bool SPLIT>
__device__ __forceinline__ void w44_xf_all(W44Xf& d, const float* rawrow, const int (&roff)[6], float* vA, float* vB, float* vC, bool lower,
                                           const W44K& k) {
    w44_preread<0>(d, rawrow, roff); w44_preread<1>(d, rawrow, roff); w44_preread<2>(d, rawrow, roff);
    w44_preread<3>(d, rawrow, roff); w44_preread<4>(d, rawrow, roff); w44_preread<5>(d, rawrow, roff);
    w44_xf_valu<1>(d, lower, k); w44_xf_valu<2>(d, lower, k); w44_xf_valu<3>(d, lower, k);
    w44_xf_valu<4>(d, lower, k); w44_xf_valu<5>(d, lower, k); w44_xf_valu<6>(d, lower, k);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        w44_vstore<SPLIT>(vA, j, d.tA[j]);
        w44_vstore<SPLIT>(vB, j, d.tB[j]);
        w44_vstore<SPLIT>(vC, j, d.tC[j]);
    }
}

// B fragment loads bypass the compiler's wait-count bookkeeping: explicit s_waitcnt vmcnt(N), tied to the destination
// registers through "+v" operands.
__device__ __forceinline__ void q_bload2(f32x4 (&b)[2], unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %2, %3\n\t"
                 "global_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(b[0]), "=&v"(b[1]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void q_bwait(f32x4 (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}

#ifdef LM_QPROF                             // (tools/build_variant.sh probe: per-phase shader-clock cycles of wave 0, one record per workgroup)
constexpr int QPROF_WG = 16384;
__device__ unsigned long long g_qprof[QPROF_WG][16];
__device__ unsigned long long g_qgap[QPROF_WG][3];      // first / last clock of the workgroup and the CU it ran on (lm_qgap_report: idle gaps between workgroups)
#define LM_QTICK(slot)                                       \
    {                                                        \
        const long long t_now = clock64();                   \
        qprof[slot] += t_now - t_last;                       \
        t_last = t_now;                                      \
    }
#else
#define LM_QTICK(slot)
#endif

// The wave's 288 accumulator registers exceed the 256 AGPRs: the compiler keeps 32 of them in VGPRs and, left to itself, swaps
// accumulators between the two files inside the loop (48 v_accvgpr_read + 48 v_accvgpr_write per phase, 8 cycles of matrix time each).
// The ninth xi therefore accumulates through this wrapper, which pins its two blocks to VGPRs; the other eight fill the AGPRs exactly.
__device__ __forceinline__ void mfma_vgpr(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <bool VACC>
__device__ __forceinline__ void q_mfma(f32x16& acc, float a, float b) {
    if constexpr (VACC) mfma_vgpr(acc, a, b);
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}
// SPLIT: one fp16 product term of 8 channels (lo / hi word pairs of a fragment: two consecutive registers = four fp16 k values)
template <bool VACC>
__device__ __forceinline__ void q_mfma16(f32x16& acc, f32x2 a, f32x2 b) {
    const f16x4 ah = __builtin_bit_cast(f16x4, a), bh = __builtin_bit_cast(f16x4, b);
    if constexpr (VACC) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh));
    else acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bh, acc, 0, 0, 0);
}
// Patch loads of unit u + 2 go out in the SECOND phase of unit u (steps 9..17: two per step in 9..13, one in 14..17), into the buffer
// whose last reads (the pre-reads of unit u's second half) were issued a phase earlier.
constexpr int q_ndma(int S) { return S < 9 ? 0 : (S < 14 ? 2 : 1); }
constexpr int q_dma0(int S) { return S < 9 ? 0 : (S < 14 ? 2 * (S - 9) : 10 + (S - 14)); }
// loads that may stay outstanding when the B fragments of step S are needed: 2 QBD younger B loads + the patch loads of steps S-QBD .. S
constexpr int q_nwait(int S) {
    int c = 0;
    for (int j = S - QBD; j <= S; ++j) c += q_ndma(((j % 18) + 18) % 18);
    return 2 * QBD + c;
}
static_assert(q_dma0(17) + q_ndma(17) == QLPW, "the second phase issues every patch load of a unit");

// A fragment of one xi: V plane [4 channel pairs][32 tiles][2] - the 16 lanes of a ds_write_b64 group of the transform (16 tiles of one
// channel pair) and the 32 lanes of a read pass here cover contiguous bytes: conflict-free on both sides ([2 k halves][32 tiles][4] with
// one ds_read_b128 made the transform's stores two-way conflicted).  Lane (tile, k half) takes pairs 2 k and 2 k + 1: channels 4 k .. 4 k + 3
template <bool SPLIT>
__device__ __forceinline__ f32x4 q_aread(const float* v) {
    if constexpr (SPLIT) {       // v = plane + 64 (lane >> 5) + tile: hi words of pairs 2 k, 2 k + 1, then their lo words (two ds_read2_b32)
        return f32x4{v[0], v[32], v[128], v[160]};
    } else {
        const f32x2 lo = *reinterpret_cast<const f32x2*>(v), hi = *reinterpret_cast<const f32x2*>(v + 64);
        return f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
}

// One step (one xi of the wave) of a slot: 8 MFMAs = 4 k steps x 2 channel blocks.  S = step within the 16-channel unit (0..17: two slots
// of nine); the B fragments of step S + QBD and the patch loads of this step (q_ndma) are issued first; q_nwait(S) loads may stay
// outstanding when this step's B fragments are needed.  Behind the first MFMA: the A fragment of the next step (not across a barrier:
// steps 4 and 8 leave it to the loop), the LDS part of this step's transform share, then its VALU part.
template <int S, bool SPLIT>
__device__ __forceinline__ void w44_step(f32x16& acc0, f32x16& acc1, f32x4 (&bq)[QRING][2], unsigned bvoff, const float* bpre,
                                         const float* anext, const f32x4& a_cur, f32x4& a_nxt, const float* const (&gsrc)[QLPW], long goff,
                                         float* rawld, int wave, W44Xf& xf, const float* prerow, const int (&roff)[6], float* vA, float* vB,
                                         float* vC, bool lower, const W44K& kk) {
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    constexpr int K = S % 9;
#ifndef LM_QABL_NOB
    q_bload2(bq[(S + QBD) % QRING], bvoff, bpre);
#endif
#ifndef LM_QABL_NOGLDS
#pragma unroll
    for (int i = 0; i < q_ndma(S); ++i) {
        constexpr int L0 = q_dma0(S);
        __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[L0 + i] + goff), (lptr_t*)(rawld + ((L0 + i) * 4 + wave) * QGRP), 16, 0, 0);
    }
#endif
    f32x4 (&b)[2] = bq[S % QRING];
#if !defined(LM_QABL_NOB) && !defined(LM_QABL_NOGLDS)
    q_bwait<q_nwait(S)>(b);
#else
    q_bwait<0>(b);
#endif
    constexpr bool VACC = K == 8;
    // SPLIT: a_cur = {hi 01, hi 23, lo 01, lo 23} of the tile's four channels, b[blk] likewise for the output channel: hi hi, hi lo, lo hi
    const f32x2 a_hi = {a_cur[0], a_cur[1]}, a_lo = {a_cur[2], a_cur[3]};
    if constexpr (SPLIT) q_mfma16<VACC>(acc0, a_hi, f32x2{b[0][0], b[0][1]});
    else q_mfma<VACC>(acc0, a_cur[0], b[0][0]);
    // (exact kernel: LDS instructions ride free behind the first f32 MFMA, VALU instructions cost matrix time wherever they stand - one block.
    //  SPLIT: an fp16 MFMA is 8 passes on a pipe of its own, and the step's VALU work (transform + split, ~110 cycles) is of the order of
    //  its six MFMAs (192 cycles): the scheduler interleaves them - see the group pattern at the end of the step)
    if constexpr (!SPLIT) __builtin_amdgcn_sched_barrier(0);
    if constexpr (K != 4 && K != 8) a_nxt = q_aread<SPLIT>(anext);
#ifndef LM_QABL_NOT
    w44_xf_lds<K, SPLIT>(xf, prerow, roff, vA, vB, vC);
    if constexpr (!SPLIT) __builtin_amdgcn_sched_barrier(0);
    w44_xf_valu<K>(xf, lower, kk);
#endif
    if constexpr (!SPLIT) __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPLIT) {
        q_mfma16<VACC>(acc1, a_hi, f32x2{b[1][0], b[1][1]});
        q_mfma16<VACC>(acc0, a_hi, f32x2{b[0][2], b[0][3]});
        q_mfma16<VACC>(acc1, a_hi, f32x2{b[1][2], b[1][3]});
        q_mfma16<VACC>(acc0, a_lo, f32x2{b[0][0], b[0][1]});
        q_mfma16<VACC>(acc1, a_lo, f32x2{b[1][0], b[1][1]});
#ifndef LM_SPLIT_NOGROUPS
        if constexpr (!VACC) {           // one MFMA, then a share of the step's VALU / LDS instructions, six times (leftovers follow)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x300, 3, 0);
            }
        }
#endif
    } else {
        q_mfma<VACC>(acc1, a_cur[0], b[1][0]);
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            q_mfma<VACC>(acc0, a_cur[t], b[0][t]);
            q_mfma<VACC>(acc1, a_cur[t], b[1][t]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Tail of one (tile, channel quad) of the fused epilogue, 16-byte path: second pass of the output transform row by row, scale / shift
// (v * 1 + 0 = v exactly: absent vectors need no second code path), GroupNorm partial sums, residual (prefetched), ReLU as
// fmaxf(v, 0 or -inf), stores.  FULL = every output of the tile exists (no per-store checks).
// fmaxf without the canonicalising v_max_f32 v, v, v the compiler puts in front of it when the operand comes out of a select (it quiets
// signalling NaNs, which no arithmetic result is): one instruction per element instead of two, the same bits for every other input
__device__ __forceinline__ float w44_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool FULL>
__device__ __forceinline__ void w44_tail_vec(const f32x4 (&z)[4][6], const f32x4 (&rpre)[16], float* yp, int rowstep, int colstep, const f32x4 sc,
                                             const f32x4 sh, float relu_lo, bool has_res, bool gn, f32x4& gsum, f32x4& gsq, int eny, int enx) {
#pragma unroll
    for (int yy = 0; yy < 4; ++yy) {
        f32x4 o[4];
        w44_at(z[yy], o);
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
#pragma clang fp contract(off)
            if (!FULL && !(yy < eny && xx < enx)) continue;
            f32x4 v = o[xx] * sc + sh;
            if (gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gsum[e] += v[e];
                    gsq[e] = __builtin_fmaf(v[e], v[e], gsq[e]);
                }
            }
            if (has_res) v += rpre[yy * 4 + xx];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w44_vmax(v[e], relu_lo);
#ifdef LM_QABL_NOSTORE                       // (timing ablation: keep the values alive, store one of sixteen)
            if (yy + xx == 0) *reinterpret_cast<f32x4*>(yp + yy * rowstep + xx * colstep) = v;
            else asm volatile("" :: "v"(v));
#else
            *reinterpret_cast<f32x4*>(yp + yy * rowstep + xx * colstep) = v;
#endif
        }
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void wino44_kernel(W44Params p) {
#ifdef LM_QPROF
    long long qprof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = clock64();
    const long long t_first = t_last;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];      // raw[2][QRAWF] | V[QVF]; the epilogue's exchange buffer over all of it
    float* const raw0 = smem;
    float* const Vbuf = smem + 2 * QRAWF;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (p.Cout + QBN - 1) / QBN;
    unsigned mblk, ntile;
    if (p.n_inner) {      // XCD-contiguous, N tile inner: the N tiles of an M block run side by side on one XCD (input lines shared in its L2)
        const unsigned bid = blockIdx.x, per = gridDim.x / 8;
        const unsigned lin = bid < per * 8 ? (bid % 8) * per + bid / 8 : bid;
        mblk = lm_fastdiv(lin, p.dnt);
        ntile = lin - mblk * (unsigned)n_tiles;
    } else {              // XCD-aware order, N tile outer: an XCD streams one N tile's U from its L2
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
    const long m0 = (long)mblk * QBM;
    const int n0 = (int)ntile * QBN;
    const W44Geom& g = p.g;
    const int bi = (int)lm_fastdiv((unsigned)m0, g.dTpad);
    const int t0 = (int)(m0 - (long)bi * g.Tpad);
    // run table: the 32 tiles are consecutive in the linear (phase, ty, tx) order = up to QSEG runs of horizontally adjacent tiles.
    // Run k holds tiles ts[k] .. ts[k+1]-1 and occupies tile SLOTS ts[k] + k .. ts[k+1] + k (one spill slot for patch columns 4, 5 of
    // its last tile); iy0 / ix0 = input pixel of patch cell (0, 0) of its first tile, oy0 / ox0 = output pixel (0, 0) of that tile
    int ts[QSEG + 1], sn[QSEG], iy0[QSEG], ix0[QSEG], oy0[QSEG], ox0[QSEG];
    {
        int at = 0, t = t0;
        const int rest = (int)lm_fastdiv((unsigned)t0, g.dTx), ph = (int)lm_fastdiv((unsigned)rest, g.dTy);
        int tx = t0 - rest * g.Tx, ty = rest - ph * g.Ty;
        int pa = (int)lm_fastdiv((unsigned)ph, g.ddil), pb = ph - pa * g.dil;
#pragma unroll
        for (int s_ = 0; s_ < QSEG; ++s_) {
            ts[s_] = at;
            const bool real = t < g.Timg && at < QBM;
            const int n = at < QBM ? min(QBM - at, g.Tx - tx) : 0;
            sn[s_] = real ? n : 0;
            iy0[s_] = (4 * ty - 1) * g.dil + pa;
            ix0[s_] = (4 * tx - 1) * g.dil + pb;
            oy0[s_] = 4 * ty * g.dil + pa;
            ox0[s_] = 4 * tx * g.dil + pb;
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
        ts[QSEG] = at;
    }
    // patch loads: load s of wave w fills chunks (s * 4 + w) * 64 .. + 63 of the raw buffer; chunk = 16 B = channel quad cq of a cell;
    // cell = patch row r x position pos; position = 16 (slot >> 2) + 4 c + (slot & 3) for column c (0..3) of tile slot `slot`: the cells
    // one column of consecutive tiles needs are neighbours in LDS; the 16 cells of a load are contiguous, consecutive loads QGRP floats apart
    const float* gsrc[QLPW];
    const int img_pix0 = bi * g.H * g.W;
    // Round 5: the set-up was 7.6 k cycles of address arithmetic per workgroup (tools/r4/qprof.py; 7.0 k with every cell served from the
    // zero block: not memory) - the run search, nine run-time divisions and four exec-masked regions per load.  Now lane l describes
    // tile SLOT l once (x / y pixel of its patch cell (0, 0), patch columns it may fetch: 4 (tiles left in its run) + 2, 0 without a run);
    // a load covers the four slots 4 gq .. 4 gq + 3 of one patch row (q = 4 s + wave = 9 r + gq: wave-uniform) and looks its slot up
    // with three ds_bpermute_b32; masks instead of selects keep the 64-bit offset out of divergent branches.
    int tab_x, tab_y, tab_n;
    {
        int n = sn[0], yb = iy0[0], xb = ix0[0], s0 = 0;
#pragma unroll
        for (int k = 1; k < QSEG; ++k) {
            const bool in = lane >= ts[k] + k;
            n = in ? sn[k] : n;
            yb = in ? iy0[k] : yb;
            xb = in ? ix0[k] : xb;
            s0 = in ? ts[k] + k : s0;
        }
        const int j = lane - s0;
        tab_x = xb + 4 * j * g.dil;
        tab_y = yb;
        tab_n = n > 0 ? 4 * (n - j) + 2 : 0;
    }
    const int gcc = (lane >> 4) & 3, gccd = gcc * g.dil, gsl4 = ((lane >> 2) & 3) << 2, gcq4 = (lane & 3) * 4;
#pragma unroll
    for (int s_ = 0; s_ < QLPW; ++s_) {
        const int q = s_ * 4 + wave;                      // 16-cell group: patch row r, tile slots 4 gq .. 4 gq + 3
        const int r = q / (QNCELL / 16), gq = q - r * (QNCELL / 16);
        const int idx = 16 * gq + gsl4;                   // (byte index of lane 4 gq + slot-in-group)
        const int x0 = __builtin_amdgcn_ds_bpermute(idx, tab_x), y0 = __builtin_amdgcn_ds_bpermute(idx, tab_y);
        const int nn = __builtin_amdgcn_ds_bpermute(idx, tab_n);
        const int yy = y0 + r * g.dil, xx = x0 + gccd;
        const bool ok = (r < 6) & (gcc < nn) & ((unsigned)yy < (unsigned)g.H) & ((unsigned)xx < (unsigned)g.W);
#ifdef LM_QABL_ZEROSRC                        // (timing ablation: every patch cell comes from the zero block - what do the scattered input lines cost?)
        gsrc[s_] = p.zeros + (ok ? 0 : 16) + gcq4;
#else
        const long m = -(long)ok;                         // all ones: the cell exists
        const long eoff = ((long)(img_pix0 + yy * g.W + xx) * p.ldx) & m;
        const unsigned long base = (unsigned long)p.zeros + (((unsigned long)p.x - (unsigned long)p.zeros) & (unsigned long)m);
        gsrc[s_] = (const float*)base + eoff + gcq4;
#endif
        // unit 0's patch load s goes out as soon as its source is known: issuing a load of cold, scattered lines stalls ~140 cycles
        // (profiles/r4_wino44_residual_issue_experiment.txt) - the address arithmetic of load s + 1 runs meanwhile
        __builtin_amdgcn_global_load_lds((gptr_t*)gsrc[s_], (lptr_t*)(raw0 + (s_ * 4 + wave) * QGRP), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // transform share: tile = lane & 31, LOWER = wave >> 1 (wave-uniform).  Bank layout of the raw reads (round 5, measured with
    // SQ_LDS_BANK_CONFLICT per ablation build, profiles/r5_wino44_lds_conflicts.txt): a ds_read_b64 is served in four groups of 16 lanes on
    // 32 banks, so the 16 lanes of a group must cover all 16 bank pairs.  A lane's address modulo 32 floats is 16 (slot & 1) [cell parity]
    // + 8 (group parity: consecutive 16-cell groups are QGRP = 264 floats apart) + 8 half + 2 cp: with the channel pair skewed by slot & 3
    // the four slot bits map onto the four position bits - any 16 consecutive slots are conflict-free, a run break inside a group costs
    // one cycle (before: cp skewed by slot >> 2 on 256-float groups, 8 positions for 16 lanes: 4.4-5.3 extra cycles per read)
    int roff[6], tvoff;
    {
        const int tl = lane & 31;
        int sg = 0;
#pragma unroll
        for (int k = 1; k < QSEG; ++k) sg += (ts[k] < QBM && tl >= ts[k]) ? 1 : 0;
        const int slot = tl + sg, slot1 = slot + 1;
        const int cp = (2 * (wave & 1) + (lane >> 5) + slot) & 3;
        const int pos0 = 16 * (slot >> 2) + (slot & 3), pos1 = 16 * (slot1 >> 2) + (slot1 & 3);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int pos = c < 4 ? pos0 + 4 * c : pos1 + 4 * (c - 4);
            roff[c] = (pos >> 4) * QGRP + (pos & 15) * 16 + 2 * cp;
        }
#ifdef LM_QABL_LINREAD                        // (timing / PMC ablation: conflict-free raw reads by construction - wrong data; the most a better layout can buy)
#pragma unroll
        for (int c = 0; c < 6; ++c) roff[c] = c * 64 + tl * 2;
#endif
        tvoff = SPLIT ? cp * 32 + tl : cp * 64 + tl * 2;
    }
    const bool lower = (wave >> 1) != 0;
    // plane rows of this thread by class (A: i % 3 == 0, B: == 1, C: == 2): upper threads i = 0, 1, 2; lower threads i = 3, 4, 5
    float* const vA = Vbuf + (lower ? 3 : 0) * (6 * 256) + tvoff;
    float* const vB = Vbuf + (lower ? 4 : 1) * (6 * 256) + tvoff;
    float* const vC = Vbuf + (lower ? 5 : 2) * (6 * 256) + tvoff;
    const W44K kk = {f32x2{2.f, 2.f}, f32x2{4.f, 4.f}, f32x2{5.f, 5.f}};
    // MFMA operands: A = V plane xi, channel pairs 2 (lane >> 5) and 2 (lane >> 5) + 1 of tile lane & 31; this wave's planes xi = xi00 + 6 ii + jj
    const int qa = wave >> 1, qb = wave & 1;
    const int xi00 = 18 * qa + 3 * qb;
    const float* const Vq = SPLIT ? Vbuf + xi00 * 256 + (lane >> 5) * 64 + (lane & 31)
                                  : Vbuf + xi00 * 256 + (lane >> 5) * 128 + (lane & 31) * 2;       // channel pairs 2 (lane >> 5) and 2 (lane >> 5) + 1
    const int nun = p.C / 16;                                    // 16-channel units
    const unsigned bvoff = (unsigned)lane * 16u;
    const long ustride = (long)p.NT * 256;                       // floats between 8-channel halves in U
    const long xstride = (long)(2 * nun) * ustride;              // floats between xi planes in U
    const float* const bbase = p.U + (long)xi00 * xstride + (long)(n0 >> 5) * 256;

    f32x16 acc[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][b][r] = 0.f;
    f32x4 bq[QRING][2];
    W44Xf xf;
    const int lowoff = lower ? (QNCELL / 16) * QGRP : 0;               // LOWER threads read patch rows 1..5
    LM_QTICK(0)
    // prologue: (raw unit 0 was requested while the sources were computed,) B of steps 0 .. QBD-1, THEN raw unit 1: the wait below leaves unit 1's loads (the youngest) in flight - they
    // are only needed before the second slot, and being older than every load of the loop they do not enter its wait counts
#define LM_QXI(K) (6 * ((K) / 3) + (K) % 3)
#pragma unroll
    for (int k = 0; k < QBD; ++k) q_bload2(bq[k], bvoff, bbase + (long)LM_QXI(k) * xstride);
    {
        const long goff1 = nun > 1 ? 16 : 0;
#pragma unroll
        for (int s_ = 0; s_ < QLPW; ++s_)
            __builtin_amdgcn_global_load_lds((gptr_t*)(gsrc[s_] + goff1), (lptr_t*)(raw0 + QRAWF + (s_ * 4 + wave) * QGRP), 16, 0, 0);
    }
    q_bwait<QLPW>(bq[0]);                      // unit 0 and the first B fragments have landed (this wave's part; the barrier collects all parts)
    __builtin_amdgcn_s_barrier();
    LM_QTICK(1)
    // V(0): the only transform with nothing to hide behind
#ifndef LM_QABL_NOT
    w44_xf_all<SPLIT>(xf, raw0 + lowoff, roff, vA, vB, vC, lower, kk);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LM_QTICK(2)
    __builtin_amdgcn_s_barrier();
    LM_QTICK(3)

    // B fragments of step S5 = S + QBD of the unit (S5 >= 18: first slot of the next unit)
#define LM_QBPRE(S5) ((S5) < 18 ? bu + (long)((S5) / 9) * ustride + (long)LM_QXI((S5) % 9) * xstride : bu_next + (long)LM_QXI((S5) - 18) * xstride)
#define LM_QSTEP(S, AC, AN) \
    w44_step<S, SPLIT>(acc[(S) % 9][0], acc[(S) % 9][1], bq, bvoff, LM_QBPRE((S) + QBD), Vq + LM_QXI(((S) % 9) + 1 < 9 ? ((S) % 9) + 1 : 0) * 256, \
                AC, AN, gsrc, goff, rawc_w, wave, xf, (S) < 9 ? rawc + 8 + lowoff : rawn + lowoff, roff, vA, vB, vC, lower, kk)
    // the barrier in the middle of a slot: this wave's late stores (steps 0..3) are done; behind it every wave's are, and the early
    // planes are free (every wave has read V(s)'s in steps 0..4)
#ifdef LM_QABL_NOMID                          // (timing ablation: what the barrier in the middle of a slot costs; results are wrong)
#define LM_QMID(AN) AN = q_aread<SPLIT>(Vq + LM_QXI(5) * 256);
#else
#define LM_QMID(AN)                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    LM_QTICK(4)                                              \
    __builtin_amdgcn_s_barrier();                            \
    LM_QTICK(5)                                              \
    AN = q_aread<SPLIT>(Vq + LM_QXI(5) * 256);
#endif
    for (int u = 0; u < nun; ++u) {
        const float* const rawc = raw0 + (u & 1) * QRAWF;              // unit u
        float* const rawc_w = raw0 + (u & 1) * QRAWF;                  // ... and the destination of unit u + 2's patch loads (second slot)
        const float* const rawn = raw0 + ((u + 1) & 1) * QRAWF;        // unit u + 1 (pre-read in the second slot)
        const long goff = u + 2 < nun ? (long)(u + 2) * 16 : 0;            // (nothing left to fetch: harmless re-read of unit 0)
        const float* const bu = bbase + (long)(2 * u) * ustride;
        const float* const bu_next = bbase + (long)(u + 1 < nun ? 2 * (u + 1) : 0) * ustride;
        f32x4 a0, a1;
        // ---- slot 2 u: channels 16 u .. 16 u + 7 multiplied, channels 16 u + 8 .. 16 u + 15 transformed behind the MFMAs
        a0 = q_aread<SPLIT>(Vq);
        LM_QSTEP(0, a0, a1); LM_QSTEP(1, a1, a0); LM_QSTEP(2, a0, a1); LM_QSTEP(3, a1, a0); LM_QSTEP(4, a0, a1);
        LM_QMID(a1)
        LM_QSTEP(5, a1, a0); LM_QSTEP(6, a0, a1); LM_QSTEP(7, a1, a0); LM_QSTEP(8, a0, a1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LM_QTICK(4)
        // unit u + 1 (patch loads of the previous unit's second slot, or of the prologue) is pre-read in the slot that follows: every
        // load older than the 18 B loads of the slot just finished has landed
        q_bwait<2 * 9>(bq[0]);
        LM_QTICK(6)
        __builtin_amdgcn_s_barrier();          // V(2 u + 1)'s early planes complete
        LM_QTICK(5)
        // ---- slot 2 u + 1: channels 16 u + 8 .. multiplied, the next unit's first half transformed, unit u + 2's patches requested
        a0 = q_aread<SPLIT>(Vq);
        LM_QSTEP(9, a0, a1);  LM_QSTEP(10, a1, a0); LM_QSTEP(11, a0, a1); LM_QSTEP(12, a1, a0); LM_QSTEP(13, a0, a1);
        LM_QMID(a1)
        LM_QSTEP(14, a1, a0); LM_QSTEP(15, a0, a1); LM_QSTEP(16, a1, a0); LM_QSTEP(17, a0, a1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LM_QTICK(4)
        __builtin_amdgcn_s_barrier();
        LM_QTICK(5)
    }
#undef LM_QMID
#undef LM_QSTEP
#undef LM_QBPRE
#undef LM_QXI
    static_assert(18 % QRING == 0 && QBD < QRING, "eighteen steps per unit walk the ring a whole number of times");
#pragma unroll
    for (int k = 0; k < QRING; ++k) q_bwait<0>(bq[k]);

    LM_QTICK(7)
#ifdef LM_QABL_NOEPI
    if (p.act != 12345) return;
#endif
    // ---- epilogue: the products of one 32-channel block go to LDS as M[xi][tile][32 channels] (144 KB; straight from the accumulator
    // registers), every thread takes one (tile, channel quad): 36 ds_read_b128, A^T M A (rows first, then columns: w44_at), tail, 16 stores
    const int etile = tid >> 3, ecq = tid & 7;
    int epix0;
    int eny = 0, enx = 0;                                   // valid output rows / columns of this thread's tile (0: no tile)
    {
        int nn = sn[0], oy = oy0[0], oxb = ox0[0], tb = 0;
#pragma unroll
        for (int k = 1; k < QSEG; ++k)
            if (etile >= ts[k]) {
                nn = sn[k]; oy = oy0[k]; oxb = ox0[k]; tb = ts[k];
            }
        const int ox = oxb + 4 * (etile - tb) * g.dil;
        epix0 = img_pix0 + oy * g.W + ox;
        if (nn > 0 && oy < g.H && ox < g.W) {
            eny = min(4, (int)lm_fastdiv((unsigned)(g.H - oy + g.dil - 1), g.ddil));
            enx = min(4, (int)lm_fastdiv((unsigned)(g.W - ox + g.dil - 1), g.ddil));
        }
    }
    float* const mw = smem + xi00 * 1024 + (4 * (lane >> 5)) * 32 + (lane & 31);      // this wave's planes, this lane's origin
    const float* const mr = smem + etile * 32 + ecq * 4;
    f32x4 gsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gsq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const bool full = eny == 4 && enx == 4;
    const int ebase = eny > 0 ? epix0 : img_pix0;                 // (a missing tile reads - and never writes - pixel 0 of its image)
    const int ey1 = max(eny - 1, 0), ex1 = max(enx - 1, 0);
    const float relu_lo = p.act == LM_ACT_RELU ? 0.f : -__builtin_inff();               // fmaxf(v, -inf) = v
    const bool has_res = p.res != nullptr, gn = p.gn_part != nullptr;
    LM_QTICK(8)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int n = n0 + blk * 32 + ecq * 4;
        const bool vec = (n + 3 < p.Cout) && ((p.ldy & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
        // residual (BasicBlock identity): the sixteen 16-byte loads of this thread's outputs go out BEFORE the exchange - inside the
        // store loop each was a memory round trip of its own in front of a store (y may alias res as far as the compiler knows).
        // Branch-free: offsets clamped into the tile's valid part (equal to the true offsets wherever an output exists)
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};             // (v * 1 + 0 = v exactly: same bits as the twin's `v + shift`)
        if (vec) {
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            if constexpr (SPLIT) sc = sc * p.post;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Cout) {
                    if (p.scale) sc[e] = p.scale[n + e];
                    if (p.shift) sh[e] = p.shift[n + e];
                    if constexpr (SPLIT) sc[e] = sc[e] * p.post;
                }
        }
        // residual (BasicBlock identity): the sixteen 16-byte loads of this thread's outputs are issued BETWEEN the product stores of the
        // exchange - issuing them costs ~140 cycles apiece here (cold lines, eight 128-byte segments 4 KB apart per wave instruction:
        // profiles/r4_wino44_residual_issue_experiment.txt), the LDS store path drains its queue meanwhile; inside the store loop of the tail
        // each would be a memory round trip of its own in front of a store (y may alias res as far as the compiler knows).
        // Branch-free: offsets clamped into the tile's valid part (equal to the true offsets wherever an output exists)
        f32x4 rpre[16];
#ifdef LM_QABL_NORES
        const bool load_res = false;
#else
        const bool load_res = vec && has_res;
#endif
        const float* const rp = p.res + (long)ebase * p.ldr + n;
        const int rs = g.W * g.dil * p.ldr, cs = g.dil * p.ldr;
        // (LDS-only barriers: __syncthreads() would also wait for the previous block's global stores to be acknowledged)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LM_QTICK(9)
        __builtin_amdgcn_s_barrier();          // patch / V buffers (blk 0) or the previous block's products are no longer read
        LM_QTICK(10)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mw[(6 * (k / 3) + k % 3) * 1024 + ((r & 3) + 8 * (r >> 2)) * 32] = acc[k][blk][r];
            if (k < 8 && load_res) {
#pragma unroll
                for (int q = 2 * k; q < 2 * k + 2; ++q)
                    rpre[q] = *reinterpret_cast<const f32x4*>(rp + min(q >> 2, ey1) * rs + min(q & 3, ex1) * cs);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LM_QTICK(12)
        __builtin_amdgcn_s_barrier();
        LM_QTICK(13)
        if (n < p.Cout && eny > 0) {
            f32x4 z[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 col[6], y4[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) col[i] = *reinterpret_cast<const f32x4*>(mr + (6 * i + j) * 1024);
                w44_at(col, y4);
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) z[yy][j] = y4[yy];
            }
            float* const yp = p.y + (long)ebase * p.ldy + n;
            const int rowstep = g.W * g.dil * p.ldy, colstep = g.dil * p.ldy;
            if (vec) {
                if (full) w44_tail_vec<true>(z, rpre, yp, rowstep, colstep, sc, sh, relu_lo, has_res, gn, gsum[blk], gsq[blk], eny, enx);
                else w44_tail_vec<false>(z, rpre, yp, rowstep, colstep, sc, sh, relu_lo, has_res, gn, gsum[blk], gsq[blk], eny, enx);
            } else {                           // channel counts / leading dimensions that rule out 16-byte accesses: element by element
                // (fully unrolled with guards: a run-time index into z would move the array - on the 16-byte path too - to scratch memory)
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    f32x4 o[4];
                    w44_at(z[yy], o);
#pragma unroll
                    for (int xx = 0; xx < 4; ++xx) {
#pragma clang fp contract(off)
                        const f32x4 v = o[xx] * sc + sh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (!(yy < eny && xx < enx && n + e < p.Cout)) continue;
                            if (gn) {
                                gsum[blk][e] += v[e];
                                gsq[blk][e] = __builtin_fmaf(v[e], v[e], gsq[blk][e]);
                            }
                            float u = v[e];
                            if (has_res) u += p.res[((long)ebase + (yy * g.W + xx) * g.dil) * p.ldr + n + e];
                            yp[yy * rowstep + xx * colstep + e] = fmaxf(u, relu_lo);
                        }
                    }
                }
            }
        }
        LM_QTICK(14)
    }
#ifdef LM_QPROF
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 15; ++k) g_qprof[blockIdx.x % QPROF_WG][k] = (unsigned long long)qprof[k];
        const long long t_end = clock64();
        g_qprof[blockIdx.x % QPROF_WG][11] = (unsigned long long)(t_end - t_first);
        g_qgap[blockIdx.x % QPROF_WG][0] = (unsigned long long)t_first;
        g_qgap[blockIdx.x % QPROF_WG][1] = (unsigned long long)t_end;
        g_qgap[blockIdx.x % QPROF_WG][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 16) |     // XCC_ID
                                           (__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) & 0xff00u);                     // HW_ID: cu, sh, se
    }
#endif
    if (p.gn_part) {      // fixed-order reduction: the 8 lanes of a wave that share a channel quad, then the four waves through LDS
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gsum[blk][e] += __shfl_xor(gsum[blk][e], o);
                    gsq[blk][e] += __shfl_xor(gsq[blk][e], o);
                }
        __syncthreads();
        if (lane < 8) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                *reinterpret_cast<f32x4*>(smem + (((wave * 2 + blk) * 8 + lane) * 2) * 4) = gsum[blk];
                *reinterpret_cast<f32x4*>(smem + (((wave * 2 + blk) * 8 + lane) * 2 + 1) * 4) = gsq[blk];
            }
        }
        __syncthreads();
        if (tid < 16) {
#pragma clang fp contract(off)
            const int blk = tid >> 3, cq = tid & 7;
            const int n = n0 + blk * 32 + cq * 4;
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(smem + (((w * 2 + blk) * 8 + cq) * 2) * 4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(smem + (((w * 2 + blk) * 8 + cq) * 2 + 1) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += a[e];
                    q[e] += b[e];
                }
            }
            const long chunk = t0 / 32;
            double* o = p.gn_part + (((long)bi * (g.Tpad / 32) + chunk) * p.Cout + n) * 2;
            for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                o[2 * e] = (double)s[e];
                o[2 * e + 1] = (double)q[e];
            }
        }
    }
}

// runs of adjacent tiles a 32-tile block can touch: floor((QBM - 2) / Tx) + 2
bool w44_ok(const W44Geom& g) { return (QBM - 2) / g.Tx + 2 <= QSEG; }

int w44_zeros(const float** out) {      // per device (a process may drive several)
    static const float* cache[64] = {nullptr};
    int dev = 0;
    LM_HIP(hipGetDevice(&dev));
    LM_REQUIRE(dev >= 0 && dev < 64, "conv_wino44: device index %d", dev);
    if (!cache[dev]) {
        void* sym = nullptr;
        LM_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_w44_zeros)));
        cache[dev] = (const float*)sym;
    }
    *out = cache[dev];
    return LM_OK;
}

}  // namespace

#ifdef LM_QPROF
extern "C" __attribute__((visibility("default"))) int lm_qprof_read(unsigned long long* out, int reset) {
    static unsigned long long host[QPROF_WG][16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_qprof), sizeof(host)) != hipSuccess) return 1;
    for (int k = 0; k < 17; ++k) out[k] = 0;
    for (int w = 0; w < QPROF_WG; ++w) {
        for (int k = 0; k < 16; ++k) out[k] += host[w][k];
        if (host[w][11]) ++out[16];
    }
    if (reset) {
        static unsigned long long zero[QPROF_WG][16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qprof), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
// idle time of a CU between two workgroups of the LAST launch (<= QPROF_WG workgroups): out = {workgroups, CUs seen, mean workgroup cycles,
// mean gap, median gap, 90th percentile gap, cycles from the first start to the last end, sum of workgroup cycles / (CUs x that span)}
extern "C" __attribute__((visibility("default"))) int lm_qgap_report(double* out) {
    static unsigned long long host[QPROF_WG][3];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_qgap), sizeof(host)) != hipSuccess) return 1;
    struct Rec { unsigned long long id, t0, t1; };
    static Rec recs[QPROF_WG];
    int n = 0;
    for (int w = 0; w < QPROF_WG; ++w)
        if (host[w][1]) recs[n++] = Rec{host[w][2], host[w][0], host[w][1]};
    qsort(recs, n, sizeof(Rec), [](const void* a, const void* b) {
        const Rec* x = (const Rec*)a; const Rec* y = (const Rec*)b;
        if (x->id != y->id) return x->id < y->id ? -1 : 1;
        return x->t0 < y->t0 ? -1 : (x->t0 > y->t0 ? 1 : 0);
    });
    static double gaps[QPROF_WG];
    int ng = 0, ncu = 0;
    double sumd = 0, sumg = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < n; ++i) {
        sumd += (double)(recs[i].t1 - recs[i].t0);
        if (recs[i].t0 < tmin) tmin = recs[i].t0;
        if (recs[i].t1 > tmax) tmax = recs[i].t1;
        if (i == 0 || recs[i].id != recs[i - 1].id) { ++ncu; continue; }
        gaps[ng] = (double)recs[i].t0 - (double)recs[i - 1].t1;
        sumg += gaps[ng++];
    }
    qsort(gaps, ng, sizeof(double), [](const void* a, const void* b) { return *(const double*)a < *(const double*)b ? -1 : (*(const double*)a > *(const double*)b ? 1 : 0); });
    out[0] = n; out[1] = ncu; out[2] = n ? sumd / n : 0; out[3] = ng ? sumg / ng : 0; out[4] = ng ? gaps[ng / 2] : 0; out[5] = ng ? gaps[(int)(0.9 * ng)] : 0;
    out[6] = (double)(tmax - tmin); out[7] = (ncu && tmax > tmin) ? sumd / ((double)ncu * (double)(tmax - tmin)) : 0;
    static unsigned long long zero[QPROF_WG][3];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_qgap), zero, sizeof(zero)) != hipSuccess) return 1;
    return 0;
}
#endif

// 1 if lm_conv3x3_winograd44_f32 covers the shape
LM_API int lm_winograd44_supported(int H, int W, int Cin, int dil) {
    if (dil < 1 || H < 1 || W < 1 || Cin < 16 || Cin % 16 != 0 || Cin > 1024) return 0;
    return w44_ok(geom44(1, H, W, dil)) ? 1 : 0;
}

// 32-tile chunks per image of the GroupNorm partial sums written by lm_conv3x3_winograd44_f32 (-> lm_gn_finalize's nchunk)
LM_API int lm_winograd44_gn_chunks(int H, int W, int dil) { return dil < 1 ? 0 : geom44(1, H, W, dil).Tpad / 32; }

// Winograd-domain products the F(4x4) kernels execute per launch: 36 per tile and (cin, cout) pair (for the executed-FLOP roofline)
LM_API long lm_winograd44_tiles(int B, int H, int W, int dil) { return dil < 1 ? 0 : geom44(B, H, W, dil).T; }

LM_API long lm_winograd44_twin_workspace_bytes(int B, int H, int W, int Cin, int CoutP, int dil) {
    if (dil < 1) return 0;
    return 36 * geom44(B, H, W, dil).T * (long)(Cin + CoutP) * (long)sizeof(float);
}

// y = act(conv3x3(x; pad = dil) * scale + shift + res), NHWC, through Winograd F(4x4, 3x3) without V / M tensors in HBM.
// wu_frag = U = G g G^T (fp64 -> fp32) repacked per wave fragment, [36][Cin/8][CoutP/32][64][4] floats:
//   wu_frag[xi][u][nt][lane][e] = U[xi][nt*32 + (lane & 31)][8 u + 4 (lane >> 5) + e]     (ops.pack_wino44_fragments), CoutP % 64 == 0
// gn_partial (optional, needs res == NULL and act == none): [B][lm_winograd44_gn_chunks][Cout][2] doubles -> lm_gn_finalize.
static int w44_launch(void* stream, const float* x, int ldx, const float* wu_frag, int CoutP, const float* scale,
                      const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                      int Cin, int Cout, int dil, int act, double* gn_partial, float split_post) {
    LM_REQUIRE(x && wu_frag && y, "conv_wino44: null pointer");
    LM_REQUIRE(lm_winograd44_supported(H, W, Cin, dil) && B > 0, "conv_wino44: unsupported shape (H=%d W=%d Cin=%d dil=%d)", H, W, Cin, dil);
    LM_REQUIRE(CoutP >= Cout && CoutP % QBN == 0, "conv_wino44: CoutP=%d must be Cout=%d rounded up to %d", CoutP, Cout, QBN);
    LM_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldy >= Cout, "conv_wino44: bad leading dimension");
    LM_REQUIRE(act == LM_ACT_NONE || act == LM_ACT_RELU, "conv_wino44: activation %d not supported", act);
    LM_REQUIRE(!gn_partial || (res == nullptr && act == LM_ACT_NONE && Cout % 4 == 0), "conv_wino44(gn stats): no residual / activation");
    W44Params p;
    p.g = geom44(B, H, W, dil);
    LM_REQUIRE((long)B * H * W * ldx < (1L << 40) && (long)B * H * W < (1L << 31) && p.g.T < (1L << 31), "conv_wino44: tensor too large");
    p.x = x; p.U = wu_frag; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.ldx = ldx; p.ldr = ldr; p.ldy = ldy; p.C = Cin; p.Cout = Cout; p.NT = CoutP / 32; p.act = act;
    p.gn_part = gn_partial;
    if (int e = w44_zeros(&p.zeros)) return e;
    // N tile inner: the N tiles of an M block run side by side on one XCD and share the input lines in its L2; U (2.4 MB per N tile at
    // Cin = 256) then streams from the Infinity Cache.  Measured at B = 16 against N outer: 256->256@288^2 4.06 vs 4.25 ms, 256->512@144^2
    // 2.08 vs 2.16, 256->256 d2@144^2 1.146 vs 1.162.  LANEMAP_W44_ORDER=0 selects N outer (experiments).
    static const int order = getenv("LANEMAP_W44_ORDER") ? atoi(getenv("LANEMAP_W44_ORDER")) : 1;
    p.n_inner = order;
    p.dnt = lm_fastdiv_make((unsigned)((Cout + QBN - 1) / QBN));
    const size_t lds = (size_t)(2 * QRAWF + QVF) * sizeof(float);
    static_assert(36 * 32 * 32 <= 2 * QRAWF + QVF, "the product buffer of the epilogue fits");
    const long blocks = (p.g.T / QBM) * ((Cout + QBN - 1) / QBN);
    LM_REQUIRE(blocks > 0 && blocks < (1L << 31) && p.g.T % QBM == 0, "conv_wino44: bad grid %ld", blocks);
    p.post = split_post != 0.f ? split_post : 1.f;
    if (split_post != 0.f) {
        if (int e = lm_ensure_dynamic_lds((const void*)wino44_kernel<true>, lds)) return e;
        hipLaunchKernelGGL(wino44_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    } else {
        if (int e = lm_ensure_dynamic_lds((const void*)wino44_kernel<false>, lds)) return e;
        hipLaunchKernelGGL(wino44_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    }
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_conv3x3_winograd44_f32(void* stream, const float* x, int ldx, const float* wu_frag, int CoutP, const float* scale,
                                     const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                     int Cin, int Cout, int dil, int act, double* gn_partial) {
    return w44_launch(stream, x, ldx, wu_frag, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, gn_partial, 0.f);
}

// SECOND LINE (never the headline path): the same convolution with the Winograd-domain products on the fp16 matrix pipe, every fp32
// operand split into two fp16 terms, three products per pair, fp32 accumulation (see w44_split2).  wu_split = lm_wino44_split_fragments
// of the fragments of U * u_scale (u_scale a power of two chosen by the caller so that max |U| u_scale ~ 2^13); post = 1 / u_scale.
LM_API int lm_conv3x3_winograd44_split_f32(void* stream, const float* x, int ldx, const float* wu_split, int CoutP, const float* scale,
                                           const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                           int Cin, int Cout, int dil, int act, double* gn_partial, float post) {
    LM_REQUIRE(post > 0.f, "conv_wino44(split): post scale must be positive");
    return w44_launch(stream, x, ldx, wu_split, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, gn_partial, post);
}

// fragments [n_quads][4] fp32 (ops.pack_wino44_fragments of the scaled U) -> the fp16 term words of the split kernel, same shape
LM_API int lm_wino44_split_fragments(void* stream, const float* frag, float* out, long n_quads) {
    LM_REQUIRE(frag && out && n_quads > 0 && n_quads < (1L << 38), "wino44_split_fragments: bad arguments");
    hipLaunchKernelGGL(wino44_split_frag_kernel, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(frag), reinterpret_cast<f32x4*>(out), n_quads);
    LM_LAUNCH_CHECK();
    return LM_OK;
}


// The same convolution through the materialising twin (bit-identical results; test infrastructure).  wu = U as [36][CoutP][Cin]
// (ops.pack_wino44); workspace >= lm_winograd44_twin_workspace_bytes.
static int w44_twin(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                    const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                    int Cin, int Cout, int dil, int act, void* workspace, long workspace_bytes, float split_post) {
    LM_REQUIRE(x && wu && y && workspace, "conv_wino44_twin: null pointer");
    LM_REQUIRE(Cin > 0 && Cin % 8 == 0 && dil >= 1 && B > 0 && H > 0 && W > 0, "conv_wino44_twin: bad shape");
    LM_REQUIRE(CoutP >= Cout && CoutP % 32 == 0 && ldx >= Cin && ldy >= Cout, "conv_wino44_twin: bad CoutP / leading dimension");
    LM_REQUIRE(act == LM_ACT_NONE || act == LM_ACT_RELU, "conv_wino44_twin: activation %d not supported", act);
    LM_REQUIRE(lm_winograd44_twin_workspace_bytes(B, H, W, Cin, CoutP, dil) <= workspace_bytes, "conv_wino44_twin: workspace too small");
    const W44Geom g = geom44(B, H, W, dil);
    LM_REQUIRE(g.T * (long)(Cin > Cout ? Cin : Cout) < (1L << 31) * 256L && g.T / 32 < 65536L * 32768L, "conv_wino44_twin: too large");
    float* V = (float*)workspace;
    float* M = V + 36 * g.T * Cin;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wino44_input_kernel, dim3((unsigned)((g.T * Cin + 255) / 256)), dim3(256), 0, s, x, ldx, g, Cin, V);
    LM_LAUNCH_CHECK();
    if (split_post != 0.f)
        hipLaunchKernelGGL(wino44_gemm_kernel<true>, dim3((unsigned)(g.T / 32), (unsigned)(CoutP / 32), 36), dim3(64), 0, s, (const float*)V, wu, M,
                           g.T, Cin, CoutP);
    else
        hipLaunchKernelGGL(wino44_gemm_kernel<false>, dim3((unsigned)(g.T / 32), (unsigned)(CoutP / 32), 36), dim3(64), 0, s, (const float*)V, wu, M,
                           g.T, Cin, CoutP);
    LM_LAUNCH_CHECK();
    W44Epi e;
    e.scale = scale; e.shift = shift; e.res = res; e.y = y; e.ldr = ldr; e.ldy = ldy; e.Cout = Cout; e.act = act;
    e.post = split_post;
    hipLaunchKernelGGL(wino44_output_kernel, dim3((unsigned)((g.T * Cout + 255) / 256)), dim3(256), 0, s, (const float*)M, g, CoutP, e);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_conv3x3_winograd44_twin_f32(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                                          const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                          int Cin, int Cout, int dil, int act, void* workspace, long workspace_bytes) {
    return w44_twin(stream, x, ldx, wu, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, workspace, workspace_bytes, 0.f);
}

// The split convolution through the materialising twin: wu = U * u_scale as [36][CoutP][Cin] fp32 (split on the fly), post = 1 / u_scale.
// Bit-identical to lm_conv3x3_winograd44_split_f32 (test infrastructure).
LM_API int lm_conv3x3_winograd44_split_twin_f32(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                                                const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                                int Cin, int Cout, int dil, int act, void* workspace, long workspace_bytes, float post) {
    LM_REQUIRE(post > 0.f, "conv_wino44_twin(split): post scale must be positive");
    return w44_twin(stream, x, ldx, wu, CoutP, scale, shift, res, ldr, y, ldy, B, H, W, Cin, Cout, dil, act, workspace, workspace_bytes, post);
}

// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores, exact fp32.
//
//   Y[m, n] = act( (sum_{tap,c} X[pix(m,tap), c] * Wp[tap][n][c]) * scale[n] + shift[n] + R[m % res_rows, n] )
//
// m indexes output pixels (b, oy, ox) of an NHWC tensor, n output channels, k = (tap, c).
// A plain GEMM is the 1x1 / H*W = M special case; the ViT patch embedding is the 8x8 stride-8 case.
//
// Replaces (reference, all fp32 cuDNN/cuBLAS via torch.nn): every Conv2d with Cin % 32 == 0 of
// FPNWrapper (baseline/models/pcencoder/postprojector.py:463-511, 563-655), every nn.Linear of
// VitSegNet (baseline/models/backbone/vitsegnet.py:51-56,32-35,165) and the shared Conv1d stack of
// ColumnProposal2 (baseline/models/heads/polyline_fpn_vit_vertex_2.py:206-228).
//
// Design (gfx950): 256 threads = 4 waves, one per SIMD; each wave owns a WM x WN output tile made of
// 32x32 accumulators driven by v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = chip f32 peak).
// K is walked one (tap, 32-channel) slab at a time.  The next slab goes global -> LDS directly
// (global_load_lds_dwordx4: no staging registers, no ds_write; ablation showed the register-staged ds_write pass
// alone cost 9 % of the MFMA issue slots), double-buffered, one barrier per slab.  The LDS image is lane-linear
// ([rows][32] floats, 8 lanes x 16 B per row), so bank conflicts are avoided by an XOR swizzle applied to the SOURCE
// address (lane p of row r fetches 16-byte chunk p ^ ((r>>1)&7)) and to the ds_read_b128 fragment address.
// One 16-byte read per operand row feeds 4 MFMAs (lanes 0-31 hold k..k+3, lanes 32-63 k+4..k+7).  Out-of-image taps
// (zero padding) fetch from a 16-byte zero block.  The summation order over k is fixed => deterministic results.
#include "common.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;         // k-slab: 32 input channels of one tap
constexpr int LDS_LD = BK;     // unpadded, lane-linear rows (floats): required by global_load_lds

__device__ __attribute__((aligned(16))) float g_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};   // source of zero-padding taps

struct ConvParams {
    const float* x; const float* wp; const float* scale; const float* shift; const float* res; float* y;
    int ldx, ldr, ldy, res_rows;
    int B, H, W, Cin, Cout, CoutP, Ho, Wo;
    int KH, KW, stride, pad_h, pad_w, dil, act;
    long M;
    const float* zero; // 16 zero bytes: source of padding taps / inactive sites (kernel argument: no GOT load in the loop)
    int taps_real;     // gather mode: taps per rulebook row (KW counts K slabs: taps, or tap pairs when TPS == 2)
    const int* nbr;    // gather mode (sparse convolution): [M][taps] input row of every (output row, tap), -1 = inactive site
    double* gn_part;   // optional [B][chunks][Cout][2] per-channel (sum, sum of squares) of the outputs of each wave tile
    int gn_chunks;     // chunks per batch element = (Ho*Wo / BM) * (BM / WM)
    int res_hi, res_wi; // > 0: `res` is a coarse [B][res_hi][res_wi][ldr] map added through bilinear (align_corners) interpolation
    float res_sy, res_sx;   // its source scales (res_hi - 1) / (Ho - 1), (res_wi - 1) / (Wo - 1): lm_bilin_axis's quotient, taken on the host
    LmFastDiv div_wo, div_ho, div_rr;   // output pixel -> (b, oy, ox) and m % res_rows without run-time divisions (M < 2^31)
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int BM, int BN, int WM, int WN, bool GATHER = false, int TPS = 1>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_mfma_kernel(ConvParams p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;   // threads: one wave per WM x WN sub-tile
    constexpr int RPP = NT / 8;                       // rows covered by one load pass (8 lanes x 16 B per row)
    constexpr int A_LOADS = BM / RPP;                 // float4 per thread per slab
    constexpr int B_LOADS = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the load pass");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;
    const int n_tiles = (p.Cout + BN - 1) / BN;   // CoutP only guarantees that weight rows up to the tile edge exist
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (private L2 each), so give every XCD one
    // contiguous run of tiles - neighbouring output rows (shared 3x3 halos) and the N tiles of one M tile share an L2.
    unsigned bid = blockIdx.x;
    {
        const unsigned nb = gridDim.x, per = nb / 8, full = per * 8;
        if (bid < full) bid = (bid % 8) * per + bid / 8;
    }
    const long m0 = (long)(bid / n_tiles) * BM;
    const int n0 = (bid % n_tiles) * BN;

    // --- per-thread gather coordinates (tap independent part) ---
    static_assert(RPP % 32 == 0, "one load pass covers whole 32-row swizzle periods");
    const int lrow = tid >> 3;                               // 0..31: row inside a load pass
    const int lc4 = (((tid & 7) ^ ((lrow >> 1) & 7))) * 4;   // swizzled source chunk (floats) for LDS slot tid & 7
    // Everything that does not depend on the tap is folded into one 32-bit element offset per row (the launcher checks
    // that the tensors stay below 2^31 elements), so a slab costs a handful of full-rate VALU ops per load: the previous
    // 64-bit multiply per (row, tap), two integer divisions and a GOT load of the zero block per slab sat in front of every
    // MFMA block.
    int a_iy0[A_LOADS], a_ix0[A_LOADS];
    int a_off[A_LOADS];                       // dense: ((b*H + iy0)*W + ix0)*ldx + lc4;  gather: output row or -1
    int b_off[B_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        long m = m0 + lrow + i * RPP;
        if (GATHER) {
            a_iy0[i] = 0;
            a_ix0[i] = 0;
            a_off[i] = m < p.M ? (int)m : -1;   // output row; its input rows come from the rulebook
        } else if (m < p.M) {
            // (set-up and epilogue index math is what bounds the tiny-K layers: two 64-bit divisions per row cost ~300 VALU instructions
            // in front of 32 MFMAs - profiles/README.md, round 4)
            const unsigned t = lm_fastdiv((unsigned)m, p.div_wo);
            const int ox = (int)((unsigned)m - t * p.div_wo.d);
            const int b = (int)lm_fastdiv(t, p.div_ho);
            const int oy = (int)(t - (unsigned)b * p.div_ho.d);
            a_iy0[i] = oy * p.stride - p.pad_h;
            a_ix0[i] = ox * p.stride - p.pad_w;
            a_off[i] = ((b * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.ldx + lc4;
        } else {
            a_iy0[i] = -(1 << 28);   // always out of range -> zero rows
            a_ix0[i] = 0;
            a_off[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) b_off[i] = (n0 + lrow + i * RPP) * p.Cin + lc4;
    const int cslabs = p.Cin / BK;
    const int KT = p.KH * p.KW * cslabs;
    const int taps = GATHER ? p.taps_real : p.KH * p.KW;   // row stride of the rulebook
    const float* const zero = p.zero;

    // load cursor: k order = (channel slab, tap), taps innermost, without divisions.  A 128-byte line is exactly one
    // (pixel, 32-channel slab): with the taps innermost the up to 9 uses of a line (by this tile and its neighbours) fall
    // within 9 consecutive slabs, so the 64 tiles resident on an XCD keep a ~3 MB working set that fits its 4 MB L2
    // (tap-major order spread the re-use over 24 slabs = 25 MB per XCD and sent every tap to the Infinity Cache).
    int cur_cs = 0, cur_kx = 0, cur_ky = 0, cur_tap = 0;
    int nb[A_LOADS];                      // gather mode: input rows of the slab that is loaded next
    int nb_cs = 0, nb_tap = 0;            // cursor of the rulebook prefetch (runs one slab ahead of the load cursor)
    auto nbr_fetch = [&]() {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            // TPS == 2 (16-channel features): a 32-float K slab holds two taps; lanes of chunks 4..7 fetch the odd one
            const int t = TPS == 2 ? 2 * nb_tap + (lc4 >> 4) : nb_tap;
            nb[i] = (a_off[i] >= 0 && t < taps) ? p.nbr[(long)a_off[i] * taps + t] : -1;
        }
        if (++nb_tap == p.KW) {
            nb_tap = 0;
            ++nb_cs;
        }
    };

    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    auto gload = [&](int buf) {   // global -> LDS, one 1 KiB (8 rows x 128 B) piece per wave-instruction
        const int dy = cur_ky * p.dil, dx = cur_kx * p.dil;
        const int cbase = cur_cs * BK;
        const int adelta = (dy * p.W + dx) * p.ldx + cbase;
        const int bdelta = cur_tap * p.CoutP * p.Cin + cbase;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const float* src;
            if (GATHER) {
                src = nb[i] >= 0 ? p.x + ((long)nb[i] * p.ldx + (TPS == 2 ? (lc4 & 15) : cbase + lc4)) : zero;
            } else {
                const bool ok = (unsigned)(a_iy0[i] + dy) < (unsigned)p.H && (unsigned)(a_ix0[i] + dx) < (unsigned)p.W;
                src = ok ? p.x + (a_off[i] + adelta) : zero;
            }
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(As + (buf * BM + i * RPP + wave * 8) * LDS_LD), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const float* src = p.wp + (b_off[i] + bdelta);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Bs + (buf * BN + i * RPP + wave * 8) * LDS_LD), 16, 0, 0);
        }
        ++cur_tap;
        if (++cur_kx == p.KW) {
            cur_kx = 0;
            if (++cur_ky == p.KH) {
                cur_ky = 0;
                cur_tap = 0;
                ++cur_cs;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (GATHER) nbr_fetch();
    gload(0);
    if (GATHER && KT > 1) nbr_fetch();
    __syncthreads();

    const int frow = lane & 31;
    const int fswz = (frow >> 1) & 7;          // wm0, wn0 and i*32 are multiples of 32: the swizzle depends on frow only
    const int fhalf = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(buf ^ 1);            // next slab lands in the other buffer under the MFMA block
        if (GATHER && kt + 2 < KT) nbr_fetch();     // rulebook entries of the slab after that (latency under the MFMAs)
        const float* Ab = As + (buf * BM + wm0 + frow) * LDS_LD;
        const float* Bb = Bs + (buf * BN + wn0 + frow) * LDS_LD;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            const int fo = (((kk >> 2) + fhalf) ^ fswz) * 4;   // physical position of logical 16-byte chunk kk/4 + half
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_LD + fo);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_LD + fo);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        __syncthreads();   // drains the global_load_lds queue (vmcnt 0) and fences the buffer swap
    }

    // --- epilogue: each wave transposes its WM x WN accumulator tile through (now idle) LDS so that every lane owns 4
    // consecutive output channels of one pixel: residual loads and output stores become coalesced 16-byte accesses
    // (the raw MFMA layout gives 4-byte stores, which made the thin 1x1 layers store-issue bound).
    constexpr int ELD = WN + 4;
    // (the launch sizes the dynamic LDS as max(K-loop buffers, 4 staging tiles))
    float* stage = smem + wave * (WM * ELD);
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ELD + j * 32 + frow] = acc[i][j][r];
    __syncthreads();
    constexpr int LPR = WN / 4;            // lanes per output row
    constexpr int RPI = 64 / LPR;          // rows per pass
    const int c4 = (lane % LPR) * 4;
    const int n = n0 + wn0 + c4;
    if (n < p.Cout) {
        const bool vec = (n + 3 < p.Cout) && ((p.ldy & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) {
                if (p.scale) sc[e] = p.scale[n + e];
                if (p.shift) sh[e] = p.shift[n + e];
            }
        f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};   // GroupNorm(C,C) statistics of this wave tile
        // Round 5: the 16-byte stores of a GROUP of passes go out together behind the group's loads and arithmetic.  One store per pass
        // put an `s_waitcnt vmcnt(0)` at the top of the next pass (loads and stores share vmcnt, and the residual loads of a pass are
        // pending on some path of the control-flow graph), i.e. every pass waited for the previous pass's store to be acknowledged:
        // eight serialised memory round trips per wave tile on the 1 x 1 laterals
        // (only the eight-wave tiles of the tiny-K layers, whose epilogue is the kernel: on the four-wave tiles the grouped stores cost the
        // K-heavy layers 30 % - 64 -> 128 3 x 3 / stride 2 0.50 -> 0.64 ms, the N = 3072 GEMM 0.46 -> 0.62 - measured)
        constexpr int NPASS = WM / RPI, GROUP = (BM / WM) * (BN / WN) == 8 ? (NPASS < 8 ? NPASS : 8) : 1;
        static_assert(NPASS % GROUP == 0, "whole store groups");
        f32x4 vout[GROUP];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = pass * RPI + lane / LPR;
            const long m = m0 + wm0 + row;
            if (m < p.M) {
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
            const long rrow = p.res_rows ? (long)((unsigned)m - lm_fastdiv((unsigned)m, p.div_rr) * p.div_rr.d) : m;
            if (vec) {
                if (p.res && p.res_hi) {
                    // residual = F.interpolate(coarse, size=(Ho, Wo), bilinear, align_corners=True)[m]: `_upsample_add` of the FPN
                    // (postprojector.py:549-561) without materialising the upsampled map; same fixed-order blend as
                    // lm_upsample_bilinear_nhwc (common.h), so the sum is bit-identical to adding that kernel's output
                    const unsigned mm = (unsigned)m;
                    const unsigned q = lm_fastdiv(mm, p.div_wo);
                    const int ox = (int)(mm - q * p.div_wo.d);
                    const int bi = (int)lm_fastdiv(q, p.div_ho);
                    const int oy = (int)(q - (unsigned)bi * p.div_ho.d);
                    int y0, y1, x0, x1;
                    float wy0, wy1, wx0, wx1;
                    lm_bilin_axis_scaled(oy, p.res_hi, p.res_sy, y0, y1, wy0, wy1);
                    lm_bilin_axis_scaled(ox, p.res_wi, p.res_sx, x0, x1, wx0, wx1);
                    const float* rb = p.res + (long)bi * p.res_hi * p.res_wi * p.ldr + n;
                    const f32x4 r00 = *reinterpret_cast<const f32x4*>(rb + ((long)y0 * p.res_wi + x0) * p.ldr);
                    const f32x4 r01 = *reinterpret_cast<const f32x4*>(rb + ((long)y0 * p.res_wi + x1) * p.ldr);
                    const f32x4 r10 = *reinterpret_cast<const f32x4*>(rb + ((long)y1 * p.res_wi + x0) * p.ldr);
                    const f32x4 r11 = *reinterpret_cast<const f32x4*>(rb + ((long)y1 * p.res_wi + x1) * p.ldr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += lm_bilerp(r00[e], r01[e], r10[e], r11[e], wy0, wy1, wx0, wx1);
                } else if (p.res) {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(p.res + rrow * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.act == LM_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                    else if (p.act == LM_ACT_GELU) v[e] = gelu_erf(v[e]);
                }
                vout[pass % GROUP] = v;
            } else {
                for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                    float u = v[e];
                    if (p.res) u += p.res[rrow * p.ldr + n + e];
                    if (p.act == LM_ACT_RELU) u = fmaxf(u, 0.f);
                    else if (p.act == LM_ACT_GELU) u = gelu_erf(u);
                    p.y[m * p.ldy + n + e] = u;
                }
            }
            }
            if (vec && pass % GROUP == GROUP - 1) {
#pragma unroll
                for (int q = 0; q < GROUP; ++q) {
                    const long ms = m0 + wm0 + (pass - (GROUP - 1) + q) * RPI + lane / LPR;
                    if (ms < p.M) *reinterpret_cast<f32x4*>(p.y + ms * p.ldy + n) = vout[q];
                }
            }
        }
        if (p.gn_part) {   // fixed-order reduction over the RPI lanes that share a channel quad, then one writer lane
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gs[e] += __shfl_xor(gs[e], o);
                    gq[e] += __shfl_xor(gq[e], o);
                }
            if (lane < LPR) {
                const long hw = (long)p.Ho * p.Wo;
                const long mt = m0 + wm0;                       // first row of this wave tile (tiles never straddle images)
                const long b = mt / hw;
                const long chunk = (mt - b * hw) / WM;
                double* o = p.gn_part + ((b * p.gn_chunks + chunk) * p.Cout + n) * 2;
                for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                    o[2 * e] = (double)gs[e];
                    o[2 * e + 1] = (double)gq[e];
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool GATHER = false, int TPS = 1>
int launch(const ConvParams& p, hipStream_t stream) {
    const size_t kloop = (size_t)2 * (BM + BN) * LDS_LD, stage = (size_t)((BM / WM) * (BN / WN)) * WM * (WN + 4);   // floats (one staging tile per wave)
    const size_t lds = (kloop > stage ? kloop : stage) * sizeof(float);
    if (int e = lm_ensure_dynamic_lds((const void*)conv_mfma_kernel<BM, BN, WM, WN, GATHER, TPS>, lds)) return e;
    const long m_tiles = (p.M + BM - 1) / BM;
    const long blocks = m_tiles * ((p.Cout + BN - 1) / BN);
    LM_REQUIRE(blocks > 0 && blocks < (1L << 31), "conv_mfma: bad grid %ld", blocks);
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, GATHER, TPS>), dim3((unsigned)blocks), dim3((BM / WM) * (BN / WN) * 64), lds, stream, p);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// ---- the FPN's 1x1 lateral convolutions (round 6) ---------------------------------------------------------------------------------
// latlayer1 / latlayer2 (postprojector.py:595-601: 128 -> 256 @144^2 + p4, 64 -> 256 @288^2 + bilinear(p3)) have K = 128 / 64: per output
// pixel 1 KB is written for 0.25 / 0.5 KB read, and the matrix time (2 K 256 FLOP per pixel at the fp32 MFMA rate) is of the same order
// as the HBM time.  The tiled kernel above runs them at a quarter of the HBM peak: its phases (weights + pixels global -> LDS, barrier,
// MFMA, barrier, LDS transposition, barrier, residual + stores) are serial inside a workgroup and two workgroups fit a CU.  Here:
//   * persistent 512-thread workgroups, one per CU; wave w owns output channels 32 w .. 32 w + 31 for the life of the workgroup and keeps
//     their weights in registers (the MFMA's A operand, K / 2 floats per lane) - no weight traffic after the first tile;
//   * the pixels of a 32-pixel tile (the B operand) go global -> LDS once for all eight waves (global_load_lds, one or two 1 KB pieces
//     per wave, the XOR chunk swizzle of conv_mfma_kernel), double-buffered, ONE barrier per tile; the next tile is requested right behind
//     the barrier, a tile period ahead of its use;
//   * with the operands swapped (weights = A) an aligned quad of accumulator registers holds four consecutive channels of one pixel: the
//     32 x 32 result is transposed through a WAVE-PRIVATE LDS patch with four ds_write_b128 + four ds_read_b128 and no barrier, after
//     which 8 lanes cover the 128 contiguous bytes of a pixel - residual loads and stores are whole lines;
//   * the residual (4 pixels x 4 taps per lane for the bilinear one) is requested before the tile's MFMAs; the stores of tile t are
//     still in flight while tile t + 1 computes (explicit s_waitcnt vmcnt(4): the compiler's own count would drain them at the barrier).
// Same k order as conv_mfma_kernel (8-channel slabs ascending, step t pairs channel 8 u + t with 8 u + 4 + t), same epilogue expressions
// (v + shift, + residual through lm_bilerp): bit-identical outputs (test_lateral_kernel_bit_identical); LM_CONV_LATERAL=0 switches it off.
// Needs B * H * W % 32 == 0 (no partial tiles: every wave issues the same memory instructions, which the explicit wait counts rely on).
constexpr int LAT_SLD = 36;                       // floats per row of the transposition patch (32 channels + 4 pad: conflict-free both ways)
template <int KS, bool RESUP, bool ROWTILE>       // ROWTILE: Wo % 32 == 0 - a tile lies in one image row (its batch index and row are wave-uniform)
__global__ __launch_bounds__(512) void lateral_mfma_kernel(ConvParams p) {
    constexpr int K = KS * 8, SLABS = K / 32, XBUF = 32 * K;       // floats per pixel buffer: [SLABS][32 rows][32], chunk-swizzled
    constexpr int NX = XBUF / 4 / 512;                             // 16-byte pieces per thread and tile (1 or 2)
    static_assert(NX >= 1 && XBUF % (4 * 512) == 0, "whole load passes");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // X[2][XBUF] | patch[8][32 * LAT_SLD]
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 32-channel block
    const int j = lane & 31, half = lane >> 5;
    float* const patch = smem + 2 * XBUF + wave * (32 * LAT_SLD);
    f32x4 wf[KS];
    {
        const float* wrow = p.wp + (long)(wave * 32 + j) * p.Cin + 4 * half;
#pragma unroll
        for (int k = 0; k < KS; ++k) wf[k] = *reinterpret_cast<const f32x4*>(wrow + 8 * k);
    }
    // epilogue role of this lane: pixels erow + 8 q (q = 0..3) of the tile, channels n .. n + 3
    const int erow = lane >> 3, n = wave * 32 + (lane & 7) * 4;
    const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned yoff[4], roff[4];                   // this lane's byte offsets inside a tile of the output / of a plain residual
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        yoff[q] = (unsigned)((erow + 8 * q) * p.ldy + n) * 4u;               // bytes
        roff[q] = (unsigned)((erow + 8 * q) * p.ldr + n) * 4u;
    }
    // tiles: every XCD (workgroups are dealt to the 8 XCDs round-robin) streams one contiguous eighth of the pixels, so that the coarse
    // rows two neighbouring output rows interpolate from meet in one L2
    const int ntiles = (int)(p.M >> 5);
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const int per = (ntiles + 7) >> 3, t_end = min(ntiles, (xcd + 1) * per);
    int tile = xcd * per + wi;
    if (tile >= t_end) return;                                     // (whole workgroup)
    // pixel loads: piece c = i * 512 + tid of a tile = slab c / 256, row (c % 256) / 8, LDS slot c % 8 <- source chunk slot ^ ((row >> 1) & 7)
    int xoff[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int c = i * 512 + tid, slab = c >> 8, row = (c & 255) >> 3, slot = c & 7;
        xoff[i] = row * p.ldx + slab * 32 + ((slot ^ ((row >> 1) & 7)) << 2);
    }
    auto load_x = [&](int t, int buf) {
        const float* xt = p.x + (long)t * 32 * p.ldx;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(xt + xoff[i]), (lptr_t*)(smem + buf * XBUF + (i * 8 + wave) * 256), 16, 0, 0);
    };
    load_x(tile, 0);
    const int fswz = (j >> 1) & 7;
    // the weights are complete before the loop (their first use is inside it, where the compiler's wait would also drain the pixel
    // prefetch of every later iteration)
#pragma unroll
    for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(wf[k]));
    for (int it = 0; tile < t_end; tile += nwg, ++it) {
        const int buf = it & 1;
        // tile's pixels have landed (everything but the four stores of the previous tile has), and every wave is done with the other buffer
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_x(min(tile + nwg, t_end - 1), buf ^ 1);               // (past the end: a harmless re-read, so that the counts stay uniform)
        // residual of this lane's four pixels (tile pixel erow + 8 q), requested before the MFMAs.  Everything that is the same for the
        // whole tile is computed on wave-uniform values (scalar ALU / once per wave), lane offsets are loop invariants: the epilogue's VALU
        // work is the kernel's second cost after the MFMAs (f32 MFMAs do not overlap the SIMD's own VALU instructions)
        f32x4 r00[4], r01[4], r10[4], r11[4];
        float wy0[4], wy1[4], wx0[4], wx1[4];
        if (RESUP && ROWTILE) {
            const unsigned mt = (unsigned)__builtin_amdgcn_readfirstlane(tile) * 32u;      // uniform
            const unsigned t = lm_fastdiv(mt, p.div_wo);
            const int ox0 = (int)(mt - t * p.div_wo.d);
            const int bi = (int)lm_fastdiv(t, p.div_ho);
            const int oy = (int)(t - (unsigned)bi * p.div_ho.d);
            int y0v, y1v;
            float wyav, wybv;
            lm_bilin_axis_scaled(oy, p.res_hi, p.res_sy, y0v, y1v, wyav, wybv);
            // (float arithmetic has no scalar unit: the uniform results are moved to scalar registers, so that the row pointers below are
            // scalar 64-bit values and every tap load takes the `scalar base + 32-bit lane offset` form - no 64-bit VALU address arithmetic)
            const int y0 = __builtin_amdgcn_readfirstlane(y0v), y1 = __builtin_amdgcn_readfirstlane(y1v);
            const float wya = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wyav)));
            const float wyb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wybv)));
            const char* row0 = reinterpret_cast<const char*>(p.res + ((long)bi * p.res_hi + y0) * p.res_wi * p.ldr);
            const char* row1 = reinterpret_cast<const char*>(p.res + ((long)bi * p.res_hi + y1) * p.res_wi * p.ldr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int x0, x1;
                lm_bilin_axis_scaled(ox0 + erow + 8 * q, p.res_wi, p.res_sx, x0, x1, wx0[q], wx1[q]);
                wy0[q] = wya;
                wy1[q] = wyb;
                const unsigned o0 = (unsigned)(x0 * p.ldr + n) * 4u, o1 = (unsigned)(x1 * p.ldr + n) * 4u;       // bytes (< 2^31: ldr * Wr * 4)
                r00[q] = *reinterpret_cast<const f32x4*>(row0 + o0);
                r01[q] = *reinterpret_cast<const f32x4*>(row0 + o1);
                r10[q] = *reinterpret_cast<const f32x4*>(row1 + o0);
                r11[q] = *reinterpret_cast<const f32x4*>(row1 + o1);
            }
        } else if (RESUP) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned m = (unsigned)(tile * 32 + erow + 8 * q);
                const unsigned t = lm_fastdiv(m, p.div_wo);
                const int ox = (int)(m - t * p.div_wo.d);
                const int bi = (int)lm_fastdiv(t, p.div_ho);
                const int oy = (int)(t - (unsigned)bi * p.div_ho.d);
                int y0, y1, x0, x1;
                lm_bilin_axis_scaled(oy, p.res_hi, p.res_sy, y0, y1, wy0[q], wy1[q]);
                lm_bilin_axis_scaled(ox, p.res_wi, p.res_sx, x0, x1, wx0[q], wx1[q]);
                const float* rb = p.res + (long)bi * p.res_hi * p.res_wi * p.ldr + n;
                r00[q] = *reinterpret_cast<const f32x4*>(rb + ((long)y0 * p.res_wi + x0) * p.ldr);
                r01[q] = *reinterpret_cast<const f32x4*>(rb + ((long)y0 * p.res_wi + x1) * p.ldr);
                r10[q] = *reinterpret_cast<const f32x4*>(rb + ((long)y1 * p.res_wi + x0) * p.ldr);
                r11[q] = *reinterpret_cast<const f32x4*>(rb + ((long)y1 * p.res_wi + x1) * p.ldr);
            }
        } else if (p.res_rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned m = (unsigned)(tile * 32 + erow + 8 * q);
                r00[q] = *reinterpret_cast<const f32x4*>(p.res + (long)(m - lm_fastdiv(m, p.div_rr) * p.div_rr.d) * p.ldr + n);
            }
        } else {
            const char* rt = reinterpret_cast<const char*>(p.res + (long)__builtin_amdgcn_readfirstlane(tile) * 32 * p.ldr);     // uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) r00[q] = *reinterpret_cast<const f32x4*>(rt + roff[q]);
        }
        // phases in issue order (the scheduler would otherwise sink the residual loads into the MFMA block and blend their values there
        // - fewer live registers, but a wait in front of every blend stops the MFMA issue)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* xb = smem + buf * XBUF + j * 32;
        // fragment k + 1 is read behind the first MFMA of fragment k (physical position of logical 16-byte chunk 2 (k % 4) + half of a row)
        f32x4 xf = *reinterpret_cast<const f32x4*>(xb + ((half ^ fswz) << 2));
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            f32x4 xn = xf;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k][0], xf[0], acc, 0, 0, 0);
            if (k + 1 < KS) {
                const int fo = (((((k + 1) & 3) << 1) + half) ^ fswz) << 2;
                __builtin_amdgcn_sched_barrier(0);
                xn = *reinterpret_cast<const f32x4*>(xb + ((k + 1) >> 2) * 1024 + fo);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 1; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k][t], xf[t], acc, 0, 0, 0);
            xf = xn;
        }
        __builtin_amdgcn_sched_barrier(0);
        // transposition: lane (pixel j, half) holds channels 8 g + 4 half .. + 3 of its pixel in acc[4 g .. 4 g + 3]
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(patch + j * LAT_SLD + 8 * g + 4 * half) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        char* const yt = reinterpret_cast<char*>(p.y + (long)__builtin_amdgcn_readfirstlane(tile) * 32 * p.ldy);               // uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(patch + (erow + 8 * q) * LAT_SLD + (lane & 7) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + sh[e];
            if (RESUP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += lm_bilerp(r00[q][e], r01[q][e], r10[q][e], r11[q][e], wy0[q], wy1[q], wx0[q], wx1[q]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r00[q][e];
            }
            *reinterpret_cast<f32x4*>(yt + yoff[q]) = v;
        }
    }
}

// 1 = launched (the shape is one of the FPN laterals), 0 = not covered (the caller takes the tiled kernel), < 0 = error code negated
int lateral_try(const ConvParams& p, hipStream_t stream) {
    static const int on = [] { const char* e = getenv("LM_CONV_LATERAL"); return e ? atoi(e) : 1; }();
    if (!on || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_h != 0 || p.pad_w != 0 || p.Cout != 256 || p.CoutP != 256) return 0;
    if ((p.Cin != 64 && p.Cin != 128) || p.scale || p.act != LM_ACT_NONE || p.gn_part || !p.res) return 0;
    if ((p.ldx & 3) || (p.ldy & 3) || (p.ldr & 3) || p.ldr < p.Cout || p.M >= (1L << 31) - 64 || (p.M & 31)) return 0;
    if (p.res_hi > 0 && p.Cin != 64) return 0;
    if (p.res_hi == 0 && p.Cin != 128) return 0;
    // persistent workgroups: two per CU would need <= 128 registers per lane; one (8 waves, two per SIMD) holds the weights + a tile in flight
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -LM_ERR_HIP;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ntiles = (int)((p.M + 31) / 32);
    int grid = (cus + 7) / 8 * 8;
    if (grid > (ntiles + 7) / 8 * 8) grid = (ntiles + 7) / 8 * 8;
    const size_t lds = (size_t)(2 * 32 * p.Cin + 8 * 32 * LAT_SLD) * sizeof(float);
#define LM_LAT_LAUNCH(KS, UP, ROW)                                                                                      \
    {                                                                                                                   \
        if (lm_ensure_dynamic_lds((const void*)lateral_mfma_kernel<KS, UP, ROW>, lds)) return -LM_ERR_HIP;              \
        hipLaunchKernelGGL((lateral_mfma_kernel<KS, UP, ROW>), dim3(grid), dim3(512), lds, stream, p);                  \
    }
    if (p.res_hi > 0 && p.Wo % 32 == 0) LM_LAT_LAUNCH(8, true, true)
    else if (p.res_hi > 0) LM_LAT_LAUNCH(8, true, false)
    else LM_LAT_LAUNCH(16, false, false)
#undef LM_LAT_LAUNCH
    if (hipGetLastError() != hipSuccess) return -LM_ERR_HIP;
    return 1;
}

int zero_block(const float** out) {   // device address of the 16 zero bytes, resolved once
    static const float* ptr = nullptr;
    if (!ptr) {
        void* sym = nullptr;
        LM_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_zero_chunk)));
        ptr = (const float*)sym;
    }
    *out = ptr;
    return LM_OK;
}

}  // namespace

static int conv_dispatch(void* stream, const float* x, int ldx, const float* wp, int CoutP, const float* scale, const float* shift,
                         const float* res, int ldr, int res_rows, float* y, int ldy, int B, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad_h, int pad_w, int dil, int act, double* gn_part, int res_hi = 0,
                         int res_wi = 0) {
    LM_REQUIRE(x && wp && y, "conv_mfma: null pointer");
    LM_REQUIRE(Cin > 0 && Cin % BK == 0, "conv_mfma: Cin=%d must be a multiple of %d", Cin, BK);
    LM_REQUIRE(CoutP >= Cout && CoutP % 128 == 0, "conv_mfma: CoutP=%d must be Cout=%d rounded up to 128", CoutP, Cout);
    LM_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldy >= Cout, "conv_mfma: bad leading dims ldx=%d ldy=%d", ldx, ldy);
    LM_REQUIRE(stride >= 1 && dil >= 1 && KH >= 1 && KW >= 1, "conv_mfma: bad geometry");
    ConvParams p;
    p.x = x; p.wp = wp; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.ldx = ldx; p.ldr = ldr; p.ldy = ldy; p.res_rows = res_rows;
    p.res_hi = res_hi; p.res_wi = res_wi;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.CoutP = CoutP;
    p.Ho = (H + 2 * pad_h - dil * (KH - 1) - 1) / stride + 1;
    p.Wo = (W + 2 * pad_w - dil * (KW - 1) - 1) / stride + 1;
    LM_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv_mfma: empty output");
    p.KH = KH; p.KW = KW; p.stride = stride; p.pad_h = pad_h; p.pad_w = pad_w; p.dil = dil; p.act = act;
    p.M = (long)B * p.Ho * p.Wo;
    LM_REQUIRE(((long)B * H * W + (long)(KH * dil + pad_h + 1) * W) * ldx < (1L << 31) && (long)KH * KW * CoutP * Cin < (1L << 31),
               "conv_mfma: tensor too large for 32-bit element offsets (B*H*W*ld = %ld)", (long)B * H * W * ldx);
    p.gn_part = gn_part;
    p.gn_chunks = 0;
    p.nbr = nullptr;
    p.taps_real = 0;
    LM_REQUIRE(p.M < (1L << 31) && res_rows >= 0, "conv_mfma: %ld output pixels do not fit 32-bit indices", p.M);
    p.div_wo = lm_fastdiv_make((unsigned)p.Wo);
    p.div_ho = lm_fastdiv_make((unsigned)p.Ho);
    p.div_rr = lm_fastdiv_make((unsigned)(res_rows > 0 ? res_rows : 1));
    p.res_sy = (res_hi > 0 && p.Ho > 1) ? (float)(res_hi - 1) / (float)(p.Ho - 1) : 0.f;
    p.res_sx = (res_wi > 0 && p.Wo > 1) ? (float)(res_wi - 1) / (float)(p.Wo - 1) : 0.f;
    if (int e = zero_block(&p.zero)) return e;
    hipStream_t s = (hipStream_t)stream;
    {   // the FPN laterals (K = 64 / 128, 256 outputs, residual): the streaming kernel
        const int r = lateral_try(p, s);
        if (r < 0) {
            lm_set_error("conv_mfma: lateral kernel launch failed");
            return -r;
        }
        if (r > 0) return LM_OK;
    }
    if (gn_part) {   // statistics mode: 128x128 tiles of 64-row wave tiles, images must be whole numbers of tiles
        LM_REQUIRE(Cout > 64 && Cout % 4 == 0 && ((long)p.Ho * p.Wo) % 128 == 0 && res == nullptr && act == LM_ACT_NONE,
                   "conv_mfma(gn stats): needs Cout > 64, Ho*Wo %% 128 == 0, no residual / activation");
        p.gn_chunks = (int)((long)p.Ho * p.Wo / 64);
        return launch<128, 128, 64, 64>(p, s);
    }
    if (Cout <= 64) return launch<128, 64, 32, 64>(p, s);
    {   // tiny-K layers (the FPN's 1x1 lateral / downsample convolutions, K = KH*KW*Cin <= 256) are epilogue- and HBM-bound: small wave
        // tiles (more waves per output) run them 15-25 % faster than four 64 x 64 wave tiles (0.599 -> 0.514 ms for 64->256 @288^2, B = 8)
        static const long tiny_k = [] { const char* e = getenv("LM_CONV_TINYK"); return e ? atol(e) : 256L; }();
        // round 5: 128 x 128 tiles of EIGHT waves (32 x 64 sub-tiles, 4 accumulators per wave) beat the 64 x 64 / four-wave tiles of round 4 by
        // 3-8 % on the laterals (64->256 @288^2 + upsample_add 1.021 -> 0.986 ms, 128->256 @144^2 0.336 -> 0.310, 256->256 @144^2 0.432 ->
        // 0.396 at B = 16; 64 x 128, 64 x 256 and 128 x 256 tiles measured between); the k order of an output does not depend on the tile
        // (69,632 B of dynamic LDS for the eight staging tiles and 512 threads: fine on gfx950's 160 KB - the only target of this library;
        //  lm_ensure_dynamic_lds reports the error on a device that cannot grant it, there is deliberately no second tile shape to fall back to)
        if ((long)KH * KW * Cin <= tiny_k) return launch<128, 128, 32, 64>(p, s);
    }
    // small-M GEMMs (ViT tokens): 128x128 tiles would leave most of the 256 CUs idle -> 64x64 tiles, 4x the workgroups
    const long big_blocks = ((p.M + 127) / 128) * ((Cout + 127) / 128);
    // (round 5: < 700 instead of < 512 - M = 5184 x N = 2048, 656 big tiles on 256 CUs = 2.56 rounds, runs 0.418 -> 0.353 ms per three launches on
    // 64 x 64 tiles (78 -> 92 TFLOP/s); N = 3072 (984 big tiles, 3.84 rounds) is better left on 128 x 128: 0.463 vs 0.476.  LM_CONV_SMALLM overrides)
    static const long small_m = [] { const char* e = getenv("LM_CONV_SMALLM"); return e ? atol(e) : 700L; }();
    if (big_blocks < small_m) return launch<64, 64, 32, 32>(p, s);
#ifdef LM_CONV_8WAVES
    return launch<128, 128, 64, 32>(p, s);
#else
    return launch<128, 128, 64, 64>(p, s);
#endif
}

LM_API int lm_conv2d_nhwc_mfma_f32(void* stream, const float* x, int ldx, const float* wp, int CoutP,
                                   const float* scale, const float* shift,
                                   const float* res, int ldr, int res_rows, float* y, int ldy,
                                   int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                   int stride, int pad_h, int pad_w, int dil, int act) {
    return conv_dispatch(stream, x, ldx, wp, CoutP, scale, shift, res, ldr, res_rows, y, ldy, B, H, W, Cin, Cout, KH, KW, stride,
                         pad_h, pad_w, dil, act, nullptr);
}

// Same convolution, additionally emitting the per-(image, 64-row chunk, channel) sum / sum of squares of its outputs:
// the first pass of GroupNorm(C groups == C channels) (postprojector.py:512-515) without re-reading the tensor.
// gn_partial: [B][Ho*Wo/64][Cout][2] doubles; finish with lm_gn_finalize.
LM_API int lm_conv2d_nhwc_mfma_f32_gnstats(void* stream, const float* x, int ldx, const float* wp, int CoutP, const float* shift,
                                           float* y, int ldy, double* gn_partial, int B, int H, int W, int Cin, int Cout,
                                           int KH, int KW, int stride, int pad_h, int pad_w, int dil) {
    LM_REQUIRE(gn_partial, "conv_mfma(gn stats): null partial buffer");
    return conv_dispatch(stream, x, ldx, wp, CoutP, nullptr, shift, nullptr, 0, 0, y, ldy, B, H, W, Cin, Cout, KH, KW, stride,
                         pad_h, pad_w, dil, LM_ACT_NONE, gn_partial);
}

// Convolution whose residual is a COARSE map added through bilinear interpolation (align_corners=True): the FPN's
// `_upsample_add(p_coarse, latlayer(c))` (postprojector.py:549-561, 595-601) in one kernel - the upsampled map is never written.
// res_coarse: [B][Hr][Wr][ldr]; vector path only (Cout, ldy, ldr multiples of 4).  Bit-identical to lm_upsample_bilinear_nhwc
// followed by lm_conv2d_nhwc_mfma_f32 with that map as the residual.
LM_API int lm_conv2d_nhwc_mfma_resup_f32(void* stream, const float* x, int ldx, const float* wp, int CoutP, const float* scale,
                                         const float* shift, const float* res_coarse, int ldr, int Hr, int Wr, float* y, int ldy,
                                         int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                                         int dil, int act) {
    LM_REQUIRE(res_coarse && Hr > 0 && Wr > 0, "conv_mfma(resup): missing coarse residual");
    LM_REQUIRE(Cout % 4 == 0 && ldy % 4 == 0 && ldr % 4 == 0 && ldr >= Cout, "conv_mfma(resup): Cout=%d, ldy=%d, ldr=%d must be multiples of 4", Cout, ldy, ldr);
    const long Ho = (H + 2 * pad_h - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad_w - dil * (KW - 1) - 1) / stride + 1;
    LM_REQUIRE((long)B * Ho * Wo < (1L << 31), "conv_mfma(resup): too many output pixels for 32-bit indices");
    return conv_dispatch(stream, x, ldx, wp, CoutP, scale, shift, res_coarse, ldr, 0, y, ldy, B, H, W, Cin, Cout, KH, KW, stride,
                         pad_h, pad_w, dil, act, nullptr, Hr, Wr);
}

// Sparse (rulebook) convolution on the same MFMA pipeline: output row m accumulates, for every kernel tap t, the feature row
// nbr[m][t] of x (skipped when -1).  Covers spconv's SubMConv3d and SparseConv3d as used by mmdet3d's SparseEncoder, which
// the reference's LidarEncoder instantiates (baseline/models/pcencoder/lidarencoder.py:29-35,93-102); the rulebook comes
// from lm_sparse_rulebook (lidar.hip).  Epilogue = BatchNorm1d(eval) scale/shift, optional residual rows, ReLU.
// Weights: Cin % 32 == 0 -> [taps][CoutP][Cin];  Cin == 16 -> tap pairs [ceil(taps/2)][CoutP][32] (k = (tap & 1) * 16 + c,
// the odd half of the last pair zero) so that a 32-float K slab of the MFMA pipeline carries two taps of 16 channels.
LM_API int lm_conv_gather_mfma_f32(void* stream, const float* x, int ldx, const int* nbr, int taps, const float* wp, int CoutP,
                                   const float* scale, const float* shift, const float* res, int ldr, float* y, int ldy,
                                   long M, int Cin, int Cout, int act) {
    LM_REQUIRE(x && nbr && wp && y, "conv_gather: null pointer");
    LM_REQUIRE(M > 0 && taps >= 1, "conv_gather: empty problem (M=%ld taps=%d)", M, taps);
    LM_REQUIRE(Cin == 16 || (Cin > 0 && Cin % BK == 0), "conv_gather: Cin=%d must be 16 or a multiple of %d (zero-pad the feature rows)",
               Cin, BK);
    LM_REQUIRE(CoutP >= Cout && CoutP % 128 == 0, "conv_gather: CoutP=%d must be Cout=%d rounded up to 128", CoutP, Cout);
    LM_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldy >= Cout, "conv_gather: bad leading dims ldx=%d ldy=%d", ldx, ldy);
    const bool pairs = Cin == 16;                     // two taps share one 32-float K slab
    const int slabs = pairs ? (taps + 1) / 2 : taps;
    ConvParams p;
    p.x = x; p.wp = wp; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.ldx = ldx; p.ldr = ldr; p.ldy = ldy; p.res_rows = 0;
    p.res_hi = 0; p.res_wi = 0;
    p.B = 1; p.H = 1; p.W = 1; p.Cin = pairs ? BK : Cin; p.Cout = Cout; p.CoutP = CoutP; p.Ho = 1; p.Wo = 1;
    p.KH = 1; p.KW = slabs; p.stride = 1; p.pad_h = 0; p.pad_w = 0; p.dil = 1; p.act = act;
    p.M = M;
    LM_REQUIRE(M < (1L << 31) && (long)slabs * CoutP * p.Cin < (1L << 31), "conv_gather: problem too large (M=%ld)", M);
    p.gn_part = nullptr;
    p.gn_chunks = 0;
    p.div_wo = p.div_ho = p.div_rr = lm_fastdiv_make(1);
    p.res_sy = p.res_sx = 0.f;
    p.nbr = nbr;
    p.taps_real = taps;
    if (int e = zero_block(&p.zero)) return e;
    hipStream_t s = (hipStream_t)stream;
    if (pairs) {
        if (Cout <= 32) return launch<128, 32, 32, 32, true, 2>(p, s);
        return launch<128, 64, 32, 64, true, 2>(p, s);
    }
    if (Cout <= 32) return launch<128, 32, 32, 32, true>(p, s);
    if (Cout <= 64) return launch<128, 64, 32, 64, true>(p, s);
    return launch<128, 128, 64, 64, true>(p, s);
}

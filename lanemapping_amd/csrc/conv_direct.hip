// Direct (VALU) convolutions for the layers that are too thin for the matrix cores, plus max-pool.
//
//  lm_stem_conv7x7_bn_relu : FPNWrapper.conv1 + bn1 + relu      (postprojector.py:458-460,566)  3 -> 64, 7x7 s2 p3
//  lm_maxpool3x3s2_nhwc    : FPNWrapper.maxpool                  (postprojector.py:461,567)
//  lm_conv2d_nhwc_small    : feature_layer / output_layer_* 1x1  (postprojector.py:509-511,628-651),
//                            head_common_layers, orient, bi_seg_proposal
//                            (heads/polyline_fpn_vit_vertex_2.py:183-189,232-237,249)
#include "common.h"

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---------------------------------------------------------------------------------------------
// stem: x [B,3,H,W] planar (the reference's `proj` tensor) -> y [B,Ho,Wo,64] NHWC, y = relu(conv*scale+shift)
// block = 16x16 output pixels, every thread owns one pixel x 64 channels.
// ---------------------------------------------------------------------------------------------
constexpr int ST = 16;             // tile edge (outputs)
constexpr int SP = 2 * ST + 5;     // input patch edge = 37
constexpr int SPP = 40;            // padded patch row

// U8: the tile comes as u8 HWC (3 channels: what the rasteriser / the PNG reader produce) and u8 / 255 - the reference's to_tensor,
// laserlane_proposals.py:85-98 - is applied while staging the patch: same bits as the f32 planar input, a quarter of the bytes.
template <bool U8>
__global__ __launch_bounds__(256) void stem_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ y, int H, int W, int Ho, int Wo) {
    // The weights are wave-uniform: they are read with SCALAR loads (uniform index into a __restrict__ const pointer -> s_load into
    // SGPRs, served by the scalar cache) and enter the packed FMAs as scalar operands: 74 VGPRs, 6 waves per SIMD.  Staging them in
    // LDS made the kernel LDS-issue bound (16 broadcast ds_read_b128 per 64 FMAs and lane, 8 waves per CU sharing one LDS, 254 VGPRs):
    // 0.90 -> 0.69 ms for 8 tiles.  Two pixels per thread (half the scalar traffic per FMA, 141 VGPRs) measured slower: 0.87 ms.
    __shared__ float in[3][SP][SPP];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * ST, ox0 = blockIdx.x * ST;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    for (int i = tid; i < 3 * SP * SP; i += 256) {
        const int c = i / (SP * SP), r = (i / SP) % SP, q = i % SP;
        const int iy = iy0 + r, ix = ix0 + q;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            if (U8) v = (float)static_cast<const unsigned char*>(xin)[(((long)b * H + iy) * W + ix) * 3 + c] / 255.0f;
            else v = static_cast<const float*>(xin)[((long)(b * 3 + c) * H + iy) * W + ix];
        }
        in[c][r][q] = v;
    }
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = in[c][2 * py + ky][2 * px + kx];
                const float* wr = w + ((ky * 7 + kx) * 3 + c) * 64;
#pragma unroll
                for (int q = 0; q < 64; ++q) acc[q] = fmaf(v, wr[q], acc[q]);
            }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < Ho && ox < Wo) {
        f32x4* out = reinterpret_cast<f32x4*>(y + (((long)b * Ho + oy) * Wo + ox) * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(acc[4 * q + e] * scale[4 * q + e] + shift[4 * q + e], 0.f);
            out[q] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same stem on the matrix cores (round 4).  The VALU kernel above reads 256 B of weights per (tap, channel) and wave through the
// scalar cache and waits for them in front of every 32 packed FMAs: it runs at ~0.55 of the packed-fp32 peak (the scalar cache is shared
// by neighbouring CUs).  v_mfma_f32_32x32x2_f32 has the same peak and takes both operands from LDS with lane-linear ds_read_b32: the
// weights sit there in fragment order (39 KB, staged once per persistent workgroup), the pixels in the channel-interleaved patch.
//   k'' = ky * 22 + (kx * 3 + c): the 21 (kx, c) of a patch row are consecutive floats of the channel-interleaved LDS patch, index 21 is a
//   pad with zero weight (fma(x, 0, acc) = acc exactly), so the two k of an MFMA (lanes 0-31 / 32-63) are always NEIGHBOURS in LDS: the
//   lane's base address carries the + 1, every read is base + immediate.  K'' = 154 -> 77 k pairs.
// The accumulation is the same fmaf chain in the same (ky, kx, c) order as the VALU kernel (the MFMA adds its two products in k order):
// bit-identical outputs (test_stem_mfma_bit_identical_to_valu).  Persistent workgroups (weights loaded once), two per CU.
// ---------------------------------------------------------------------------------------------
constexpr int SMW = 112;            // floats per patch row: 37 pixels x 3 channels + 1 pad
constexpr int SMJ = 77;             // k pairs
typedef float f32x16s __attribute__((ext_vector_type(16)));

template <bool U8>
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, int H, int W, int Ho, int Wo, LmFastDiv div_tx, LmFastDiv div_ty,
                                                           int ntiles) {
    __shared__ float in[SP * SMW];
    __shared__ float wl[SMJ * 2 * 64];                             // B operands in fragment order [k pair][channel block][lane]
    __shared__ float u8lut[256];                                   // u8 / 255 (the reference's to_tensor) as a table: the IEEE division costs ~10
    u8lut[threadIdx.x] = (float)threadIdx.x / 255.0f;              // VALU instructions per element, and fp32 VALU time is MFMA time here
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    // weights: B operand of k pair j and channel block nb, lane l = w''[2 j + l / 32][32 nb + l % 32]; staged once per (persistent) workgroup
    for (int i = tid; i < SMJ * 2 * 64; i += 256) {
        const int j = i >> 7, nb = (i >> 6) & 1, l = i & 63;
        const int ky = (2 * j) / 22, idx = (2 * j) % 22 + (l >> 5);
        wl[i] = idx == 21 ? 0.f : w[(ky * 21 + idx) * 64 + nb * 32 + (l & 31)];
    }
    float sc[2], sh[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        sc[nb] = scale[nb * 32 + l32];
        sh[nb] = shift[nb * 32 + l32];
    }
    // this lane's pixels: M block mb of the wave = tile rows 4 wave + 2 mb, + 1; pixel l32 of it = (row l32 / 16, column l32 % 16)
    int abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) abase[mb] = (2 * (4 * wave + 2 * mb + (l32 >> 4))) * SMW + (2 * (l32 & 15)) * 3 + half;
    // patch staging: element k of this thread is patch cell i = tid + 256 k = (row r, column q, channel c), the same for every tile:
    // source offset relative to the tile's first patch pixel and (r, q) for the bounds are made once
    constexpr int NST = (SP * SMW + 255) / 256;
    int poff[NST], prq[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 256;
        const int r = i / SMW, e = i - r * SMW;
        const int q = e / 3, c = e - q * 3;
        const bool cell = i < SP * SMW && q < SP;                  // (the pad column and the tail of the last round hold zeros)
        poff[k] = U8 ? (r * W + q) * 3 + c : c * H * W + r * W + q;
        prq[k] = cell ? (r << 16) | q : -1;
    }
    auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
        const unsigned row = lm_fastdiv((unsigned)t, div_tx);       // tile row over the whole batch
        b = (int)lm_fastdiv(row, div_ty);
        oy0 = (int)(row - (unsigned)b * div_ty.d) * ST;
        ox0 = (int)((unsigned)t - row * div_tx.d) * ST;
    };
    unsigned raw[NST], okmask = 0;                                 // loaded values, untouched until they are written to LDS: a conversion
    auto fetch = [&](int t) {                                      // or select right behind the load would put the wait in front of the MFMAs
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
        const long base = U8 ? (((long)b * H + iy0) * W + ix0) * 3 : ((long)b * 3 * H + iy0) * W + ix0;
        okmask = 0;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            // (unconditional load from a clamped address, select later: a load under a branch makes the compiler wait at the join)
            const int r = prq[k] >> 16, q = prq[k] & 0xffff;
            const bool ok = (prq[k] >= 0) & ((unsigned)(iy0 + r) < (unsigned)H) & ((unsigned)(ix0 + q) < (unsigned)W);
            okmask |= ok ? 1u << k : 0u;
            const long src = ok ? base + poff[k] : 0;
            if (U8) raw[k] = static_cast<const unsigned char*>(xin)[src];
            else raw[k] = static_cast<const unsigned*>(xin)[src];
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) fetch(t);
    for (; t < ntiles; t += gridDim.x) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        __syncthreads();                                           // the previous tile's reads are done
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 256;
            const float val = U8 ? u8lut[raw[k] & 255u] : __uint_as_float(raw[k]);
            if (i < SP * SMW) in[i] = (okmask >> k) & 1u ? val : 0.f;
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);     // the next tile's patch travels under this tile's MFMAs
        f32x16s acc[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
        // operands of k pair j + 1 are read from LDS BEFORE the four MFMAs of pair j are issued (the compiler left to itself reads them right
        // in front of their use and waits ~2 LDS round trips per 16 MFMAs)
        float av[2][2], wv[2][2];
        auto rd = [&](int j, float (&a)[2], float (&wq)[2]) {
            const int koff = ((2 * j) / 22) * SMW + (2 * j) % 22;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) a[mb] = in[abase[mb] + koff];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wq[nb] = wl[(j * 2 + nb) * 64 + lane];
        };
        rd(0, av[0], wv[0]);
#pragma unroll
        for (int j = 0; j < SMJ; ++j) {
            if (j + 1 < SMJ) rd(j + 1, av[(j + 1) & 1], wv[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][mb], wv[j & 1][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // accumulator register r of (mb, nb): pixel (r % 4) + 8 (r / 4) + 4 half of the M block, channel 32 nb + l32, i.e. tile row
        // 4 wave + 2 mb + (r >> 3), tile column (r & 3) + 8 ((r >> 2) & 1) + 4 half.  Wave-uniform base + 32-bit lane offset; tiles inside
        // the image (all but the last row / column of tiles) store without per-element bounds
        float* const yt = y + (((long)b * Ho + oy0) * Wo + ox0) * 64;
        const unsigned lbyte = (unsigned)((4 * wave * Wo + 4 * half) * 64 + l32) << 2;
        auto store_tile = [&](auto guarded) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ry = 2 * mb + (r >> 3), cx = (r & 3) + 8 * ((r >> 2) & 1);
                    // (wave-uniform row pointer + this lane's 32-bit byte offset: scalar base / vector offset stores, no address VALU)
                    char* const rowp = reinterpret_cast<char*>(yt) + (long)((ry * Wo + cx) * 64) * 4;
                    if (!decltype(guarded)::value || (oy0 + 4 * wave + ry < Ho && ox0 + cx + 4 * half < Wo)) {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            *reinterpret_cast<float*>(rowp + nb * 128 + lbyte) = fmaxf(acc[mb][nb][r] * sc[nb] + sh[nb], 0.f);
                    }
                }
        };
        if (oy0 + ST <= Ho && ox0 + ST <= Wo) store_tile(std::false_type{});      // (wave-uniform: one branch per tile, straight-line stores)
        else store_tile(std::true_type{});
    }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int H, int W, int C, int Ho, int Wo, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = C / 4;
    const int c = (int)(i % c4);
    long t = i / c4;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iy = oy * 2 - 1 + dy, ix = ox * 2 - 1 + dx;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + iy) * W + ix) * C + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
}

// ---------------------------------------------------------------------------------------------
// small conv: one thread per output pixel, all Cout (<= 16) channels in registers, weights in LDS
// as [tap][cin][16].  y = act(conv(pre_relu ? relu(x) : x) * scale + shift)
// ---------------------------------------------------------------------------------------------
struct SmallConvParams {
    const float* x; const float* w; const float* scale; const float* shift; float* y;
    int ldx, ldy, B, H, W, Cin, Cout, Ho, Wo, KH, KW, stride, pad_h, pad_w, pre_relu, act;
    long M;
};

__global__ __launch_bounds__(256) void small_conv_kernel(SmallConvParams p) {
    // weights [taps*Cin][16] are wave-uniform: scalar loads (see stem_kernel), no LDS staging
    const float* __restrict__ wl = p.w;
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= p.M) return;
    const int ox = (int)(m % p.Wo);
    long t = m / p.Wo;
    const int oy = (int)(t % p.Ho);
    const int b = (int)(t / p.Ho);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int ky = 0; ky < p.KH; ++ky) {
        const int iy = oy * p.stride - p.pad_h + ky;
        if ((unsigned)iy >= (unsigned)p.H) continue;
        for (int kx = 0; kx < p.KW; ++kx) {
            const int ix = ox * p.stride - p.pad_w + kx;
            if ((unsigned)ix >= (unsigned)p.W) continue;
            const float* xp = p.x + (((long)b * p.H + iy) * p.W + ix) * p.ldx;
            const float* wt = wl + (ky * p.KW + kx) * p.Cin * 16;
            for (int c = 0; c < p.Cin; c += 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xp + c);
                if (p.pre_relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* wr = wt + (c + e) * 16;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = fmaf(v[e], wr[q], acc[q]);
                }
            }
        }
    }
    float* yp = p.y + m * p.ldy;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        if (n < p.Cout) {
            float v = acc[n];
            if (p.scale) v *= p.scale[n];
            if (p.shift) v += p.shift[n];
            if (p.act == LM_ACT_RELU) v = fmaxf(v, 0.f);
            yp[n] = v;
        }
    }
}

template <bool U8>
int launch_stem(void* stream, const void* x, const float* w_k64, const float* scale, const float* shift, float* y, int B, int H, int W, int Ho,
                int Wo) {
    // LM_STEM_VALU=1: the VALU kernel (one workgroup per 16 x 16 tile); default: the MFMA kernel, persistent workgroups
    static const bool valu = [] { const char* e = getenv("LM_STEM_VALU"); return e && atoi(e) != 0; }();
    const int tx = lm_cdiv(Wo, ST), ty = lm_cdiv(Ho, ST);
    if (valu) {
        hipLaunchKernelGGL(stem_kernel<U8>, dim3(tx, ty, B), dim3(256), 0, (hipStream_t)stream, x, w_k64, scale, shift, y, H, W, Ho, Wo);
    } else {
        const long ntiles = (long)tx * ty * B;
        LM_REQUIRE(ntiles < (1L << 31), "stem: too many tiles");
        const int grid = (int)(ntiles < 512 ? ntiles : 512);          // two resident workgroups per CU
        hipLaunchKernelGGL(stem_mfma_kernel<U8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w_k64, scale, shift, y, H, W, Ho, Wo,
                           lm_fastdiv_make((unsigned)tx), lm_fastdiv_make((unsigned)ty), (int)ntiles);
    }
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// ---------------------------------------------------------------------------------------------
// 3x3 convolutions with 16 input channels and <= 16 output channels on the matrix cores (round 4): head_common_layers / orient of
// ColumnProposal2 (heads/polyline_fpn_vit_vertex_2.py:183-189,232-237; 16 -> 16 @288^2 stride 1 and 2, 16 -> 8 @144^2).  The VALU kernel
// above streams 1 KB of weights per tap and wave through the scalar cache and waits for every 16-byte input load: ~0.4 of the packed
// rate.  Here: v_mfma_f32_16x16x4_f32, M = 16 pixels of an output row, N = 16 output channels, K = 4 input channels; the 36 weight
// fragments of the 3 x 3 x 16 reduction stay in VGPRs (persistent workgroups), the input patch of a 16 x 16 output tile sits in LDS with
// the channels of a pixel permuted (position 4 (c % 4) + c / 4) so that the four values a lane feeds into the four MFMAs of a tap - channels
// h, 4 + h, 8 + h, 12 + h for k slot h = lane / 16 - are ONE ds_read_b128.  Same (tap, channel)-ascending fmaf chain per output as the VALU
// kernel (padding taps add fma(0, w, acc) = acc): bit-identical (test_small_conv_mfma_bit_identical_to_valu).
// ---------------------------------------------------------------------------------------------
template <int STRIDE>
__global__ __launch_bounds__(256) void small_conv3x3_mfma_kernel(SmallConvParams p, LmFastDiv div_tx, LmFastDiv div_ty, int ntiles) {
    constexpr int R = 15 * STRIDE + 3;                                 // patch edge (pixels)
    extern __shared__ __attribute__((aligned(16))) float patch[];       // [R][R][16 permuted channels]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, h = lane >> 4;
    // weights: B operand of (tap, channel group j) = w[(tap * 16 + 4 j + h) * 16 + l16]
    float wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[t][j] = p.w[(t * 16 + 4 * j + h) * 16 + l16];
    float sc = 1.f, sh = 0.f;
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
    if (has_sc && l16 < p.Cout) sc = p.scale[l16];
    if (has_sh && l16 < p.Cout) sh = p.shift[l16];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned trow = lm_fastdiv((unsigned)t, div_tx);
        const int b = (int)lm_fastdiv(trow, div_ty);
        const int oy0 = (int)(trow - (unsigned)b * div_ty.d) * 16, ox0 = (int)((unsigned)t - trow * div_tx.d) * 16;
        const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
        __syncthreads();                                               // the previous tile's reads are done
        // patch: R * R pixels x 4 channel quads; quad g of a pixel (channels 4 g .. 4 g + 3) goes to positions g, 4 + g, 8 + g, 12 + g
        constexpr int NCH = R * R * 4;
        for (int i0 = 0; i0 < NCH; i0 += 256 * 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + tid;
                const int pix = i >> 2, g = i & 3;
                const int r = pix / R, q = pix - r * R;
                const int iy = iy0 + r, ix = ix0 + q;
                // (unconditional load from a clamped address + select: a load under a branch is waited for at the join)
                const bool ok = (i < NCH) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const long src = ok ? (((long)b * p.H + iy) * p.W + ix) * p.ldx + 4 * g : 0;
                const f32x4 ld = *reinterpret_cast<const f32x4*>(p.x + src);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = ok ? (p.pre_relu ? fmaxf(ld[e], 0.f) : ld[e]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < NCH) {
                    float* d = patch + (i >> 2) * 16 + (i & 3);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[4 * e] = v[u][e];
                }
            }
        }
        __syncthreads();
        // wave w: output rows 4 w .. 4 w + 3 of the tile, one 16-pixel M block each
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ry = 4 * wave + mb;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int ky = tp / 3, kx = tp % 3;
                const f32x4 a = *reinterpret_cast<const f32x4*>(patch + ((ry * STRIDE + ky) * R + l16 * STRIDE + kx) * 16 + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wr[tp][j], acc, 0, 0, 0);
            }
            // accumulator register r: pixel 4 h + r of the row, output channel l16
            const int oy = oy0 + ry;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + 4 * h + r;
                if (oy < p.Ho && ox < p.Wo && l16 < p.Cout) {
                    float v = acc[r];
                    if (has_sc) v *= sc;
                    if (has_sh) v += sh;
                    if (p.act == LM_ACT_RELU) v = fmaxf(v, 0.f);
                    p.y[(((long)b * p.Ho + oy) * p.Wo + ox) * p.ldy + l16] = v;
                }
            }
        }
    }
}

template <int STRIDE>
int launch_small_mfma(const SmallConvParams& p, hipStream_t stream) {
    constexpr int R = 15 * STRIDE + 3;
    const size_t lds = (size_t)R * R * 16 * sizeof(float);
    if (int e = lm_ensure_dynamic_lds((const void*)small_conv3x3_mfma_kernel<STRIDE>, lds)) return e;
    const int tx = lm_cdiv(p.Wo, 16), ty = lm_cdiv(p.Ho, 16);
    const long ntiles = (long)tx * ty * p.B;
    LM_REQUIRE(ntiles < (1L << 31), "small_conv: too many tiles");
    const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
    hipLaunchKernelGGL(small_conv3x3_mfma_kernel<STRIDE>, dim3(grid), dim3(256), lds, stream, p, lm_fastdiv_make((unsigned)tx),
                       lm_fastdiv_make((unsigned)ty), (int)ntiles);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

}  // namespace

LM_API int lm_stem_conv7x7_bn_relu(void* stream, const float* x_chw, const float* w_k64, const float* scale,
                                   const float* shift, float* y_nhwc, int B, int H, int W) {
    LM_REQUIRE(x_chw && w_k64 && scale && shift && y_nhwc, "stem: null pointer");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return launch_stem<false>(stream, (const void*)x_chw, w_k64, scale, shift, y_nhwc, B, H, W, Ho, Wo);
}

// the same stem on a u8 HWC tile [B][H][W][3] (x = u8 / 255 applied on the fly): bit-identical to lm_tile_ingest_u8 + the f32 stem
LM_API int lm_stem_conv7x7_bn_relu_u8(void* stream, const unsigned char* x_hwc3, const float* w_k64, const float* scale,
                                      const float* shift, float* y_nhwc, int B, int H, int W) {
    LM_REQUIRE(x_hwc3 && w_k64 && scale && shift && y_nhwc, "stem_u8: null pointer");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return launch_stem<true>(stream, (const void*)x_hwc3, w_k64, scale, shift, y_nhwc, B, H, W, Ho, Wo);
}

LM_API int lm_maxpool3x3s2_nhwc(void* stream, const float* x, float* y, int B, int H, int W, int C) {
    LM_REQUIRE(x && y && C % 4 == 0, "maxpool: bad args (C=%d)", C);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total4 = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(lm_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, H, W, C, Ho, Wo, total4);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

LM_API int lm_conv2d_nhwc_small(void* stream, const float* x, int ldx, const float* w_tc16, const float* scale,
                                const float* shift, float* y, int ldy, int B, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad_h, int pad_w, int pre_relu, int act) {
    LM_REQUIRE(x && w_tc16 && y, "small_conv: null pointer");
    LM_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0 && Cout >= 1 && Cout <= 16, "small_conv: Cin=%d (mult of 4) Cout=%d (<=16)", Cin, Cout);
    SmallConvParams p;
    p.x = x; p.w = w_tc16; p.scale = scale; p.shift = shift; p.y = y;
    p.ldx = ldx; p.ldy = ldy; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.stride = stride; p.pad_h = pad_h; p.pad_w = pad_w; p.pre_relu = pre_relu; p.act = act;
    p.Ho = (H + 2 * pad_h - KH) / stride + 1;
    p.Wo = (W + 2 * pad_w - KW) / stride + 1;
    p.M = (long)B * p.Ho * p.Wo;
    if (Cin == 16 && KH == 3 && KW == 3 && pad_h == 1 && pad_w == 1 && (stride == 1 || stride == 2) && act != LM_ACT_GELU) {
        // LM_SMALL_CONV_VALU=1: the VALU kernel for these shapes too
        static const bool valu = [] { const char* e = getenv("LM_SMALL_CONV_VALU"); return e && atoi(e) != 0; }();
        if (!valu) return stride == 1 ? launch_small_mfma<1>(p, (hipStream_t)stream) : launch_small_mfma<2>(p, (hipStream_t)stream);
    }
    hipLaunchKernelGGL(small_conv_kernel, dim3(lm_cdiv(p.M, 256)), dim3(256), 0, (hipStream_t)stream, p);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

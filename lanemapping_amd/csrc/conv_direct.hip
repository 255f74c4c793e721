// Direct (VALU) convolutions for the layers that are too thin for the matrix cores, plus max-pool.
//
//  lm_stem_conv7x7_bn_relu : FPNWrapper.conv1 + bn1 + relu      (postprojector.py:458-460,566)  3 -> 64, 7x7 s2 p3
//  lm_maxpool3x3s2_nhwc    : FPNWrapper.maxpool                  (postprojector.py:461,567)
//  lm_conv2d_nhwc_small    : feature_layer / output_layer_* 1x1  (postprojector.py:509-511,628-651),
//                            head_common_layers, orient, bi_seg_proposal
//                            (heads/polyline_fpn_vit_vertex_2.py:183-189,232-237,249)
#include "common.h"

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

}  // namespace

LM_API int lm_stem_conv7x7_bn_relu(void* stream, const float* x_chw, const float* w_k64, const float* scale,
                                   const float* shift, float* y_nhwc, int B, int H, int W) {
    LM_REQUIRE(x_chw && w_k64 && scale && shift && y_nhwc, "stem: null pointer");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    dim3 grid(lm_cdiv(Wo, ST), lm_cdiv(Ho, ST), B);
    hipLaunchKernelGGL(stem_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const void*)x_chw, w_k64, scale, shift, y_nhwc, H, W, Ho, Wo);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// the same stem on a u8 HWC tile [B][H][W][3] (x = u8 / 255 applied on the fly): bit-identical to lm_tile_ingest_u8 + the f32 stem
LM_API int lm_stem_conv7x7_bn_relu_u8(void* stream, const unsigned char* x_hwc3, const float* w_k64, const float* scale,
                                      const float* shift, float* y_nhwc, int B, int H, int W) {
    LM_REQUIRE(x_hwc3 && w_k64 && scale && shift && y_nhwc, "stem_u8: null pointer");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    dim3 grid(lm_cdiv(Wo, ST), lm_cdiv(Ho, ST), B);
    hipLaunchKernelGGL(stem_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const void*)x_hwc3, w_k64, scale, shift, y_nhwc, H, W, Ho, Wo);
    LM_LAUNCH_CHECK();
    return LM_OK;
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
    hipLaunchKernelGGL(small_conv_kernel, dim3(lm_cdiv(p.M, 256)), dim3(256), 0, (hipStream_t)stream, p);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

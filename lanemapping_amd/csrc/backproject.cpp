// BEV polylines -> LAS frame.  Host code, double precision, same operation order as the reference
// so that the result is bit-identical to numpy's (built with FP contraction off):
//   transform_coordinate_from_img_2_pc      baseline/utils/coor_img2pc.py:127-183
//   modify_empty_pixel_elevation (roi form) baseline/utils/coor_img2pc.py:94-122
//   LeastSuqare                             baseline/utils/coor_img2pc.py:59-73
//   rotateByQuanternion3D / multiplyQuanternion  baseline/utils/coor_img2pc.py:22-53
// The elevation fill mutates the tile as it goes (a filled pixel counts as non-empty for later vertices), so the vertices
// are processed sequentially in the reference's order; with <= 72 x 144 vertices per tile this is microseconds of host
// work overlapped with the GPU like the rest of the polyline tail.
#include "common.h"

#include <cmath>

namespace {

void qmul(const double a[4], const double b[4], double o[4]) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

}  // namespace

// bev_hwc: [H][W][C] u8 tile, MODIFIED in place (elevation channel 1 of empty vertex pixels, reference behaviour).
// img_seqs: [L][Vmax][2] (row, col) doubles, seq_lens [L].  params13: img_reso[2], bev_img_offset[2], ele_reso,
// local_min_ele, las_rotation_trans_quan[7] = (tx, ty, tz, w, x, y, z).  las_read_offset[3].  out: [L][Vmax][3].
LM_API int lm_polyline_backproject(unsigned char* bev_hwc, int H, int W, int C, const double* img_seqs, const int* seq_lens, int L,
                                   int Vmax, const double* params13, const double* las_read_offset, double* out) {
    LM_REQUIRE(bev_hwc && img_seqs && seq_lens && params13 && las_read_offset && out, "polyline_backproject: null pointer");
    LM_REQUIRE(H > 0 && W > 0 && C >= 2 && L >= 0 && Vmax >= 0, "polyline_backproject: bad sizes");
    const double reso0 = params13[0], reso1 = params13[1], off0 = params13[2], off1 = params13[3];
    const double ele_reso = params13[4], min_ele = params13[5];
    const double* trans = params13 + 6;
    const double quan[4] = {params13[9], params13[10], params13[11], params13[12]};
    auto px = [&](int h, int w, int c) -> unsigned char& { return bev_hwc[((long)h * W + w) * C + c]; };
    for (int l = 0; l < L; ++l) {
        LM_REQUIRE(seq_lens[l] >= 0 && seq_lens[l] <= Vmax, "polyline_backproject: seq_lens[%d]=%d out of range", l, seq_lens[l]);
        for (int v = 0; v < Vmax; ++v) {
            const double r = img_seqs[((long)l * Vmax + v) * 2], c = img_seqs[((long)l * Vmax + v) * 2 + 1];
            LM_REQUIRE(r >= 0 && r < H && c >= 0 && c < W, "polyline_backproject: vertex (%g, %g) outside the %dx%d tile", r, c, H, W);
        }
    }
    // 1) elevation of empty vertex pixels := mean G of the smallest non-empty window around them (growing square, the
    //    reference's half-open [p - step, p + step) window)
    for (int l = 0; l < L; ++l)
        for (int v = 0; v < seq_lens[l]; ++v) {
            const int ph = (int)img_seqs[((long)l * Vmax + v) * 2], pw = (int)img_seqs[((long)l * Vmax + v) * 2 + 1];
            unsigned long s = 0;
            for (int c = 0; c < C; ++c) s += px(ph, pw, c);
            if ((ph == 0 && pw == 0) || s > 1) continue;
            for (int step = 1;; ++step) {
                LM_REQUIRE(step <= H + W, "polyline_backproject: the tile is empty around vertex (%d, %d)", ph, pw);
                const int h0 = ph - step > 0 ? ph - step : 0, h1 = ph + step < H ? ph + step : H;
                const int w0 = pw - step > 0 ? pw - step : 0, w1 = pw + step < W ? pw + step : W;
                unsigned long total = 0, gsum = 0, valid = 0;
                for (int h = h0; h < h1; ++h)
                    for (int w = w0; w < w1; ++w) {
                        unsigned long ps = 0;
                        for (int c = 0; c < C; ++c) ps += px(h, w, c);
                        total += ps;
                        gsum += px(h, w, 1);
                        valid += ps > 0;
                    }
                if (total > 0) {
                    px(ph, pw, 1) = (unsigned char)((double)gsum / (double)valid);   // numpy float -> uint8 assignment truncates
                    break;
                }
            }
        }
    // 2) affine + elevation lookup (every slot, padding included, like the reference)
    for (long i = 0; i < (long)L * Vmax; ++i) {
        const double r = img_seqs[i * 2], c = img_seqs[i * 2 + 1];
        out[i * 3 + 0] = r * reso0 + off0;
        out[i * 3 + 1] = c * reso1 + off1;
        out[i * 3 + 2] = (double)px((int)r, (int)c, 1) * ele_reso + min_ele;
    }
    // 3) per-line least-squares elevation over the vertex index (python `sum`: sequential left-to-right additions)
    for (int l = 0; l < L; ++l) {
        const int n = seq_lens[l];
        if (n == 0) continue;                                  // (the reference divides by zero here)
        double* z = out + (long)l * Vmax * 3 + 2;
        double sxy = 0.0, sy = 0.0;
        long sx = 0, sxx = 0;
        for (int i = 0; i < n; ++i) {
            sxy = sxy + (double)i * z[i * 3];
            sy = sy + z[i * 3];
            sx += i;
            sxx += (long)i * i;
        }
        const double p = (double)n * sxy - (double)sx * sy;
        const long q = (long)n * sxx - sx * sx;
        const double wgt = (std::fabs((double)q) < 1e-6) ? 0.0 : p / (double)q;
        double sb = 0.0;
        for (int i = 0; i < n; ++i) sb = sb + (z[i * 3] - wgt * (double)i);
        const double b = sb / (double)n;
        for (int i = 0; i < n; ++i) z[i * 3] = wgt * (double)i + b;
    }
    // 4) q v q* / |q|, + translation, + las_read_offset
    const double qn = std::sqrt(((quan[0] * quan[0] + quan[1] * quan[1]) + quan[2] * quan[2]) + quan[3] * quan[3]);
    LM_REQUIRE(qn > 1e-6, "polyline_backproject: zero quaternion");
    const double qinv[4] = {quan[0] / qn, (quan[1] * -1.0) / qn, (quan[2] * -1.0) / qn, (quan[3] * -1.0) / qn};
    for (long i = 0; i < (long)L * Vmax; ++i) {
        const double qv[4] = {0.0, out[i * 3], out[i * 3 + 1], out[i * 3 + 2]};
        double t[4], o[4];
        qmul(quan, qv, t);
        qmul(t, qinv, o);
        for (int a = 0; a < 3; ++a) out[i * 3 + a] = (o[1 + a] + trans[a]) + las_read_offset[a];
    }
    return LM_OK;
}

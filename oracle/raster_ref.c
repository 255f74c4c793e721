/* Oracle (TEST INFRASTRUCTURE ONLY): scalar C restatement of the build-defined LAS->BEV rule.
 *
 * PARITY UNPINNED: the reference ships no rasteriser.  The rule is anchored on
 *   - read_las's record/intensity normalisation   baseline/datasets/laserlane_proposals.py:618-636
 *   - the inverse transform image->point cloud    baseline/utils/coor_img2pc.py:127-183 (and :22-53)
 *   - the tile contract u8/255, empty <=> R+G+B<1  laserlane_proposals.py:85-98, coor_img2pc.py:78,106
 * Same fp32 expression order as lanemapping_amd/csrc/raster.hip (both built with FP contraction off),
 * so the u8 image must match bit for bit.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    float quat[4], trans[3], bev_img_offset[2], img_reso[2], local_min_ele, ele_reso, inten_lo, inten_hi;
} RasterParams;

/* acc: [H*W] u32 keys (intensity<<8 | elevation), out_u8: [H][W][3] */
void raster_ref(const float* pts, long n, const RasterParams* P, uint32_t* acc, uint8_t* out_u8, int H, int W) {
    memset(acc, 0, (size_t)H * W * sizeof(uint32_t));
    /* inverse of the reference's rotation r(v) = q v q* / |q| = |q| R(q^) v   =>   M = R(q^)^T / |q|  (double -> float) */
    const double nq = sqrt((double)P->quat[0] * P->quat[0] + (double)P->quat[1] * P->quat[1] + (double)P->quat[2] * P->quat[2] +
                           (double)P->quat[3] * P->quat[3]);
    const double w = P->quat[0] / nq, x = P->quat[1] / nq, y = P->quat[2] / nq, z = P->quat[3] / nq;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    float m[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i * 3 + j] = (float)(R[j * 3 + i] / nq);
    const float irow = 1.0f / P->img_reso[0], icol = 1.0f / P->img_reso[1], iele = 1.0f / P->ele_reso;
    const float iscale = 255.0f / P->inten_hi;
    for (long i = 0; i < n; ++i) {
        const float* p = pts + 4 * i;
        const float dx = p[0] - P->trans[0], dy = p[1] - P->trans[1], dz = p[2] - P->trans[2];
        const float vx = (m[0] * dx + m[1] * dy) + m[2] * dz;
        const float vy = (m[3] * dx + m[4] * dy) + m[5] * dz;
        const float vz = (m[6] * dx + m[7] * dy) + m[8] * dz;
        const int row = (int)floorf((vx - P->bev_img_offset[0]) * irow + 0.5f);
        const int col = (int)floorf((vy - P->bev_img_offset[1]) * icol + 0.5f);
        if (row < 0 || row >= H || col < 0 || col >= W) continue;
        const float it = fminf(fmaxf(p[3], P->inten_lo), P->inten_hi) - P->inten_lo;
        int I = (int)floorf(it * iscale + 0.5f);
        I = I < 1 ? 1 : (I > 255 ? 255 : I);
        int G = (int)floorf((vz - P->local_min_ele) * iele + 0.5f);
        G = G < 0 ? 0 : (G > 255 ? 255 : G);
        const uint32_t key = (uint32_t)((I << 8) | G);
        uint32_t* a = acc + (long)row * W + col;
        if (key > *a) *a = key;
    }
    for (long i = 0; i < (long)H * W; ++i) {
        out_u8[3 * i + 0] = (uint8_t)((acc[i] >> 8) & 255u);
        out_u8[3 * i + 1] = (uint8_t)(acc[i] & 255u);
        out_u8[3 * i + 2] = (uint8_t)((acc[i] >> 8) & 255u);
    }
}

/* Reference-side inverse (coor_img2pc.py:136-177 without the elevation smoothing): pixel -> point. */
void pixel_to_point_ref(const RasterParams* P, double row, double col, double g, double out[3]) {
    double v[3] = {row * P->img_reso[0] + P->bev_img_offset[0], col * P->img_reso[1] + P->bev_img_offset[1],
                   g * P->ele_reso + P->local_min_ele};
    double q[4] = {P->quat[0], P->quat[1], P->quat[2], P->quat[3]};
    double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double qi[4] = {q[0] / nrm, -q[1] / nrm, -q[2] / nrm, -q[3] / nrm};
    double a[4] = {0, v[0], v[1], v[2]}, t[4], r[4];
    /* multiplyQuanternion(quan, v) then (.., quan_inv)  coor_img2pc.py:22-53 */
    t[0] = q[0] * a[0] - q[1] * a[1] - q[2] * a[2] - q[3] * a[3];
    t[1] = q[0] * a[1] + q[1] * a[0] + q[2] * a[3] - q[3] * a[2];
    t[2] = q[0] * a[2] - q[1] * a[3] + q[2] * a[0] + q[3] * a[1];
    t[3] = q[0] * a[3] + q[1] * a[2] - q[2] * a[1] + q[3] * a[0];
    r[1] = t[0] * qi[1] + t[1] * qi[0] + t[2] * qi[3] - t[3] * qi[2];
    r[2] = t[0] * qi[2] - t[1] * qi[3] + t[2] * qi[0] + t[3] * qi[1];
    r[3] = t[0] * qi[3] + t[1] * qi[2] - t[2] * qi[1] + t[3] * qi[0];
    out[0] = r[1] + P->trans[0];
    out[1] = r[2] + P->trans[1];
    out[2] = r[3] + P->trans[2];
}

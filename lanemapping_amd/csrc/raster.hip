// LAS -> BEV rasteriser and tile ingest (row a1).
//
// The reference has NO rasteriser (it points at the external MIXIAOXIN/Las2BEV repo), so the
// pixel rule is build-defined and PARITY IS UNPINNED.  What the reference does pin is
//   * the point record and intensity normalisation of `read_las`
//     (baseline/datasets/laserlane_proposals.py:618-636): [x,y,z,intensity], intensity clipped to
//     [800, 33000] then (i-800)/33000;
//   * the INVERSE geometry, image -> point cloud (baseline/utils/coor_img2pc.py:127-183):
//     X = row*img_reso[0] + bev_img_offset[0], Y = col*img_reso[1] + bev_img_offset[1],
//     Z = G*ele_reso + local_min_ele, then rotate by quaternion [w,x,y,z] (q v q* / |q|), + translation;
//   * the tile contract of `load_img` (laserlane_proposals.py:85-98): u8 HWC -> f32 CHW / 255, and
//     "pixel empty <=> R+G+B < 1" (coor_img2pc.py:78,106).
// Rule implemented here (scatter-max, order independent => deterministic):
//   v = M (p - t) with M = R(q)^T / |q| (the exact inverse of the reference's rotation), evaluated in fp32 as
//   (M0*dx + M1*dy) + M2*dz;  row = floor((v.x-off0)/reso0 + .5), col likewise;  I = round(255*(clip(i)-lo)/hi)
//   in 1..255;  G = clamp(round((v.z-min_ele)/ele_reso), 0, 255);  a pixel keeps the max over its points of
//   key = I<<8 | G, i.e. R = B = brightest return, G = its elevation.  Untouched pixels stay 0 (empty).
//
// Kernel design (HBM-bound; points arrive in acquisition order, i.e. spatially unsorted, so a workgroup cannot
// own a pixel region directly; one device-scope atomicMax per point (v0) ran at 0.45 TB/s):
//   pass 1  partition: every workgroup streams 16,384 points with coalesced 16-byte non-temporal loads (8 in flight per
//           lane), turns each into a 4-byte record {pixel-in-band:16 | I:8 | G:8}, counting-sorts the records by band (16 or 12
//           rows, band_rows_for) in LDS (rank = LDS atomic add on 8x replicated counters) and writes each band's run, padded to 16
//           bytes, with aligned dwordx4 stores into its own static slot [tile][band][workgroup][16384] plus one contiguous row of
//           counts [tile][workgroup][band].  No global atomics, no memset, no inter-workgroup communication.
//   pass 2  one 1024-thread workgroup per (tile, band): the band's rows x W u32 image lives in LDS; the runs are read 8 at a time
//           per 16-lane group (all counts, then all first quads: two memory round trips) and applied with LDS atomic max, then the
//           band is written once as u8 HWC (4 pixels = 3 packed dwords per thread) and / or fp32 CHW.
// HBM traffic per tile (PMC, N = 4,194,304, u8 output) = 16 N (points) + 4.2 N (records out) + 5.0 N (records + counts in) + H W 3
// = 110 MB = 1.32x the algorithmic bytes (16 N + 3 H W 4): at the 6.3 TB/s of a pure copy that alone is 17.5 us = 0.59 of 8 TB/s,
// the ceiling of any two-pass scheme; measured 23-25 us = 0.42-0.45 (DESIGN.md §3.2 has the counter-backed breakdown).
#include "common.h"

#include <cmath>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct LmRasterParams {      // mirrors the reference's per-tile parameter file (utils/io_utils.py:125-150)
    float quat[4];           // las_rotation_trans_quan[3:7] = [w,x,y,z]
    float trans[3];          // las_rotation_trans_quan[0:3]
    float bev_img_offset[2];
    float img_reso[2];
    float local_min_ele;
    float ele_reso;
    float inten_lo, inten_hi;   // 800, 33000
};

namespace {

// Pass-1 workgroup size (x 32 points per thread = the chunk) and pass-2 prefetch depth: 512 threads / 16,384-point chunks (twice as
// long record runs, half the count rows; 71 KB of LDS = two workgroups per CU) with 3 quads per lane measured 24.3 us per tile against
// 25.4-25.7 for 256 / 2 on the same box; 1024 threads (one workgroup per CU: its load, sort and write-out phases no longer overlap
// with another workgroup's) 41 us.
#ifndef LM_RASTER_NT
#define LM_RASTER_NT 512
#endif
#ifndef LM_RASTER_BQ
#define LM_RASTER_BQ 3
#endif
constexpr int NT = LM_RASTER_NT;       // threads per pass-1 workgroup
constexpr int PER_THREAD = 32;
constexpr int CHUNK = NT * PER_THREAD; // points per pass-1 workgroup = record capacity of one (tile, band, workgroup) slot
constexpr int PART_CAPQ = (NT * 32 + 96 * 3 + 3) / 4;       // quads of the sorted buffer: every band's run is padded to 16 bytes
constexpr size_t PART_LDS = (size_t)PART_CAPQ * 16 + ((PART_CAPQ + 15) / 16) * 16;
constexpr int BQ = LM_RASTER_BQ;       // quads per lane of a 16-lane group fetched with a run's first round trip (pass 2)
constexpr int MAX_BANDS = 96;             // (12-row bands of a 1152-row tile)
constexpr int MAX_TILES = 16;          // tiles per launch (kernel-argument block)
constexpr int REP = 8;                 // replication of the LDS rank counters (fewer same-address collisions)

struct TileXf {                        // derived per-tile constants (host, double -> float)
    float m[9], t[3], off[2], irow, icol, min_ele, iele, lo, hi, iscale;
    long start, count;                 // point range of the tile in the concatenated buffer
};
struct BatchArgs {
    TileXf tile[MAX_TILES];
};

__device__ __forceinline__ bool point_record(const f32x4 p, const TileXf& X, int H, int W, int band_rows, int& band, unsigned& rec) {
    const float dx = p[0] - X.t[0], dy = p[1] - X.t[1], dz = p[2] - X.t[2];
    const float vx = (X.m[0] * dx + X.m[1] * dy) + X.m[2] * dz;
    const float vy = (X.m[3] * dx + X.m[4] * dy) + X.m[5] * dz;
    const float vz = (X.m[6] * dx + X.m[7] * dy) + X.m[8] * dz;
    const int row = (int)floorf((vx - X.off[0]) * X.irow + 0.5f);
    const int col = (int)floorf((vy - X.off[1]) * X.icol + 0.5f);
    if ((unsigned)row >= (unsigned)H || (unsigned)col >= (unsigned)W) return false;
    const float it = fminf(fmaxf(p[3], X.lo), X.hi) - X.lo;
    int I = (int)floorf(it * X.iscale + 0.5f);
    I = I < 1 ? 1 : (I > 255 ? 255 : I);
    int G = (int)floorf((vz - X.min_ele) * X.iele + 0.5f);
    G = G < 0 ? 0 : (G > 255 ? 255 : G);
    band = row / band_rows;
    rec = ((unsigned)((row - band * band_rows) * W + col) << 16) | (unsigned)((I << 8) | G);
    return true;
}

// records: [tile][band][blk][CHUNK] u32 (zero padded to 16 B per run), counts: [tile][nblk_max][band]
// grid: (nblk_max, tiles)
__global__ __launch_bounds__(NT) void raster_partition_kernel(const f32x4* __restrict__ pts, BatchArgs A, unsigned* __restrict__ counts,
                                                               unsigned* __restrict__ records, int nblk_max, int H, int W, int nbands,
                                                               int band_rows) {
    constexpr int CAPQ = PART_CAPQ;
    __shared__ unsigned hist[MAX_BANDS * REP];                 // [band][replica] count, then start offset in `sorted`
    __shared__ unsigned qstart[MAX_BANDS + 1];                 // first 16-byte quad of each band's run
    extern __shared__ __attribute__((aligned(16))) unsigned part_lds[];      // sorted[CAPQ * 4] | qband[CAPQ] (dynamic: > 64 KB from 512 threads on)
    unsigned* const sorted = part_lds;
    unsigned char* const qband = reinterpret_cast<unsigned char*>(part_lds + CAPQ * 4);
    const int tile = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const TileXf& X = A.tile[tile];
    const long first = (long)blk * CHUNK;
    if (first >= X.count) return;                              // pass 2 never looks at slots beyond the tile's last chunk
    for (int i = tid; i < MAX_BANDS * REP; i += NT) hist[i] = 0;
    for (int i = tid; i < CAPQ * 4; i += NT) sorted[i] = 0;   // padding records must be 0
    __syncthreads();
    unsigned rec[PER_THREAD];
    unsigned meta[PER_THREAD];                                 // slot << 16 | rank, 0xFFFFFFFF = dropped
    const f32x4* base = pts + X.start + first;
    const long left = X.count - first;
    const int rep = tid & (REP - 1);
    constexpr int LB = 8;                                      // loads kept in flight per thread
    // (explicitly issuing batch k+1 before batch k is consumed - two register sets - measured slower: 290 -> 303 us per 16 tiles,
    // 134 VGPRs; the fully unrolled loop below already lets the compiler hoist the next batch's loads)
#pragma unroll
    for (int j0 = 0; j0 < PER_THREAD; j0 += LB) {
        f32x4 p[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const long i = (long)(j0 + j) * NT + tid;
            // unconditional (index clamped into the chunk): a predicated load made the compiler wait for each load before
            // issuing the next one, i.e. 1 load in flight per lane instead of 8; the tail lanes are masked below
            p[j] = __builtin_nontemporal_load(base + (i < left ? i : left - 1));
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const long i = (long)(j0 + j) * NT + tid;
            meta[j0 + j] = 0xFFFFFFFFu;
            int band;
            if (i < left && point_record(p[j], X, H, W, band_rows, band, rec[j0 + j])) {
                const unsigned slot = (unsigned)band * REP + rep;
                meta[j0 + j] = (slot << 16) | atomicAdd(&hist[slot], 1u);
            }
        }
    }
    __syncthreads();
    if (tid < 64) {   // wave 0: exclusive scan over bands (lane owns bands tid and tid+64); every run starts on a quad
        unsigned bc[2], bq[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = tid + 64 * k;
            unsigned c = 0;
            if (b < nbands)
                for (int r = 0; r < REP; ++r) c += hist[b * REP + r];
            bc[k] = c;
            bq[k] = (c + 3) / 4;
        }
        unsigned incl0 = bq[0], incl1 = bq[1];
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v0 = __shfl_up(incl0, o), v1 = __shfl_up(incl1, o);
            if (tid >= o) {
                incl0 += v0;
                incl1 += v1;
            }
        }
        const unsigned tot0 = __shfl(incl0, 63);
        const unsigned start[2] = {incl0 - bq[0], tot0 + incl1 - bq[1]};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = tid + 64 * k;
            if (b < nbands) {
                qstart[b] = start[k];
                unsigned run = start[k] * 4;
                for (int r = 0; r < REP; ++r) {
                    const unsigned c = hist[b * REP + r];
                    hist[b * REP + r] = run;
                    run += c;
                }
                counts[((long)tile * nblk_max + blk) * nbands + b] = bc[k];      // [tile][chunk][band]: one contiguous row per workgroup
            }
        }
        if (tid == 63) qstart[nbands] = tot0 + incl1;          // total quads (bands >= nbands contribute 0)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j)
        if (meta[j] != 0xFFFFFFFFu) sorted[hist[meta[j] >> 16] + (meta[j] & 0xFFFFu)] = rec[j];
    for (int b = tid; b < nbands; b += NT)
        for (unsigned q = qstart[b]; q < qstart[b + 1]; ++q) qband[q] = (unsigned char)b;
    __syncthreads();
    const unsigned totq = qstart[nbands];
    const u32x4* s4 = reinterpret_cast<const u32x4*>(sorted);
    for (unsigned q = tid; q < totq; q += NT) {
        const unsigned b = qband[q];
        u32x4* dst = reinterpret_cast<u32x4*>(records + (((long)tile * nbands + b) * nblk_max + blk) * CHUNK) + (q - qstart[b]);
        *dst = s4[q];
    }
}

// grid: (bands, tiles); dynamic LDS = band_rows * W * 4 bytes; 1024 threads = 64 groups of 16 lanes, one run per group
constexpr int BT = 1024;

struct BandArgs {
    int nblk[MAX_TILES];                                       // valid chunks per tile
};

__device__ __forceinline__ void apply4(unsigned* img, const u32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (v[e]) atomicMax(&img[v[e] >> 16], v[e] & 0xFFFFu);   // a real record has key >= 256; padding is 0
}

__global__ __launch_bounds__(BT) void raster_band_kernel(const unsigned* __restrict__ counts, const unsigned* __restrict__ records,
                                                         BandArgs A, int nblk_max, float* __restrict__ out_chw,
                                                         unsigned char* __restrict__ out_u8, int H, int W, int nbands, int band_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned img[];
    const int band = blockIdx.x, tile = blockIdx.y;
    const int npix = band_rows * W;
    for (int i = threadIdx.x; i < npix; i += BT) img[i] = 0;
    __syncthreads();
    const int nblk = A.nblk[tile];
    const unsigned* cnt = counts + (long)tile * nblk_max * nbands + band;          // cnt[b * nbands] = records of chunk b in this band
    const unsigned* rbase = records + ((long)tile * nbands + band) * (long)nblk_max * CHUNK;
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    // A 16-lane group owns the runs b = grp + 64 k.  Eight runs per step: first ALL their counts (one memory round trip), then ALL
    // their first 32 quads (16 loads in flight per lane, a second round trip) - the loop used to pay two dependent round trips per
    // pair of runs, i.e. eight per workgroup at 512 chunks per tile.
    constexpr int RUNS = 8;
    for (int b0 = grp; b0 < nblk; b0 += 64 * RUNS) {
        unsigned nq[RUNS];
#pragma unroll
        for (int u = 0; u < RUNS; ++u) {
            const int b = b0 + 64 * u;
            nq[u] = b < nblk ? (cnt[(long)b * nbands] + 3) / 4 : 0;
        }
        u32x4 v[RUNS][BQ];
#pragma unroll
        for (int u = 0; u < RUNS; ++u) {
            const u32x4* r4 = reinterpret_cast<const u32x4*>(rbase + (long)(b0 + 64 * u) * CHUNK);
#pragma unroll
            for (int k = 0; k < BQ; ++k)
                v[u][k] = ((unsigned)(gl + 16 * k) < nq[u]) ? __builtin_nontemporal_load(r4 + gl + 16 * k) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < RUNS; ++u) {
#pragma unroll
            for (int k = 0; k < BQ; ++k) apply4(img, v[u][k]);
            if (nq[u] > 16 * BQ) {                             // long run (spatially skewed chunk): finish it here
                const u32x4* r4 = reinterpret_cast<const u32x4*>(rbase + (long)(b0 + 64 * u) * CHUNK);
                for (unsigned q = 16 * BQ + gl; q < nq[u]; q += 16) apply4(img, r4[q]);
            }
        }
    }
    __syncthreads();
    const long HW = (long)H * W;
    const long pix0 = (long)band * npix;
    if (out_chw) {
        float* o = out_chw + (long)tile * 3 * HW + pix0;
        for (int i = threadIdx.x; i < npix; i += BT) {
            const unsigned k = img[i];
            const float fi = (float)(k >> 8) / 255.0f, fg = (float)(k & 255u) / 255.0f;   // == u8 / 255 (to_tensor)
            o[i] = fi;
            o[HW + i] = fg;
            o[2 * HW + i] = fi;
        }
    }
    if (out_u8) {
        // 4 pixels = 12 bytes = three dwords per thread (byte stores are an order of magnitude slower per byte); npix % 4 == 0
        unsigned* o = reinterpret_cast<unsigned*>(out_u8 + ((long)tile * HW + pix0) * 3);
        for (int i = threadIdx.x * 4; i < npix; i += BT * 4) {
            const unsigned k0 = img[i], k1 = img[i + 1], k2 = img[i + 2], k3 = img[i + 3];
            const unsigned i0 = k0 >> 8, g0 = k0 & 255u, i1 = k1 >> 8, g1 = k1 & 255u, i2 = k2 >> 8, g2 = k2 & 255u, i3 = k3 >> 8, g3 = k3 & 255u;
            unsigned* d = o + (i / 4) * 3;                  // bytes I0 G0 I0 I1 | G1 I1 I2 G2 | I2 I3 G3 I3 (little endian)
            d[0] = i0 | (g0 << 8) | (i0 << 16) | (i1 << 24);
            d[1] = g1 | (i1 << 8) | (i2 << 16) | (g2 << 24);
            d[2] = i2 | (i3 << 8) | (g3 << 16) | (i3 << 24);
        }
    }
}

// tile ingest: u8 HWC (PNG decode) -> f32 CHW / 255, first 3 channels (load_img contract)
__global__ __launch_bounds__(256) void ingest_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long HW,
                                                     int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over B*HW
    if (i >= total) return;
    const long b = i / HW, r = i - b * HW;
    const unsigned char* s = src + i * C;
    float* d = dst + b * 3 * HW + r;
    d[0] = (float)s[0] / 255.0f;
    d[HW] = (float)s[1] / 255.0f;
    d[2 * HW] = (float)s[2] / 255.0f;
}

void derive(const LmRasterParams& P, long start, long count, TileXf& X) {
    // inverse of the reference's rotation r(v) = q v q* / |q| = |q| R(q^) v   =>   M = R(q^)^T / |q|
    const double n = std::sqrt((double)P.quat[0] * P.quat[0] + (double)P.quat[1] * P.quat[1] + (double)P.quat[2] * P.quat[2] +
                               (double)P.quat[3] * P.quat[3]);
    const double w = P.quat[0] / n, x = P.quat[1] / n, y = P.quat[2] / n, z = P.quat[3] / n;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) X.m[i * 3 + j] = (float)(R[j * 3 + i] / n);
    for (int i = 0; i < 3; ++i) X.t[i] = P.trans[i];
    X.off[0] = P.bev_img_offset[0];
    X.off[1] = P.bev_img_offset[1];
    X.irow = 1.0f / P.img_reso[0];
    X.icol = 1.0f / P.img_reso[1];
    X.min_ele = P.local_min_ele;
    X.iele = 1.0f / P.ele_reso;
    X.lo = P.inten_lo;
    X.hi = P.inten_hi;
    X.iscale = 255.0f / P.inten_hi;
    X.start = start;
    X.count = count;
}

}  // namespace

static long nblk_of(long n) { return (n + CHUNK - 1) / CHUNK; }

// Rows per band: the band kernel runs (bands x tiles) workgroups on 256 CUs x floor(160 KB / band image) slots; with 16-row bands a
// 16-tile batch of 1152-row tiles is 1152 workgroups on 512 slots = 2.25 rounds (the third one a quarter full), with 12-row bands
// 1536 on 512 = exactly 3 rounds of 3/4 the work each.  Picks the candidate with the least rounds x rows.  LM_RASTER_BAND_ROWS overrides.
static int band_rows_for(int B, int H, int W) {
    static const int forced = [] { const char* e = getenv("LM_RASTER_BAND_ROWS"); return e ? atoi(e) : 0; }();
    if (forced > 0 && H % forced == 0 && H / forced <= MAX_BANDS && (long)forced * W <= 65536 && forced % 4 == 0) return forced;
    int best = 0;
    long best_cost = 0;
    for (int r : {16, 12}) {
        if (H % r != 0 || H / r > MAX_BANDS || (long)r * W > 65536) continue;
        const long lds = (long)r * W * 4, per_cu = lds > 0 ? (160 * 1024) / lds : 1;
        const long slots = 256 * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));     // 1024-thread workgroups: at most 2 per CU
        const long wgs = (long)(H / r) * (B < MAX_TILES ? B : MAX_TILES);
        const long cost = ((wgs + slots - 1) / slots) * r;
        if (!best || cost < best_cost) {
            best = r;
            best_cost = cost;
        }
    }
    return best;
}

LM_API long lm_bev_raster_workspace_bytes(int B, long max_points_per_tile, int H, int W) {
    const int br = band_rows_for(B, H, W);
    if (!br) return 0;
    const long nbands = H / br, nblk = nblk_of(max_points_per_tile) > 0 ? nblk_of(max_points_per_tile) : 1;
    const long head = (((long)B * nbands * nblk * (long)sizeof(unsigned) + 255) / 256) * 256;
    return head + (long)B * nbands * nblk * CHUNK * (long)sizeof(unsigned);
}

// points: device [sum N, 4] f32; tile_offsets: HOST [B+1] (point index of each tile's first record); params: HOST [B]
LM_API int lm_bev_raster_batch(void* stream, const float* points_xyzi, const long* tile_offsets, const LmRasterParams* params,
                               int B, void* workspace, long workspace_bytes, float* out_chw, unsigned char* out_hwc_u8,
                               int H, int W) {
    LM_REQUIRE(tile_offsets && params && workspace && (out_chw || out_hwc_u8) && B >= 1, "bev_raster: null pointer");
    const int band_rows = band_rows_for(B, H, W);
    LM_REQUIRE(band_rows > 0 && W > 0, "bev_raster: H=%d must be a multiple of 16 or 12 (at most %d bands) and rows*W <= 65536", H, MAX_BANDS);
    const int nbands = H / band_rows;
    long nmax = 0;
    for (int b = 0; b < B; ++b) {
        const long n = tile_offsets[b + 1] - tile_offsets[b];
        LM_REQUIRE(n >= 0, "bev_raster: tile offsets must be non-decreasing");
        LM_REQUIRE(params[b].img_reso[0] > 0 && params[b].img_reso[1] > 0 && params[b].ele_reso > 0, "bev_raster: bad resolution");
        nmax = n > nmax ? n : nmax;
    }
    LM_REQUIRE(points_xyzi || nmax == 0, "bev_raster: null points");
    LM_REQUIRE(lm_bev_raster_workspace_bytes(B, nmax, H, W) <= workspace_bytes, "bev_raster: workspace too small (%ld B needed)",
               lm_bev_raster_workspace_bytes(B, nmax, H, W));
    const int nblk_max = (int)(nblk_of(nmax) > 0 ? nblk_of(nmax) : 1);
    hipStream_t s = (hipStream_t)stream;
    unsigned* counts = (unsigned*)workspace;
    unsigned* records = (unsigned*)((char*)workspace + (((long)B * nbands * nblk_max * sizeof(unsigned) + 255) / 256) * 256);
    const size_t lds = (size_t)band_rows * W * sizeof(unsigned);
    if (int e = lm_ensure_dynamic_lds((const void*)raster_band_kernel, lds)) return e;
    const long HW = (long)H * W;
    // (launching the two passes per group of 8 / 4 / 2 tiles, so that a group's records - 17.7 MB per tile - could stay inside the 256 MB
    // Infinity Cache between them, measured 25.3 / 27.1 / 27.4 us per tile against 23.8 for all 16 at once: the launch tails cost more)
    // (round 4 also tried the band pass of group g on a helper stream beside the partition pass of group g + 1: it lost,
    // profiles/r4_raster_overlap_experiment.txt; removed in round 5)
    const int group = MAX_TILES;
    for (int b0 = 0; b0 < B; b0 += group) {
        const int nb = (B - b0) < group ? (B - b0) : group;
        BatchArgs A;
        BandArgs BA;
        long maxn = 0;
        for (int b = 0; b < nb; ++b) {
            const long n = tile_offsets[b0 + b + 1] - tile_offsets[b0 + b];
            derive(params[b0 + b], tile_offsets[b0 + b], n, A.tile[b]);
            BA.nblk[b] = (int)nblk_of(n);
            maxn = n > maxn ? n : maxn;
        }
        unsigned* cnt = counts + (long)b0 * nbands * nblk_max;
        unsigned* rec = records + (long)b0 * nbands * nblk_max * CHUNK;
        if (maxn > 0) {
            if (int e = lm_ensure_dynamic_lds((const void*)raster_partition_kernel, PART_LDS)) return e;
            hipLaunchKernelGGL(raster_partition_kernel, dim3((unsigned)nblk_of(maxn), nb), dim3(NT), PART_LDS, s,
                               reinterpret_cast<const f32x4*>(points_xyzi), A, cnt, rec, nblk_max, H, W, nbands, band_rows);
            LM_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(raster_band_kernel, dim3(nbands, nb), dim3(BT), lds, s, cnt, rec, BA, nblk_max,
                           out_chw ? out_chw + (long)b0 * 3 * HW : nullptr, out_hwc_u8 ? out_hwc_u8 + (long)b0 * 3 * HW : nullptr,
                           H, W, nbands, band_rows);
        LM_LAUNCH_CHECK();
    }
    return LM_OK;
}

LM_API int lm_tile_ingest_u8(void* stream, const unsigned char* src_hwc, float* dst_chw, int B, int H, int W, int C) {
    LM_REQUIRE(src_hwc && dst_chw && C >= 3, "tile_ingest: bad args (C=%d)", C);
    const long HW = (long)H * W, total = (long)B * HW;
    hipLaunchKernelGGL(ingest_kernel, dim3(lm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src_hwc, dst_chw, HW, C, total);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

// Device-wide exclusive scan and stable radix sort for gfx950 (64-wide waves), the two primitives the sparse-voxel path needs
// (lidar.hip: points -> voxels numbered by first appearance, output-site compaction of the sparse convolutions).  Round 1 called
// hipCUB for both; these are the library's own kernels.
//
// Scan: three phases over tiles of 4096 elements (256 threads x 16, blocked so a thread's 16 elements are 4 x 16-byte loads):
//   tile sums -> scan of the tile sums (recursively, at most three levels below 2^31 elements) -> tile scan with the tile's offset.
//   All arithmetic is u32 (mod 2^32), the order of the additions does not matter: any launch geometry gives the same bits.
// Sort: least-significant-digit radix sort, 8-bit digits.  Per pass: (1) per-tile digit histogram (LDS atomics) into
//   hist[digit][tile]; (2) exclusive scan of that table in digit-major order = the first output slot of every (digit, tile);
//   (3) scatter: a tile's 4 waves rank their 1024 elements each in index order - per 64-element chunk the lanes that share a digit
//   find each other with 8 ballots, rank = wave counter of the digit + lower lanes in the group - then the tile is reordered by digit
//   through LDS so that the copy-out writes runs of consecutive addresses.  Equal digits keep their index order inside a wave, across
//   the waves of a tile and across tiles: the sort is stable, hence deterministic.
#include "prim.h"

namespace {

constexpr int TILE = 4096, IPT = 16;          // elements per workgroup / per thread

__device__ __forceinline__ unsigned wave_excl_scan(unsigned v, unsigned& total) {
    // exclusive scan over the 64 lanes of a wave; total = sum over the wave
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    total = __shfl(incl, 63);
    return incl - v;
}

// exclusive scan of one value per thread over the 256 threads of a workgroup (lds: 4 words); total = workgroup sum
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* lds, unsigned& total) {
    unsigned wtot;
    const unsigned ex = wave_excl_scan(v, wtot);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) lds[w] = wtot;
    __syncthreads();
    unsigned off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned t = lds[k];
        if (k < w) off += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return ex + off;
}

__global__ __launch_bounds__(256) void scan_tile_sums_kernel(const unsigned* __restrict__ in, long n, unsigned* __restrict__ sums) {
    __shared__ unsigned lds[4];
    const long base = (long)blockIdx.x * TILE + (long)threadIdx.x * IPT;
    unsigned s = 0;
    if (base + IPT <= n) {
#pragma unroll
        for (int k = 0; k < IPT / 4; ++k) {
            const uint4 v = *reinterpret_cast<const uint4*>(in + base + 4 * k);
            s += v.x + v.y + v.z + v.w;
        }
    } else {
        for (int k = 0; k < IPT; ++k)
            if (base + k < n) s += in[base + k];
    }
    unsigned tot;
    block_excl_scan(s, lds, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// offsets == nullptr: a single tile
__global__ __launch_bounds__(256) void scan_tile_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, long n,
                                                        const unsigned* __restrict__ offsets) {
    __shared__ unsigned lds[4];
    const long base = (long)blockIdx.x * TILE + (long)threadIdx.x * IPT;
    unsigned v[IPT];
    const bool full = base + IPT <= n;
    if (full) {
#pragma unroll
        for (int k = 0; k < IPT / 4; ++k) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + base + 4 * k);
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < IPT; ++k) v[k] = base + k < n ? in[base + k] : 0u;
    }
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const unsigned t = v[k];
        v[k] = s;
        s += t;
    }
    unsigned tot;
    const unsigned off = block_excl_scan(s, lds, tot) + (offsets ? offsets[blockIdx.x] : 0u);
    if (full) {
#pragma unroll
        for (int k = 0; k < IPT / 4; ++k)
            *reinterpret_cast<uint4*>(out + base + 4 * k) = make_uint4(v[4 * k] + off, v[4 * k + 1] + off, v[4 * k + 2] + off, v[4 * k + 3] + off);
    } else {
#pragma unroll
        for (int k = 0; k < IPT; ++k)
            if (base + k < n) out[base + k] = v[k] + off;
    }
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// ---------------------------------------------------------------------------------------------------------------- radix sort
__global__ __launch_bounds__(256) void sort_hist_kernel(const unsigned* __restrict__ keys, long n, int shift, unsigned mask, unsigned nblocks,
                                                        unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * TILE;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, long n,
                                                           int shift, unsigned mask, unsigned nblocks, const unsigned* __restrict__ hist_scan,
                                                           unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    __shared__ unsigned cnt[4][256];          // per wave: elements of each digit seen so far (ends as the wave's digit counts)
    __shared__ unsigned dstart[256];          // first slot of a digit in the tile's digit-ordered staging
    __shared__ unsigned gbase[256];           // first global slot of (digit, this tile)
    __shared__ unsigned scan_lds[4];
    __shared__ unsigned stage_k[TILE], stage_v[TILE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt[k][tid] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * TILE + (long)w * (TILE / 4);
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    unsigned key[IPT], val[IPT], rank[IPT];
#pragma unroll
    for (int c = 0; c < IPT; ++c) {
        const long i = base + c * 64 + lane;
        const bool ok = i < n;
        key[c] = ok ? keys[i] : 0u;
        val[c] = ok ? vals[i] : 0u;
        const unsigned d = (key[c] >> shift) & mask;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot(ok && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned below = (unsigned)__popcll(peers & lt);
        const unsigned before = cnt[w][d];                      // every lane of the group reads the same counter ...
        __builtin_amdgcn_wave_barrier();
        if (ok && below == 0) cnt[w][d] = before + (unsigned)__popcll(peers);     // ... then its first lane advances it
        __builtin_amdgcn_wave_barrier();
        rank[c] = before + below;
    }
    __syncthreads();
    // per digit: the four waves' counts -> exclusive offsets of the waves, tile total -> exclusive scan over the digits
    unsigned tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned t = cnt[k][tid];
        cnt[k][tid] = tot;
        tot += t;
    }
    unsigned all;
    dstart[tid] = block_excl_scan(tot, scan_lds, all);
    gbase[tid] = hist_scan[(size_t)tid * nblocks + blockIdx.x];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < IPT; ++c) {
        const long i = base + c * 64 + lane;
        if (i < n) {
            const unsigned d = (key[c] >> shift) & mask;
            const unsigned pos = dstart[d] + cnt[w][d] + rank[c];
            stage_k[pos] = key[c];
            stage_v[pos] = val[c];
        }
    }
    __syncthreads();
    for (unsigned j = tid; j < all; j += 256) {
        const unsigned k = stage_k[j];
        const unsigned d = (k >> shift) & mask;
        const size_t o = (size_t)gbase[d] + (j - dstart[d]);
        keys_out[o] = k;
        vals_out[o] = stage_v[j];
    }
}

}  // namespace

size_t lm_prim_scan_temp_bytes(long n) {
    size_t b = 0;
    for (long m = (n + TILE - 1) / TILE; m > 1; m = (m + TILE - 1) / TILE) b += align256((size_t)m * 4);
    return b + 256;
}

int lm_prim_exclusive_scan_u32(hipStream_t s, const unsigned* in, unsigned* out, long n, void* temp, size_t temp_bytes) {
    if (n <= 0) return LM_OK;
    LM_REQUIRE(in && out, "exclusive_scan: null pointer");
    LM_REQUIRE(n < (1L << 40) && lm_prim_scan_temp_bytes(n) <= temp_bytes && (temp || n <= TILE), "exclusive_scan: scratch too small");
    const long nb = (n + TILE - 1) / TILE;
    if (nb == 1) {
        hipLaunchKernelGGL(scan_tile_kernel, dim3(1), dim3(256), 0, s, in, out, n, (const unsigned*)nullptr);
        LM_LAUNCH_CHECK();
        return LM_OK;
    }
    unsigned* sums = (unsigned*)temp;
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, n, sums);
    LM_LAUNCH_CHECK();
    const size_t used = align256((size_t)nb * 4);
    const int rc = lm_prim_exclusive_scan_u32(s, sums, sums, nb, (char*)temp + used, temp_bytes - used);      // (depth <= 3)
    if (rc != LM_OK) return rc;
    hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, out, n, (const unsigned*)sums);
    LM_LAUNCH_CHECK();
    return LM_OK;
}

size_t lm_prim_sort_temp_bytes(long n) {
    const long nb = (n + TILE - 1) / TILE;
    return align256((size_t)(nb > 0 ? nb : 1) * 256 * 4) + lm_prim_scan_temp_bytes((nb > 0 ? nb : 1) * 256);
}

int lm_prim_sort_pairs_u32(hipStream_t s, unsigned* keys, unsigned* keys_alt, unsigned* vals, unsigned* vals_alt, long n, int end_bit,
                           void* temp, size_t temp_bytes, unsigned** keys_res, unsigned** vals_res) {
    LM_REQUIRE(end_bit >= 0 && end_bit <= 32 && n >= 0 && n < (1L << 31), "sort_pairs: bad arguments (n=%ld, end_bit=%d)", n, end_bit);
    if (keys_res) *keys_res = keys;
    if (vals_res) *vals_res = vals;
    if (n == 0 || end_bit == 0) return LM_OK;
    LM_REQUIRE(keys && keys_alt && vals && vals_alt && temp, "sort_pairs: null pointer");
    LM_REQUIRE(lm_prim_sort_temp_bytes(n) <= temp_bytes, "sort_pairs: scratch too small");
    const unsigned nb = (unsigned)((n + TILE - 1) / TILE);
    unsigned* hist = (unsigned*)temp;
    const size_t hbytes = align256((size_t)nb * 256 * 4);
    unsigned *ki = keys, *ko = keys_alt, *vi = vals, *vo = vals_alt;
    for (int shift = 0; shift < end_bit; shift += 8) {
        // the last pass only looks at the bits below end_bit: keys may carry payload above it ([begin, end) semantics of the contract)
        const unsigned mask = (1u << (end_bit - shift < 8 ? end_bit - shift : 8)) - 1u;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nb), dim3(256), 0, s, ki, n, shift, mask, nb, hist);
        LM_LAUNCH_CHECK();
        const int rc = lm_prim_exclusive_scan_u32(s, hist, hist, (long)nb * 256, (char*)temp + hbytes, temp_bytes - hbytes);
        if (rc != LM_OK) return rc;
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nb), dim3(256), 0, s, ki, vi, n, shift, mask, nb, hist, ko, vo);
        LM_LAUNCH_CHECK();
        unsigned* t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    if (keys_res) *keys_res = ki;
    if (vals_res) *vals_res = vi;
    return LM_OK;
}

LM_API long lm_scan_workspace_bytes(long n) { return (long)lm_prim_scan_temp_bytes(n > 0 ? n : 1); }

// out[i] = sum of in[0 .. i) (u32, wraps): device pointers, in == out allowed
LM_API int lm_exclusive_scan_u32(void* stream, const unsigned* in, unsigned* out, long n, void* workspace, long workspace_bytes) {
    LM_REQUIRE(n >= 0 && workspace_bytes >= 0, "exclusive_scan: bad sizes");
    return lm_prim_exclusive_scan_u32((hipStream_t)stream, in, out, n, workspace, (size_t)workspace_bytes);
}

LM_API long lm_sort_pairs_workspace_bytes(long n) { return (long)(lm_prim_sort_temp_bytes(n > 0 ? n : 1) + 2 * align256((size_t)(n > 0 ? n : 1) * 4)); }

// Stable sort of (key, value) pairs by the low end_bit bits of the keys; keys_io / vals_io are sorted in place (device pointers).
LM_API int lm_sort_pairs_u32(void* stream, unsigned* keys_io, unsigned* vals_io, long n, int end_bit, void* workspace, long workspace_bytes) {
    LM_REQUIRE(n >= 0 && lm_sort_pairs_workspace_bytes(n) <= workspace_bytes, "sort_pairs: workspace too small (%ld B needed)",
               lm_sort_pairs_workspace_bytes(n));
    if (n == 0) return LM_OK;
    LM_REQUIRE(workspace, "sort_pairs: null workspace");
    const size_t seg = align256((size_t)n * 4);
    char* w = (char*)workspace;
    unsigned *ka = (unsigned*)w, *va = (unsigned*)(w + seg), *kr = nullptr, *vr = nullptr;
    const int rc = lm_prim_sort_pairs_u32((hipStream_t)stream, keys_io, ka, vals_io, va, n, end_bit, w + 2 * seg, (size_t)workspace_bytes - 2 * seg,
                                          &kr, &vr);
    if (rc != LM_OK) return rc;
    if (kr != keys_io) {
        LM_HIP(hipMemcpyAsync(keys_io, kr, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        LM_HIP(hipMemcpyAsync(vals_io, vr, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return LM_OK;
}

// Device-wide primitives of the library, hand-written for gfx950 (prim.hip): exclusive prefix sum and stable LSD radix sort of
// (u32 key, u32 value) pairs.  Used by the voxeliser / sparse-convolution rulebooks (lidar.hip); exported through the C-ABI as
// lm_exclusive_scan_u32 / lm_sort_pairs_u32 for the tests.
#pragma once
#include "common.h"

// bytes of scratch the calls below need for n elements
size_t lm_prim_scan_temp_bytes(long n);
size_t lm_prim_sort_temp_bytes(long n);

// out[i] = in[0] + .. + in[i-1] (mod 2^32); in == out allowed
int lm_prim_exclusive_scan_u32(hipStream_t s, const unsigned* in, unsigned* out, long n, void* temp, size_t temp_bytes);

// Stable sort by the low `end_bit` bits of the keys (8-bit digits, ceil(end_bit / 8) passes, ping-pong between the two buffer pairs:
// BOTH are overwritten).  *keys_res / *vals_res receive the pair that holds the result.
int lm_prim_sort_pairs_u32(hipStream_t s, unsigned* keys, unsigned* keys_alt, unsigned* vals, unsigned* vals_alt, long n, int end_bit,
                           void* temp, size_t temp_bytes, unsigned** keys_res, unsigned** vals_res);

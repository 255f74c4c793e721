// PNG tile reader: the reference's `load_img` (datasets/laserlane_proposals.py:85-98, laserlane.py:214-219) is
// `np.array(Image.open(path))` -> uint8 HWC; this is the same decode without PIL and without the GIL, so a batch of tiles is inflated on
// the host thread pool while the GPU works on the previous batch.  The DEFLATE decoder is this library's own (inflate.h: two literals per
// table lookup - a BEV tile is sensor noise that deflate stores as short literal codes - ~2.3 x zlib's inflate on such a tile).
// Scope = what BEV tiles are: 8-bit, non-interlaced, greyscale / grey+alpha / RGB / RGBA.  Everything else (palette, 16-bit, Adam7) is
// refused with a message, never guessed.  Chunk CRCs and the zlib Adler checksum are verified: a damaged tile is an error, not noise.
// Host code only (no HIP calls).
#include "common.h"
#include "inflate.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

const unsigned char PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

struct PngHeader {
    int W, H, depth, color, interlace, channels;
};

// returns nullptr on success, else a message
const char* parse_header(const unsigned char* d, long n, PngHeader& h) {
    if (n < 8 + 25 || memcmp(d, PNG_SIG, 8) != 0) return "not a PNG file (bad signature)";
    if (be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) return "first chunk is not IHDR";
    const unsigned char* p = d + 16;
    const unsigned w = be32(p), hh = be32(p + 4);
    if (w == 0 || hh == 0 || w > 65535u || hh > 65535u) return "image size out of range";
    h.W = (int)w;
    h.H = (int)hh;
    h.depth = p[8];
    h.color = p[9];
    h.interlace = p[12];
    if (p[10] != 0 || p[11] != 0) return "unknown compression / filter method";
    if (h.depth != 8) return "only 8-bit PNG tiles are supported";
    if (h.interlace != 0) return "interlaced (Adam7) PNG is not supported";
    switch (h.color) {
        case 0: h.channels = 1; break;
        case 2: h.channels = 3; break;
        case 4: h.channels = 2; break;
        case 6: h.channels = 4; break;
        case 3: return "palette PNG is not supported (BEV tiles are RGB)";
        default: return "bad colour type";
    }
    if ((unsigned long long)w * hh * (unsigned)h.channels > (1ull << 30)) return "image larger than 2^30 bytes (not a BEV tile)";
    return nullptr;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Paeth rows of 3- and 4-byte pixels: one pixel per step in 16-bit lanes, the predictor chosen by compares and masks instead of
// branches (on sensor noise the three-way choice is a coin toss: the scalar loop spends most of its time on mispredictions).  The
// only serial dependency is the pixel to the left.  Row ends are handled bytewise by the caller's scalar loop.
#if defined(__SSE2__)
// one pixel (16-bit lanes): the new `a`
inline __m128i paeth_px(const __m128i a, const __m128i b, const __m128i c, const __m128i x) {
    const __m128i zero = _mm_setzero_si128();
    __m128i pa = _mm_sub_epi16(b, c), pb = _mm_sub_epi16(a, c);
    __m128i pc = _mm_add_epi16(pa, pb);
    pa = _mm_max_epi16(pa, _mm_sub_epi16(zero, pa));
    pb = _mm_max_epi16(pb, _mm_sub_epi16(zero, pb));
    pc = _mm_max_epi16(pc, _mm_sub_epi16(zero, pc));
    const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
    const __m128i is_a = _mm_cmpeq_epi16(smallest, pa), is_b = _mm_cmpeq_epi16(smallest, pb);
    // a where pa is the minimum, else b where pb is, else c (ties go to a, then b: the order of the specification)
    const __m128i bc = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
    const __m128i pred = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, bc));
    return _mm_add_epi8(x, pred);                                  // (bytewise: the sum wraps inside the low byte, the high byte stays 0)
}
inline __m128i load_px16(const unsigned char* p) {
    int v;
    memcpy(&v, p, 4);
    return _mm_unpacklo_epi8(_mm_cvtsi32_si128(v), _mm_setzero_si128());
}
template <int BP>
inline void store_px16(unsigned char* p, const __m128i a) {
    const int out = _mm_cvtsi128_si32(_mm_packus_epi16(a, a));
    memcpy(p, &out, BP);
}

template <int BP>
inline size_t paeth_row_simd(const unsigned char* src, unsigned char* cur, const unsigned char* up, size_t stride) {
    // pixels [1, n): pixel 0 has no left neighbour; the last pixel is left to the scalar loop (4-byte loads / stores stay inside the row)
    if (stride < (size_t)BP * 3) return BP;
    __m128i a = load_px16(cur), c = load_px16(up);                // pixel 0 (already unfiltered by the caller) and the one above it
    size_t i = BP;
    for (; i + 4 <= stride - (BP == 3 ? BP : 0); i += BP) {
        const __m128i b = load_px16(up + i);
        a = paeth_px(a, b, c, load_px16(src + i));
        c = b;
        store_px16<BP>(cur + i, a);
    }
    return i;
}

// TWO consecutive Paeth rows as a wavefront: the lower row runs one pixel behind the upper one, so that its `b` (the upper row's pixel
// above) is the upper row's previous result and its `c` the one before - two independent dependency chains per step instead of one
// (a Paeth row is latency-bound: ~13 dependent cycles per pixel).  cur0 / cur1 = the two output rows (cur1 = cur0 + stride), up0 = the row above.
template <int BP>
inline void paeth_two_rows_simd(const unsigned char* src0, unsigned char* cur0, const unsigned char* up0, const unsigned char* src1,
                                unsigned char* cur1, size_t stride) {
    for (size_t k = 0; k < (size_t)BP; ++k) cur0[k] = (unsigned char)(src0[k] + up0[k]);                        // upper row, pixel 0
    for (size_t k = BP; k < 2 * (size_t)BP; ++k) cur0[k] = (unsigned char)(src0[k] + paeth(cur0[k - BP], up0[k], up0[k - BP]));   // pixel 1
    for (size_t k = 0; k < (size_t)BP; ++k) cur1[k] = (unsigned char)(src1[k] + cur0[k]);                        // lower row, pixel 0
    __m128i a0 = load_px16(cur0 + BP), c0 = load_px16(up0 + BP);
    __m128i a1 = load_px16(cur1), c1 = load_px16(cur0);
    size_t i = 2 * BP;                                            // byte offset of the upper row's next pixel; the lower row is at i - BP
    for (; i + 4 <= stride - (BP == 3 ? BP : 0); i += BP) {
        const __m128i b0 = load_px16(up0 + i), b1 = a0;
        const __m128i n0 = paeth_px(a0, b0, c0, load_px16(src0 + i));
        const __m128i n1 = paeth_px(a1, b1, c1, load_px16(src1 + i - BP));
        c0 = b0;
        c1 = b1;
        a0 = n0;
        a1 = n1;
        store_px16<BP>(cur0 + i, a0);
        store_px16<BP>(cur1 + i - BP, a1);
    }
    for (size_t k = i; k < stride; ++k) cur0[k] = (unsigned char)(src0[k] + paeth(cur0[k - BP], up0[k], up0[k - BP]));
    for (size_t k = i - BP; k < stride; ++k) cur1[k] = (unsigned char)(src1[k] + paeth(cur1[k - BP], cur0[k], cur0[k - BP]));
}
#endif

#if defined(__SSE2__)
// Sub rows of 3- and 4-byte pixels as a prefix sum, four pixels (12 / 16 bytes) per step: the sums inside the group come from two shifted
// adds, the running value of the pixel to the left of the group rides in `carry`, already broadcast to the four pixel slots - the only
// serial dependency is one add per group.  (The generic loop reads cur[i - bp] back from memory: a store-to-load round trip per byte.)
// Returns the index where the bytewise loop takes over.
template <int BP>
inline size_t sub_row_simd(const unsigned char* src, unsigned char* cur, size_t stride) {
    constexpr int G = 4 * BP;                                     // bytes per group
    if (stride < (size_t)BP + 16) return BP;
    const __m128i px_mask = BP == 3 ? _mm_set_epi32(0, 0, 0, 0x00FFFFFF) : _mm_set_epi32(0, 0, 0, -1);
    auto bcast = [&](__m128i lastpx) {                            // pixel in bytes 0 .. BP-1 -> the four pixel slots
        if constexpr (BP == 4) return _mm_shuffle_epi32(lastpx, 0);
        __m128i x = _mm_and_si128(lastpx, px_mask);
        x = _mm_or_si128(x, _mm_slli_si128(x, 3));
        return _mm_or_si128(x, _mm_slli_si128(x, 6));
    };
    int first;
    memcpy(&first, cur, 4);                                       // pixel 0 (already in place; BP == 3: one byte too many, masked)
    __m128i carry = bcast(_mm_cvtsi32_si128(first));
    size_t i = BP;
    for (; i + 16 <= stride; i += G) {                            // 16-byte loads / stores stay inside the row
        __m128i t = _mm_loadu_si128((const __m128i*)(src + i));
        t = _mm_add_epi8(t, _mm_slli_si128(t, BP));
        t = _mm_add_epi8(t, _mm_slli_si128(t, 2 * BP));           // bytes 0 .. G-1: sums of pixels 0 .. k of the group
        _mm_storeu_si128((__m128i*)(cur + i), _mm_add_epi8(t, carry));     // (BP == 3: bytes 12-15 are rewritten by the next group / the tail loop)
        carry = _mm_add_epi8(carry, bcast(_mm_srli_si128(t, 3 * BP)));
    }
    return i;
}

// Average rows of 3- and 4-byte pixels, one pixel per step in 16-bit lanes: (a + b) >> 1 needs nine bits; the bytewise add of the
// filtered value wraps inside the low byte and leaves the high byte 0 - three dependent instructions per pixel (add, shift, add)
template <int BP>
inline size_t avg_row_simd(const unsigned char* src, unsigned char* cur, const unsigned char* up, size_t stride) {
    if (stride < (size_t)BP * 3) return BP;
    const __m128i zero = _mm_setzero_si128();
    auto load = [&](const unsigned char* p) {
        int v;
        memcpy(&v, p, 4);
        return _mm_unpacklo_epi8(_mm_cvtsi32_si128(v), zero);
    };
    __m128i a = load(cur);
    size_t i = BP;
    for (; i + 4 <= stride - (BP == 3 ? BP : 0); i += BP) {
        const __m128i b = load(up + i), x = load(src + i);
        a = _mm_add_epi8(x, _mm_srli_epi16(_mm_add_epi16(a, b), 1));
        const int out = _mm_cvtsi128_si32(_mm_packus_epi16(a, a));
        memcpy(cur + i, &out, BP);
    }
    return i;
}
#endif

// Reverses the scanline filter `ft` of one row in place of `src` -> `cur` (`up` = previous output row or nullptr).
inline bool unfilter_row(int ft, const unsigned char* src, unsigned char* cur, const unsigned char* up, size_t stride, size_t bp) {
    // the first pixel has no left neighbour; the rest of the row runs without per-byte conditions
    switch (ft) {
        case 0: memcpy(cur, src, stride); return true;
        case 1: {
            memcpy(cur, src, bp);
            size_t i = bp;
#if defined(__SSE2__)
            if (bp == 3) i = sub_row_simd<3>(src, cur, stride);
            else if (bp == 4) i = sub_row_simd<4>(src, cur, stride);
#endif
            for (; i < stride; ++i) cur[i] = (unsigned char)(src[i] + cur[i - bp]);
            return true;
        }
        case 2:
            if (!up) memcpy(cur, src, stride);
            else
                for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(src[i] + up[i]);
            return true;
        case 3:
            if (!up) {
                memcpy(cur, src, bp);
                for (size_t i = bp; i < stride; ++i) cur[i] = (unsigned char)(src[i] + (cur[i - bp] >> 1));
            } else {
                for (size_t i = 0; i < bp; ++i) cur[i] = (unsigned char)(src[i] + (up[i] >> 1));
                size_t i = bp;
#if defined(__SSE2__)
                if (bp == 3) i = avg_row_simd<3>(src, cur, up, stride);
                else if (bp == 4) i = avg_row_simd<4>(src, cur, up, stride);
#endif
                for (; i < stride; ++i) cur[i] = (unsigned char)(src[i] + ((cur[i - bp] + up[i]) >> 1));
            }
            return true;
        case 4:
            if (!up) {                                           // b = c = 0 -> predictor = a: a Sub row
                return unfilter_row(1, src, cur, up, stride, bp);
            } else {
                for (size_t i = 0; i < bp; ++i) cur[i] = (unsigned char)(src[i] + up[i]);          // a = c = 0 -> predictor = b
                size_t i = bp;
#if defined(__SSE2__)
                if (bp == 3) i = paeth_row_simd<3>(src, cur, up, stride);
                else if (bp == 4) i = paeth_row_simd<4>(src, cur, up, stride);
#endif
                for (; i < stride; ++i) cur[i] = (unsigned char)(src[i] + paeth(cur[i - bp], up[i], up[i - bp]));
            }
            return true;
        default: return false;
    }
}

// Scratch of one decode: the Huffman tables (52 KB), the filtered image (H (stride + 1) bytes - DEFLATE matches reach 32 KB back into
// it), the file image and, for files with more than one IDAT chunk, the joined zlib stream.  The batch entry starts its worker threads
// per call, so the blocks are kept in a process-wide pool instead of thread_local storage: ~10 MB per concurrently decoding thread,
// touched once (no page faults per batch).
struct Scratch {
    lm_inflate::Decoder dec;
    std::vector<unsigned char> filtered, joined, file;
};

struct ScratchPool {
    std::mutex m;
    std::vector<Scratch*> idle;
    Scratch* take() {
        {
            std::lock_guard<std::mutex> g(m);
            if (!idle.empty()) {
                Scratch* s = idle.back();
                idle.pop_back();
                return s;
            }
        }
        return new Scratch();
    }
    void give(Scratch* s) {
        std::lock_guard<std::mutex> g(m);
        idle.push_back(s);
    }
};

ScratchPool& scratch_pool() {
    static ScratchPool* p = new ScratchPool();      // (never destroyed: worker threads may outlive static destructors at exit)
    return *p;
}

struct ScratchLease {
    Scratch* s;
    ScratchLease() : s(scratch_pool().take()) {}
    ~ScratchLease() { scratch_pool().give(s); }
    ScratchLease(const ScratchLease&) = delete;
    ScratchLease& operator=(const ScratchLease&) = delete;
};

// out: [H][W][channels]; returns nullptr on success.  Chunk walk (CRC of every chunk checked) -> the zlib stream of the IDAT chunks ->
// inflate into the filtered image; after every DEFLATE block the scanlines that are complete are unfiltered into `out` while they are
// still in cache.
const char* decode(const unsigned char* d, long n, const PngHeader& h, unsigned char* out, Scratch& sc) {
    const size_t bp = (size_t)h.channels, stride = (size_t)h.W * bp, need = (stride + 1) * (size_t)h.H;
    const unsigned char* z = nullptr;          // the zlib stream: in place when the file has one IDAT chunk
    size_t zn = 0;
    int idats = 0;
    bool end = false;
    long pos = 8;
    while (!end) {
        if (pos + 12 > n) return "truncated file (chunk header)";
        const unsigned len = be32(d + pos);
        const unsigned char* type = d + pos + 4;
        if ((long)len > n - pos - 12) return "truncated file (chunk data)";
        if (lm_inflate::crc32(type, (size_t)len + 4) != be32(d + pos + 8 + len)) return "chunk CRC mismatch";
        if (memcmp(type, "IDAT", 4) == 0) {
            if (idats == 0) {
                z = d + pos + 8;
                zn = len;
            } else {
                if (idats == 1) sc.joined.assign(z, z + zn);
                sc.joined.insert(sc.joined.end(), d + pos + 8, d + pos + 8 + len);
            }
            ++idats;
        } else if (memcmp(type, "IEND", 4) == 0) {
            end = true;
        } else if (!(type[0] & 0x20) && memcmp(type, "IHDR", 4) != 0 && memcmp(type, "PLTE", 4) != 0) {
            return "unknown critical chunk";
        }
        pos += 12 + (long)len;
    }
    if (!idats) return "no IDAT chunk";
    if (idats > 1) {
        z = sc.joined.data();
        zn = sc.joined.size();
    }
    if (sc.filtered.size() < need) sc.filtered.resize(need);
    unsigned char* const f = sc.filtered.data();
    size_t row = 0, produced = 0;
    bool bad_filter = false;
    auto unfilter_done_rows = [&](size_t bytes) {
        for (; !bad_filter && (row + 1) * (stride + 1) <= bytes; ++row) {
            unsigned char* cur = out + stride * row;
            const unsigned char* src = f + row * (stride + 1);
#if defined(__SSE2__)
            // two Paeth rows in a row, both complete: as a wavefront (paeth_two_rows_simd)
            if (src[0] == 4 && row > 0 && (bp == 3 || bp == 4) && stride >= 4 * bp && (row + 2) * (stride + 1) <= bytes && src[stride + 1] == 4) {
                if (bp == 3) paeth_two_rows_simd<3>(src + 1, cur, cur - stride, src + stride + 2, cur + stride, stride);
                else paeth_two_rows_simd<4>(src + 1, cur, cur - stride, src + stride + 2, cur + stride, stride);
                ++row;
                continue;
            }
#endif
            if (!unfilter_row(src[0], src + 1, cur, row ? cur - stride : nullptr, stride, bp)) bad_filter = true;
        }
    };
    const char* e = sc.dec.zlib_inflate(z, zn, f, need, &produced, unfilter_done_rows);
    if (bad_filter) return "bad scanline filter type";
    if (e) return e;
    if (produced != need) return "image data shorter than the header says";
    return nullptr;
}

const char* read_file(const char* path, std::vector<unsigned char>& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return "cannot open file";
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) {
        fclose(f);
        return "empty file";
    }
    buf.resize((size_t)n);
    const size_t got = fread(buf.data(), 1, (size_t)n, f);
    fclose(f);
    return got == (size_t)n ? nullptr : "short read";
}

}  // namespace

// Header of an in-memory PNG: height, width, channels (1 grey, 2 grey+alpha, 3 RGB, 4 RGBA).
LM_API int lm_png_info(const unsigned char* data, long size, int* H, int* W, int* C) {
    LM_REQUIRE(data && H && W && C, "png_info: null pointer");
    PngHeader h;
    const char* e = parse_header(data, size, h);
    LM_REQUIRE(!e, "png_info: %s", e);
    *H = h.H;
    *W = h.W;
    *C = h.channels;
    return LM_OK;
}

// In-memory PNG -> out [H][W][C] uint8 (the array np.array(Image.open(...)) returns); out_bytes must be H*W*C of lm_png_info.
LM_API int lm_png_decode_u8(const unsigned char* data, long size, unsigned char* out, long out_bytes) {
    LM_REQUIRE(data && out, "png_decode: null pointer");
    PngHeader h;
    const char* e = parse_header(data, size, h);
    LM_REQUIRE(!e, "png_decode: %s", e);
    LM_REQUIRE(out_bytes == (long)h.H * h.W * h.channels, "png_decode: output buffer is %ld bytes, the image needs %ld", out_bytes,
               (long)h.H * h.W * h.channels);
    {
        ScratchLease lease;
        e = decode(data, size, h, out, *lease.s);
    }
    LM_REQUIRE(!e, "png_decode: %s", e);
    return LM_OK;
}

// A zlib stream (RFC 1950: the payload of the IDAT chunks) -> out[0..capacity); *produced = inflated bytes.  The decoder the PNG reader runs
// on, exposed so that it can be held against any other inflate (tests: Python's zlib on every block type, level and strategy).
LM_API int lm_zlib_inflate(const unsigned char* data, long size, unsigned char* out, long capacity, long* produced) {
    LM_REQUIRE(data && (out || capacity == 0) && produced && size >= 0 && capacity >= 0, "zlib_inflate: bad arguments");
    ScratchLease lease;
    unsigned char none = 0;
    size_t got = 0;
    const char* e = lease.s->dec.zlib_inflate(data, (size_t)size, out ? out : &none, (size_t)capacity, &got, [](size_t) {});
    *produced = (long)got;
    LM_REQUIRE(!e, "zlib_inflate: %s", e);
    return LM_OK;
}

// n files of identical geometry -> out [n][H][W][C] uint8, decoded on `threads` host threads (>= 1).  Every file must be H x W with C
// channels (a tile set is homogeneous); the first offending file is named in lm_last_error.
LM_API int lm_png_decode_files_u8(const char* const* paths, int n, unsigned char* out, int H, int W, int C, int threads) {
    LM_REQUIRE(paths && out && n >= 0 && H > 0 && W > 0 && C >= 1 && C <= 4 && threads >= 1, "png_decode_files: bad arguments");
    std::atomic<int> next{0}, failed{-1};
    std::vector<const char*> msg((size_t)n, nullptr);
    const size_t per = (size_t)H * W * C;
    auto work = [&]() {
        ScratchLease lease;
        std::vector<unsigned char>& buf = lease.s->file;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            const char* e = read_file(paths[i], buf);
            PngHeader h{};
            if (!e) e = parse_header(buf.data(), (long)buf.size(), h);
            if (!e && (h.H != H || h.W != W || h.channels != C)) e = "geometry differs from the requested H x W x C";
            if (!e) e = decode(buf.data(), (long)buf.size(), h, out + per * (size_t)i, *lease.s);
            if (e) {
                msg[(size_t)i] = e;
                int expect = -1;
                failed.compare_exchange_strong(expect, i);
            }
        }
    };
    const int nt = threads < n ? threads : (n > 0 ? n : 1);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    for (int i = 0; i < n; ++i)
        LM_REQUIRE(!msg[(size_t)i], "png_decode_files: %s: %s", paths[i], msg[(size_t)i]);
    return LM_OK;
}

// DEFLATE / zlib stream decoder for the PNG tile reader (png_reader.cpp; SURVEY.md §8f row f4 - the reference's `load_img` is PIL over
// zlib, datasets/laserlane_proposals.py:85-98).  Host code, header only.  Written for what a BEV tile is: a stream dominated by short
// literal codes (sensor noise deflate cannot match) or by long runs (empty area).  One table lookup resolves up to TWO literals (an entry
// of the 12-bit primary table holds the pair when both codes fit), the bit buffer is 64 bits wide and refilled without a branch, matches
// are copied a word at a time.  RFC 1950 / 1951 complete: stored, fixed and dynamic blocks, the Adler-32 trailer is verified, every
// malformed input (over-subscribed or incomplete code, distance before the start of the output, missing end-of-block code, truncation,
// trailing bytes) is refused with a message and nothing is ever read or written outside the two buffers.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace lm_inflate {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------------------------- checksums
// CRC-32 (PNG chunk checksum, ISO 3309 polynomial 0xEDB88320), slicing-by-16.
struct Crc32Tables {
    u32 t[16][256];
    Crc32Tables() {
        for (u32 i = 0; i < 256; ++i) {
            u32 c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (u32 i = 0; i < 256; ++i)
            for (int s = 1; s < 16; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 255u];
    }
};

inline u32 crc32(const u8* p, size_t n, u32 crc = 0) {
    static const Crc32Tables T;
    u32 c = ~crc;
    for (; n >= 16; n -= 16, p += 16) {
        u32 w[4];
        memcpy(w, p, 16);
        w[0] ^= c;
        c = T.t[15][w[0] & 255u] ^ T.t[14][(w[0] >> 8) & 255u] ^ T.t[13][(w[0] >> 16) & 255u] ^ T.t[12][w[0] >> 24] ^
            T.t[11][w[1] & 255u] ^ T.t[10][(w[1] >> 8) & 255u] ^ T.t[9][(w[1] >> 16) & 255u] ^ T.t[8][w[1] >> 24] ^
            T.t[7][w[2] & 255u] ^ T.t[6][(w[2] >> 8) & 255u] ^ T.t[5][(w[2] >> 16) & 255u] ^ T.t[4][w[2] >> 24] ^
            T.t[3][w[3] & 255u] ^ T.t[2][(w[3] >> 8) & 255u] ^ T.t[1][(w[3] >> 16) & 255u] ^ T.t[0][w[3] >> 24];
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 255u];
    return ~c;
}

// Adler-32 of the inflated data.  Blocks of 32 bytes: with S = s1 at the start of a run of blocks and P_j = the bytes summed before
// block j, s2 advances by sum_j (32 (S + P_j) + sum_k (32 - k) p_j[k]).  SSE2 (part of every x86-64): psadbw gives the byte sums,
// pmaddwd the weighted ones.  A run is at most 5536 bytes (the largest multiple of 32 that keeps the scalar tail below zlib's 5552).
inline u32 adler32(const u8* p, size_t n, u32 adler = 1) {
    u32 s1 = adler & 0xFFFFu, s2 = adler >> 16;
    while (n) {
        size_t chunk = n < 5536 ? n : 5536;
        n -= chunk;
#if defined(__SSE2__)
        if (chunk >= 32) {
            const __m128i zero = _mm_setzero_si128();
            const __m128i w0 = _mm_set_epi16(25, 26, 27, 28, 29, 30, 31, 32), w1 = _mm_set_epi16(17, 18, 19, 20, 21, 22, 23, 24);
            const __m128i w2 = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), w3 = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
            __m128i vsum = zero, vweighted = zero, vprefix = zero;
            const u32 blocks = (u32)(chunk / 32);
            for (u32 j = 0; j < blocks; ++j, p += 32) {
                const __m128i a = _mm_loadu_si128((const __m128i*)p), b = _mm_loadu_si128((const __m128i*)(p + 16));
                vprefix = _mm_add_epi32(vprefix, vsum);
                vsum = _mm_add_epi32(vsum, _mm_add_epi32(_mm_sad_epu8(a, zero), _mm_sad_epu8(b, zero)));
                const __m128i m0 = _mm_madd_epi16(_mm_unpacklo_epi8(a, zero), w0), m1 = _mm_madd_epi16(_mm_unpackhi_epi8(a, zero), w1);
                const __m128i m2 = _mm_madd_epi16(_mm_unpacklo_epi8(b, zero), w2), m3 = _mm_madd_epi16(_mm_unpackhi_epi8(b, zero), w3);
                vweighted = _mm_add_epi32(vweighted, _mm_add_epi32(_mm_add_epi32(m0, m1), _mm_add_epi32(m2, m3)));
            }
            chunk -= (size_t)blocks * 32;
            u32 ts[4], tw[4], tp[4];
            _mm_storeu_si128((__m128i*)ts, vsum);
            _mm_storeu_si128((__m128i*)tw, vweighted);
            _mm_storeu_si128((__m128i*)tp, vprefix);
            const u64 prefix = (u64)tp[0] + tp[2];               // (psadbw leaves its two sums in lanes 0 and 2)
            const u64 weighted = (u64)tw[0] + tw[1] + tw[2] + tw[3];
            s2 = (u32)((s2 + 32ull * blocks * s1 + 32ull * prefix + weighted) % 65521u);
            s1 += ts[0] + ts[2];
        }
#endif
        for (; chunk; --chunk) {
            s1 += *p++;
            s2 += s1;
        }
        s1 %= 65521u;
        s2 %= 65521u;
    }
    return (s2 << 16) | s1;
}

// ---------------------------------------------------------------------------------------------------------------- tables
// Entry (u32): bits 0-5 = stream bits this entry consumes (code + extra bits: one shift moves the bit buffer on, the extra value is read
// off the dependency chain); bits 10-12 = kind; bits 16-31 = payload.  K_LIT: bits 8-9 = literals it yields (1 or 2), byte 0 in 16-23,
// byte 1 in 24-31.  K_LEN: base 16-24, extra-bit count 25-27.  K_DIST: base 16-30, extra-bit count 6-9.  K_SUB: width of the sub-table
// index 6-9, its first entry 16-31 (a sub-table entry carries the bits of the whole code).
enum : u32 { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };
constexpr int LIT_TB = 12, DIST_TB = 8, MAX_BITS = 15;
constexpr int LIT_ENTRIES = (1 << LIT_TB) + 288 * 16, DIST_ENTRIES = (1 << DIST_TB) + 32 * 128;
constexpr u32 BAD_ENTRY = (K_BAD << 10) | 1u;

inline u32 kind_of(u32 e) { return (e >> 10) & 7u; }

struct Tables {
    u32 lit[LIT_ENTRIES];
    u32 dist[DIST_ENTRIES];
};

inline u32 rev_bits(u32 code, int len) {
    u32 r = 0;
    for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman code `lens[0..n)` -> lookup table with `tb` primary bits.  make(sym, len) gives the entry of a symbol without its
// bit count.  Returns nullptr or a message.  Incomplete codes are accepted only where zlib accepts them (`allow_single`: one code of
// one bit, or no code at all - a block without matches); what they leave unassigned decodes to K_BAD.
template <class Make>
const char* build_table(const u8* lens, int n, int tb, u32* table, int capacity, bool allow_single, Make make) {
    int count[MAX_BITS + 1] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    int used = n - count[0], maxlen = 0;
    long left = 1;
    for (int l = 1; l <= MAX_BITS; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return "zlib stream is corrupt (over-subscribed Huffman code)";
        if (count[l]) maxlen = l;
    }
    if (left > 0 && !(allow_single && (used == 0 || (used == 1 && maxlen == 1)))) return "zlib stream is corrupt (incomplete Huffman code)";
    const int primary = 1 << tb;
    for (int i = 0; i < primary; ++i) table[i] = BAD_ENTRY;
    u32 next[MAX_BITS + 2];
    next[1] = 0;
    for (int l = 1; l <= MAX_BITS; ++l) next[l + 1] = (next[l] + (u32)count[l]) << 1;
    // widest code below each primary index that needs a sub-table
    u8 sub_bits[1 << LIT_TB];
    bool any_long = maxlen > tb;
    if (any_long) {
        memset(sub_bits, 0, (size_t)primary);
        u32 nx[MAX_BITS + 2];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const u32 r = rev_bits(nx[l]++, l);
            if (l > tb && sub_bits[r & (u32)(primary - 1)] < l - tb) sub_bits[r & (u32)(primary - 1)] = (u8)(l - tb);
        }
    }
    int top = primary;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const u32 r = rev_bits(next[l]++, l);
        if (l <= tb) {
            const u32 e = make(s) + (u32)l;
            for (u32 i = r; i < (u32)primary; i += 1u << l) table[i] = e;
        } else {
            const u32 pi = r & (u32)(primary - 1);
            const int sb = sub_bits[pi];
            if (kind_of(table[pi]) != K_SUB) {
                if (top + (1 << sb) > capacity) return "zlib stream is corrupt (Huffman table overflow)";
                table[pi] = (K_SUB << 10) | ((u32)sb << 6) | ((u32)top << 16);      // (width in bits 6-9, first entry in 16-31)
                for (int i = 0; i < (1 << sb); ++i) table[top + i] = BAD_ENTRY;
                top += 1 << sb;
            }
            const u32 base = table[pi] >> 16;
            const u32 e = make(s) + (u32)l;
            for (u32 i = r >> tb; i < (1u << sb); i += 1u << (l - tb)) table[base + i] = e;
        }
    }
    return nullptr;
}

inline u32 sub_width(u32 e) { return (e >> 6) & 15u; }

const u16 LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const u8 LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const u16 DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
                           12289, 16385, 24577};
const u8 DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline u32 make_litlen(int s) {
    if (s < 256) return (K_LIT << 10) | (1u << 8) | ((u32)s << 16);
    if (s == 256) return K_EOB << 10;
    if (s > 285) return K_BAD << 10;                            // 286 / 287 take part in the fixed code but never appear in a stream
    return (K_LEN << 10) | ((u32)LEN_BASE[s - 257] << 16) | ((u32)LEN_EXTRA[s - 257] << 25) | LEN_EXTRA[s - 257];
}
inline u32 make_dist(int s) {
    if (s > 29) return K_BAD << 10;
    return (K_DIST << 10) | ((u32)DIST_BASE[s] << 16) | ((u32)DIST_EXTRA[s] << 6) | DIST_EXTRA[s];
}
inline u32 dist_extra(u32 e) { return (e >> 6) & 15u; }

// Two literals per lookup: where a one-literal entry leaves room in the primary index for a second complete literal code, fold it in.
// Descending order: entry i only looks at entry i >> len <= i, which is still in its one-literal form.
inline void pair_literals(u32* lit) {
    for (int i = (1 << LIT_TB) - 1; i >= 0; --i) {
        const u32 e = lit[i];
        if ((e & 0x1F00u) != ((K_LIT << 10) | (1u << 8))) continue;
        const int l1 = (int)(e & 63u);
        if (l1 >= LIT_TB) continue;
        const u32 e2 = lit[i >> l1];
        if ((e2 & 0x1F00u) != ((K_LIT << 10) | (1u << 8))) continue;
        const int l2 = (int)(e2 & 63u);
        if (l1 + l2 > LIT_TB) continue;
        lit[i] = (K_LIT << 10) | (2u << 8) | (u32)(l1 + l2) | (e & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8);
    }
}

inline const char* build_block_tables(const u8* litlen_lens, int nlit, const u8* dist_lens, int ndist, Tables& t) {
    if (litlen_lens[256] == 0) return "zlib stream is corrupt (no end-of-block code)";
    const char* e = build_table(litlen_lens, nlit, LIT_TB, t.lit, LIT_ENTRIES, true, make_litlen);
    if (e) return e;
    pair_literals(t.lit);
    return build_table(dist_lens, ndist, DIST_TB, t.dist, DIST_ENTRIES, true, make_dist);
}

struct FixedTables {
    Tables t;
    FixedTables() {
        u8 ll[288], dl[32];
        for (int i = 0; i < 288; ++i) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = 0; i < 32; ++i) dl[i] = 5;
        build_table(ll, 288, LIT_TB, t.lit, LIT_ENTRIES, false, make_litlen);
        pair_literals(t.lit);
        build_table(dl, 32, DIST_TB, t.dist, DIST_ENTRIES, true, make_dist);
    }
};

// ---------------------------------------------------------------------------------------------------------------- bit reader
struct Bits {
    const u8 *in, *end;
    u64 bb = 0;          // bit buffer, next bit = bit 0; bits at and above `cnt` are either zero or the stream's own next bits
    int cnt = 0;

    // at least 8 readable bytes at `in`: brings cnt to 56..63
    inline void refill_fast() {
        u64 w;
        memcpy(&w, in, 8);
        bb |= w << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
    }
    inline void refill_safe() {
        while (cnt <= 56 && in < end) {
            bb |= (u64)*in++ << cnt;
            cnt += 8;
        }
    }
    inline bool need(int n) {
        if (cnt >= n) return true;
        refill_safe();
        return cnt >= n;
    }
    inline u32 peek(int n) const { return (u32)(bb & ((1ull << n) - 1)); }
    inline void drop(int n) {
        bb >>= n;
        cnt -= n;
    }
    // to the next byte boundary; the whole bytes still in the buffer go back to the input
    inline void align() {
        drop(cnt & 7);
        in -= cnt >> 3;
        bb = 0;
        cnt = 0;
    }
};

struct Decoder {
    Tables dyn;

    // zlib stream in[0..n) -> out[0..cap); *produced = bytes written.  after_block(bytes so far) runs after every DEFLATE block (the PNG
    // reader unfilters the finished scanlines while they are still in cache).
    template <class AfterBlock>
    const char* zlib_inflate(const u8* in, size_t n, u8* out, size_t cap, size_t* produced, AfterBlock after_block) {
        *produced = 0;
        if (n < 2) return "truncated zlib stream (header)";
        if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0) return "zlib stream is corrupt (bad header)";
        if (in[1] & 0x20) return "zlib stream is corrupt (preset dictionary)";
        Bits b;
        b.in = in + 2;
        b.end = in + n;
        u8* op = out;
        u8* const oend = out + cap;
        static const FixedTables fixed;
        for (bool last = false; !last;) {
            if (!b.need(3)) return "truncated zlib stream (block header)";
            last = b.peek(1);
            const u32 type = (b.peek(3) >> 1);
            b.drop(3);
            const char* e = nullptr;
            if (type == 0) e = stored(b, op, oend);
            else if (type == 1) e = codes(b, fixed.t, out, op, oend);
            else if (type == 2) {
                e = dynamic_header(b);
                if (!e) e = codes(b, dyn, out, op, oend);
            } else e = "zlib stream is corrupt (block type 3)";
            if (e) return e;
            after_block((size_t)(op - out));
        }
        b.align();
        if (b.end - b.in < 4) return "truncated zlib stream (checksum)";
        const u32 want = ((u32)b.in[0] << 24) | ((u32)b.in[1] << 16) | ((u32)b.in[2] << 8) | b.in[3];
        *produced = (size_t)(op - out);
        if (adler32(out, *produced) != want) return "zlib stream is corrupt (Adler-32 mismatch)";
        if (b.end - b.in != 4) return "data after the end of the zlib stream";
        return nullptr;
    }

   private:
    static const char* stored(Bits& b, u8*& op, u8* oend) {
        b.align();
        if (b.end - b.in < 4) return "truncated zlib stream (stored block)";
        const u32 len = b.in[0] | ((u32)b.in[1] << 8), nlen = b.in[2] | ((u32)b.in[3] << 8);
        if ((len ^ nlen) != 0xFFFFu) return "zlib stream is corrupt (stored block length)";
        b.in += 4;
        if ((size_t)(b.end - b.in) < len) return "truncated zlib stream (stored block)";
        if ((size_t)(oend - op) < len) return "image data larger than the header says";
        memcpy(op, b.in, len);
        op += len;
        b.in += len;
        return nullptr;
    }

    const char* dynamic_header(Bits& b) {
        if (!b.need(14)) return "truncated zlib stream (dynamic block header)";
        const int nlit = (int)b.peek(5) + 257;
        b.drop(5);
        const int ndist = (int)b.peek(5) + 1;
        b.drop(5);
        const int nclen = (int)b.peek(4) + 4;
        b.drop(4);
        if (nlit > 286 || ndist > 30) return "zlib stream is corrupt (too many length or distance symbols)";
        static const u8 ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        u8 cl[19] = {0};
        for (int i = 0; i < nclen; ++i) {
            if (!b.need(3)) return "truncated zlib stream (code lengths)";
            cl[ORDER[i]] = (u8)b.peek(3);
            b.drop(3);
        }
        u32 ct[128];
        const char* e = build_table(cl, 19, 7, ct, 128, false, [](int s) { return (u32)s << 16; });
        if (e) return e;
        u8 lens[286 + 30];
        for (int i = 0; i < nlit + ndist;) {
            b.refill_safe();
            const u32 en = ct[b.peek(7)];
            const int l = (int)(en & 63u);
            if (kind_of(en) == K_BAD) return "zlib stream is corrupt (bad code length code)";
            if (l > b.cnt) return "truncated zlib stream (code lengths)";
            b.drop(l);
            const int s = (int)(en >> 16);
            if (s < 16) {
                lens[i++] = (u8)s;
                continue;
            }
            const int xb = s == 16 ? 2 : s == 17 ? 3 : 7;
            if (!b.need(xb)) return "truncated zlib stream (code lengths)";
            const int rep = (int)b.peek(xb) + (s == 18 ? 11 : 3);
            b.drop(xb);
            u8 v = 0;
            if (s == 16) {
                if (i == 0) return "zlib stream is corrupt (repeat without a previous length)";
                v = lens[i - 1];
            }
            if (i + rep > nlit + ndist) return "zlib stream is corrupt (code length repeat overruns)";
            memset(lens + i, v, (size_t)rep);
            i += rep;
        }
        return build_block_tables(lens, nlit, lens + nlit, ndist, dyn);
    }

    static inline void copy_match(u8* op, u32 dist, u32 len) {          // may write up to 15 bytes past op + len
        const u8* src = op - dist;
        if (dist >= 8) {
            u8* const stop = op + len;
            do {
                u64 w;
                memcpy(&w, src, 8);
                memcpy(op, &w, 8);
                src += 8;
                op += 8;
                memcpy(&w, src, 8);
                memcpy(op, &w, 8);
                src += 8;
                op += 8;
            } while (op < stop);
        } else if (dist == 1) {
            memset(op, *src, len);
        } else {
            // period 2..7 (pixel-sized repeats): the first multiple of the period >= 8 bytewise, the rest from that distance a word at a time
            u32 m = dist;
            while (m < 8) m += dist;
            const u32 head = len < m ? len : m;
            for (u32 i = 0; i < head; ++i) op[i] = src[i];
            if (len > head) {
                u8* d = op + head;
                const u8* s = d - m;
                u8* const stop = op + len;
                do {
                    u64 w;
                    memcpy(&w, s, 8);
                    memcpy(d, &w, 8);
                    s += 8;
                    d += 8;
                } while (d < stop);
            }
        }
    }

    // The Huffman-coded body of one block.  Fast loop while >= 16 input bytes (two refills of 8 bytes, the second up to 7 bytes on) and >= FAST_OUT output bytes are left (no bound checks:
    // one round consumes <= 56 bits and yields <= 6 + 258 (+ 15 of copy overshoot) bytes), then a checked loop for the ends.
    static const char* codes(Bits& bits, const Tables& t, const u8* out, u8*& op_ref, u8* oend) {
        constexpr u32 LMASK = (1u << LIT_TB) - 1, DMASK = (1u << DIST_TB) - 1;
        constexpr ptrdiff_t FAST_OUT = 6 + 258 + 16;
        u8* op = op_ref;
        const u32* const lit = t.lit;
        const u32* const dst = t.dist;
        {
            // the bit reader lives in locals here: the byte stores to `op` may alias anything a reference points to, and a reload of
            // the bit buffer after every store would sit in the lookup -> shift -> lookup dependency chain
            const u8* in = bits.in;
            u64 bb = bits.bb;
            u32 cnt = (u32)bits.cnt;
            const char* err = nullptr;
            bool eob = false;
#define LM_REFILL()                          \
    do {                                     \
        u64 w_;                              \
        memcpy(&w_, in, 8);                  \
        bb |= w_ << cnt;                     \
        in += (63 - cnt) >> 3;               \
        cnt |= 56;                           \
    } while (0)
#define LM_DROP(n_)          \
    do {                     \
        const u32 k_ = (n_); \
        bb >>= k_;           \
        cnt -= k_;           \
    } while (0)
            while (bits.end - in >= 16 && oend - op >= FAST_OUT) {
                LM_REFILL();
                u32 e = lit[bb & LMASK];
                // up to three literal entries (<= 15 bits each, sub-table codes included) per refill
                int rounds = 3;
                for (;;) {
                    if ((e & 0x1C00u) == (K_LIT << 10)) {
                        const u16 two = (u16)(e >> 16);
                        memcpy(op, &two, 2);
                        op += (e >> 8) & 3u;
                        LM_DROP(e & 63u);
                        if (--rounds == 0) break;
                        e = lit[bb & LMASK];
                        continue;
                    }
                    if (kind_of(e) != K_SUB) break;
                    e = lit[(e >> 16) + ((bb >> LIT_TB) & ((1u << sub_width(e)) - 1))];          // (never another K_SUB)
                }
                if (rounds == 0) continue;
                const u32 k = kind_of(e);
                if (k == K_LEN) {
                    LM_REFILL();                                   // (the entry's bits are still there: a refill only adds above them)
                    const u32 xb = (e >> 25) & 7u, tot = e & 63u;
                    const u32 len = ((e >> 16) & 511u) + (u32)((bb >> (tot - xb)) & ((1u << xb) - 1));
                    LM_DROP(tot);
                    u32 d = dst[bb & DMASK];
                    if (kind_of(d) == K_SUB) d = dst[(d >> 16) + ((bb >> DIST_TB) & ((1u << sub_width(d)) - 1))];
                    if (kind_of(d) != K_DIST) {
                        err = "zlib stream is corrupt (bad distance code)";
                        break;
                    }
                    const u32 dxb = dist_extra(d), dtot = d & 63u;
                    const u32 dist = ((d >> 16) & 32767u) + (u32)((bb >> (dtot - dxb)) & ((1u << dxb) - 1));
                    LM_DROP(dtot);
                    if (dist > (size_t)(op - out)) {
                        err = "zlib stream is corrupt (distance before the start of the data)";
                        break;
                    }
                    copy_match(op, dist, len);
                    op += len;
                } else if (k == K_EOB) {
                    LM_DROP(e & 63u);
                    eob = true;
                    break;
                } else {
                    err = "zlib stream is corrupt (bad literal / length code)";
                    break;
                }
            }
#undef LM_REFILL
#undef LM_DROP
            bits.in = in;
            bits.bb = bb;
            bits.cnt = (int)cnt;
            op_ref = op;
            if (err) return err;
            if (eob) return nullptr;
        }
        Bits& b = bits;
        // checked loop
        for (;;) {
            b.refill_safe();
            u32 e = lit[b.bb & LMASK];
            if (kind_of(e) == K_SUB) e = lit[(e >> 16) + ((b.bb >> LIT_TB) & ((1u << sub_width(e)) - 1))];
            op_ref = op;
            if ((int)(e & 63u) > b.cnt) return "truncated zlib stream";
            const u32 k = kind_of(e);
            if (k == K_LIT) {
                const u32 cntl = (e >> 8) & 3u;
                if ((size_t)(oend - op) < cntl) return "image data larger than the header says";
                op[0] = (u8)(e >> 16);
                if (cntl == 2) op[1] = (u8)(e >> 24);
                op += cntl;
                b.drop((int)(e & 63u));
            } else if (k == K_LEN) {
                const u32 xb = (e >> 25) & 7u, tot = e & 63u;
                const u32 len = ((e >> 16) & 511u) + (u32)((b.bb >> (tot - xb)) & ((1u << xb) - 1));
                b.drop((int)tot);
                b.refill_safe();
                u32 d = dst[b.bb & DMASK];
                if (kind_of(d) == K_SUB) d = dst[(d >> 16) + ((b.bb >> DIST_TB) & ((1u << sub_width(d)) - 1))];
                if (kind_of(d) != K_DIST) return "zlib stream is corrupt (bad distance code)";
                const u32 dxb = dist_extra(d), dtot = d & 63u;
                if ((int)dtot > b.cnt) return "truncated zlib stream";
                const u32 dist = ((d >> 16) & 32767u) + (u32)((b.bb >> (dtot - dxb)) & ((1u << dxb) - 1));
                b.drop((int)dtot);
                if (dist > (size_t)(op - out)) return "zlib stream is corrupt (distance before the start of the data)";
                if ((size_t)(oend - op) < len) return "image data larger than the header says";
                for (u32 i = 0; i < len; ++i) op[i] = op[(ptrdiff_t)i - (ptrdiff_t)dist];
                op += len;
            } else if (k == K_EOB) {
                b.drop((int)(e & 63u));
                op_ref = op;
                return nullptr;
            } else {
                return "zlib stream is corrupt (bad literal / length code)";
            }
        }
    }
};

}  // namespace lm_inflate

// PNG tile reader: the reference's `load_img` (datasets/laserlane_proposals.py:85-98, laserlane.py:214-219) is
// `np.array(Image.open(path))` -> uint8 HWC; this is the same decode without PIL and without the GIL, so a batch of tiles is inflated on
// the host thread pool while the GPU works on the previous batch (at 220 tiles/s one GPU consumes ~5 cores of zlib inflate).
// Scope = what BEV tiles are: 8-bit, non-interlaced, greyscale / grey+alpha / RGB / RGBA.  Everything else (palette, 16-bit, Adam7) is
// refused with a message, never guessed.  Chunk CRCs and the zlib Adler checksum are verified: a damaged tile is an error, not noise.
// Host code only (no HIP calls).
#include "common.h"

#include <zlib.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
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

// Reverses the scanline filter `ft` of one row in place of `src` -> `cur` (`up` = previous output row or nullptr).
inline bool unfilter_row(int ft, const unsigned char* src, unsigned char* cur, const unsigned char* up, size_t stride, size_t bp) {
    // the first pixel has no left neighbour; the rest of the row runs without per-byte conditions
    switch (ft) {
        case 0: memcpy(cur, src, stride); return true;
        case 1:
            memcpy(cur, src, bp);
            for (size_t i = bp; i < stride; ++i) cur[i] = (unsigned char)(src[i] + cur[i - bp]);
            return true;
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
                for (size_t i = bp; i < stride; ++i) cur[i] = (unsigned char)(src[i] + ((cur[i - bp] + up[i]) >> 1));
            }
            return true;
        case 4:
            if (!up) {                                           // b = c = 0 -> predictor = a
                memcpy(cur, src, bp);
                for (size_t i = bp; i < stride; ++i) cur[i] = (unsigned char)(src[i] + cur[i - bp]);
            } else {
                for (size_t i = 0; i < bp; ++i) cur[i] = (unsigned char)(src[i] + up[i]);          // a = c = 0 -> predictor = b
                for (size_t i = bp; i < stride; ++i) cur[i] = (unsigned char)(src[i] + paeth(cur[i - bp], up[i], up[i - bp]));
            }
            return true;
        default: return false;
    }
}

// out: [H][W][channels]; returns nullptr on success.  The IDAT chunks are inflated as one stream straight from the file image, one
// scanline at a time into a row buffer that is unfiltered into `out`: no copy of the compressed data, no full-size filtered image.
const char* decode(const unsigned char* d, long n, const PngHeader& h, unsigned char* out) {
    const size_t bp = (size_t)h.channels, stride = (size_t)h.W * bp;
    std::vector<unsigned char> line(stride + 1);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return "zlib initialisation failed";
    zs.next_out = line.data();
    zs.avail_out = (uInt)line.size();
    int row = 0;
    bool stream_end = false, saw_idat = false, end = false;
    const char* err = nullptr;
    long pos = 8;
    while (!end && !err) {
        if (pos + 12 > n) { err = "truncated file (chunk header)"; break; }
        const unsigned len = be32(d + pos);
        const unsigned char* type = d + pos + 4;
        if ((long)len > n - pos - 12) { err = "truncated file (chunk data)"; break; }
        if ((unsigned)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(d + pos + 8 + len)) { err = "chunk CRC mismatch"; break; }
        if (memcmp(type, "IDAT", 4) == 0) {
            saw_idat = true;
            zs.next_in = const_cast<unsigned char*>(d + pos + 8);
            zs.avail_in = len;
            while (zs.avail_in > 0 && !err) {
                if (stream_end) { err = "data after the end of the zlib stream"; break; }
                unsigned char spill;
                if (row >= h.H) {                                // all rows are out: only the stream trailer may follow
                    zs.next_out = &spill;
                    zs.avail_out = 1;
                }
                const int zr = inflate(&zs, Z_NO_FLUSH);
                if (zr != Z_OK && zr != Z_STREAM_END) { err = zr == Z_DATA_ERROR ? "zlib stream is corrupt" : "zlib failure"; break; }
                if (row >= h.H) {
                    if (zs.avail_out == 0) { err = "image data larger than the header says"; break; }
                } else if (zs.avail_out == 0) {
                    unsigned char* cur = out + stride * (size_t)row;
                    if (!unfilter_row(line[0], line.data() + 1, cur, row ? cur - stride : nullptr, stride, bp)) { err = "bad scanline filter type"; break; }
                    ++row;
                    zs.next_out = line.data();
                    zs.avail_out = (uInt)line.size();
                }
                if (zr == Z_STREAM_END) {
                    stream_end = true;
                    if (zs.avail_in > 0) err = "data after the end of the zlib stream";
                }
            }
        } else if (memcmp(type, "IEND", 4) == 0) {
            end = true;
        } else if (!(type[0] & 0x20) && memcmp(type, "IHDR", 4) != 0 && memcmp(type, "PLTE", 4) != 0) {
            err = "unknown critical chunk";
        }
        pos += 12 + (long)len;
    }
    inflateEnd(&zs);
    if (err) return err;
    if (!saw_idat) return "no IDAT chunk";
    if (row != h.H) return "image data shorter than the header says";
    if (!stream_end) return "zlib stream is not terminated";
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
    e = decode(data, size, h, out);
    LM_REQUIRE(!e, "png_decode: %s", e);
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
        std::vector<unsigned char> buf;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            const char* e = read_file(paths[i], buf);
            PngHeader h{};
            if (!e) e = parse_header(buf.data(), (long)buf.size(), h);
            if (!e && (h.H != H || h.W != W || h.channels != C)) e = "geometry differs from the requested H x W x C";
            if (!e) e = decode(buf.data(), (long)buf.size(), h, out + per * (size_t)i);
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

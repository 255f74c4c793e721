// Per-tile polyline JSON writer: the text of the reference's
//   save_lane_seq_2d (baseline/utils/io_utils.py:58-93) = json.dump(list of {seq_len, seq, init_vertex, end_vertex}, indent=4)
// produced without Python's indented JSON encoder, which is pure Python and costs 13-30 ms per tile - more than the GPU needs for
// the tile.  Numbers are formatted exactly like CPython's float.__repr__ (shortest round-trip digits; exponent form iff the decimal
// point position is <= -4 or > 16; ".0" appended to integral values), so the files are byte-identical to the reference's.
// Host code only.
#include "common.h"

#include <charconv>
#include <cmath>
#include <cstring>
#include <string>

namespace {

// CPython: PyOS_double_to_string(x, 'r', 0, Py_DTSF_ADD_DOT_0, NULL)  (Python/pystrtod.c format_float_short)
void py_float_repr(double x, std::string& o) {
    if (std::isnan(x)) { o += "NaN"; return; }                 // json.dump spelling (allow_nan=True)
    if (std::isinf(x)) { o += x < 0 ? "-Infinity" : "Infinity"; return; }
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest round trip
    const char* p = buf;
    if (*p == '-') { o += '-'; ++p; }
    char digits[24];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int e10 = 0;
    {
        const char* q = p + 1;
        const bool neg = *q == '-';
        if (*q == '-' || *q == '+') ++q;
        for (; q < r.ptr; ++q) e10 = e10 * 10 + (*q - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;                                  // value = 0.d1d2... * 10^decpt
    if (decpt <= -4 || decpt > 16) {                            // exponent form: d[.ddd]e[+-]XX (at least two exponent digits)
        o += digits[0];
        if (nd > 1) {
            o += '.';
            o.append(digits + 1, (size_t)(nd - 1));
        }
        o += 'e';
        int e = decpt - 1;
        o += e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) o += eb[--ne];
    } else if (decpt <= 0) {                                    // 0.000ddd
        o += "0.";
        o.append((size_t)(-decpt), '0');
        o.append(digits, (size_t)nd);
    } else if (decpt >= nd) {                                   // ddd000.0
        o.append(digits, (size_t)nd);
        o.append((size_t)(decpt - nd), '0');
        o += ".0";
    } else {                                                    // dd.ddd
        o.append(digits, (size_t)decpt);
        o += '.';
        o.append(digits + decpt, (size_t)(nd - decpt));
    }
}

void vertex_list(const double* v, int n, int indent, std::string& o) {     // "[\n<indent+4>a,\n<indent+4>b\n<indent>]"
    o += "[\n";
    for (int i = 0; i < n; ++i) {
        o.append((size_t)indent + 4, ' ');
        py_float_repr(v[i], o);
        o += i + 1 < n ? ",\n" : "\n";
    }
    o.append((size_t)indent, ' ');
    o += ']';
}

void build(const double* lv, int n_lines, int row_size, int with_sem, std::string& o) {
    const int nv = with_sem ? 3 : 2;
    bool first = true;
    for (int l = 0; l < n_lines; ++l) {
        const double* lane = lv + (size_t)l * row_size * 3;
        int cnt = 0, first_i = -1, last_i = -1;
        for (int i = 0; i < row_size; ++i)
            if (lane[i * 3 + 1] > 0) {
                if (first_i < 0) first_i = i;
                last_i = i;
                ++cnt;
            }
        if (cnt < 2) continue;
        o += first ? "[\n    {\n" : ",\n    {\n";
        first = false;
        o += "        \"seq_len\": " + std::to_string(cnt) + ",\n        \"seq\": [\n";
        int k = 0;
        for (int i = 0; i < row_size; ++i)
            if (lane[i * 3 + 1] > 0) {
                o.append(12, ' ');
                vertex_list(lane + i * 3, nv, 12, o);
                o += ++k < cnt ? ",\n" : "\n";
            }
        o += "        ],\n        \"init_vertex\": ";
        vertex_list(lane + first_i * 3, nv, 8, o);
        o += ",\n        \"end_vertex\": ";
        vertex_list(lane + last_i * 3, nv, 8, o);
        o += "\n    }";
    }
    o += first ? "[]" : "\n]";
}

// records of the 3-D / merged polyline files: keys in the order the reference builds its dicts (coor_img2pc.py:205-212)
void build_seqs(const double* seqs, const int* lens, int L, int Vmax, int D, std::string& o) {
    if (L == 0) { o += "[]"; return; }
    for (int l = 0; l < L; ++l) {
        const double* sq = seqs + (size_t)l * Vmax * D;
        const int n = lens[l];
        o += l ? ",\n    {\n" : "[\n    {\n";
        o += "        \"seq\": [";
        if (n == 0) {
            o += "]";
        } else {
            o += "\n";
            for (int i = 0; i < n; ++i) {
                o.append(12, ' ');
                vertex_list(sq + (size_t)i * D, D, 12, o);
                o += i + 1 < n ? ",\n" : "\n";
            }
            o += "        ]";
        }
        o += ",\n        \"seq_len\": " + std::to_string(n) + ",\n        \"init_vertex\": ";
        vertex_list(sq, D, 8, o);
        o += ",\n        \"end_vertex\": ";
        vertex_list(sq + (size_t)(n > 0 ? n - 1 : 0) * D, D, 8, o);
        o += "\n    }";
    }
    o += "\n]";
}

}  // namespace

// lane_vertexes: [n_lines][row_size][3] doubles = (row, col, semantic) per vertex, the layout of pack_lane_vertices; a vertex exists iff
// col > 0, lines with fewer than 2 vertices are dropped (io_utils.py:60-71).  Writes the JSON text into out (cap bytes incl. the
// terminating NUL) and returns its length; if cap is too small nothing is written and the needed length is returned (call twice).
LM_API long lm_lane_json_text(const double* lane_vertexes, int n_lines, int row_size, int with_pervertex_semantics, char* out, long cap) {
    if (!lane_vertexes || n_lines < 0 || row_size < 0) {
        lm_set_error("lane_json_text: bad arguments");
        return -1;
    }
    std::string s;
    s.reserve((size_t)n_lines * row_size * 96 + 64);
    build(lane_vertexes, n_lines, row_size, with_pervertex_semantics, s);
    if (out && cap > (long)s.size()) memcpy(out, s.c_str(), s.size() + 1);
    return (long)s.size();
}

// Same text straight into a file (what save_lane_seq_2d does for a .json path).
LM_API int lm_lane_json_write(const double* lane_vertexes, int n_lines, int row_size, int with_pervertex_semantics, const char* path) {
    LM_REQUIRE(lane_vertexes && path && n_lines >= 0 && row_size >= 0, "lane_json_write: bad arguments");
    std::string s;
    s.reserve((size_t)n_lines * row_size * 96 + 64);
    build(lane_vertexes, n_lines, row_size, with_pervertex_semantics, s);
    FILE* f = fopen(path, "wb");
    LM_REQUIRE(f, "lane_json_write: cannot open %s for writing", path);
    const size_t w = fwrite(s.data(), 1, s.size(), f);
    const int rc = fclose(f);
    LM_REQUIRE(w == s.size() && rc == 0, "lane_json_write: short write to %s", path);
    return LM_OK;
}

// 3-D (or any D >= 1) polylines: seqs [L][Vmax][D] doubles, lens [L] (1 <= lens[l] <= Vmax) -> the text of
// json.dump([{"seq": .., "seq_len": n, "init_vertex": seq[0], "end_vertex": seq[n-1]}, ..], indent=4) (save_seqs_json,
// baseline/utils/io_utils.py:11-15, as called from coor_img2pc.py:205-214), written to `path`.
LM_API int lm_seqs_json_write(const double* seqs, const int* lens, int L, int Vmax, int D, const char* path) {
    LM_REQUIRE(path && L >= 0 && Vmax >= 0 && D >= 1 && (L == 0 || (seqs && lens)), "seqs_json_write: bad arguments");
    for (int l = 0; l < L; ++l) LM_REQUIRE(lens[l] >= 1 && lens[l] <= Vmax, "seqs_json_write: lens[%d]=%d out of range", l, lens[l]);
    std::string s;
    s.reserve((size_t)L * Vmax * D * 40 + 64);
    build_seqs(seqs, lens, L, Vmax, D, s);
    FILE* f = fopen(path, "wb");
    LM_REQUIRE(f, "seqs_json_write: cannot open %s for writing", path);
    const size_t w = fwrite(s.data(), 1, s.size(), f);
    const int rc = fclose(f);
    LM_REQUIRE(w == s.size() && rc == 0, "seqs_json_write: short write to %s", path);
    return LM_OK;
}

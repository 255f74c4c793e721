// Cross-tile merge of LAS-frame 3-D polylines into map-level lines - host C++ behind the C-ABI.
//
// Replaces baseline/utils/merge_lines.py: `merge_lines` :166-291 (the driver), `merge_2_seqs` :67-104, `merge_2_reversed_seqs`
// :106-132, `downsample_seqs` :133-153 and the distance / heading helpers :17-65, :157-164.  The step after the all-gather: tiles
// arrive in sorted file-name order, every polyline of the new tile is matched THROUGH ITS FIRST VERTEX against the live tails of
// the map's open lines, woven into one (same heading), stitched onto one (opposite heading) or opens a new line; lines the tile did
// not touch are retired.
//
// Design: a streaming merger object (create -> add_tile per tile -> finish -> copy results) that owns the open lines, so a caller
// can feed tiles as they leave the GPU pipeline instead of re-reading JSON files; every line is a std::vector of points; the
// principal axis of the xy scatter matrix is computed in closed form (2x2 symmetric eigen problem) instead of a LAPACK call.
// Behaviours of the reference that decide WHICH lines exist are kept, because they are the specification here:
//   * the insertion loop of the same-heading weave indexes the GROWING base with positions found before the first insertion (:84-91);
//   * the retirement sweep erases from the list it walks, so the entry behind every retired line is skipped that round (:270-281);
//   * a weave whose candidate starts before every base vertex yet overlaps its end fails in the reference (IndexError at :104):
//     reported as LM_ERR_ARG here.
#include "common.h"

#include <cmath>
#include <vector>

namespace {

struct P3 {
    double x, y, z;
};
typedef std::vector<P3> Line;

constexpr double EPS = 1e-6;

inline P3 sub(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// unit vector first -> last vertex in the xy plane (:58-64); numpy: d / (sqrt(sum(d^2)) + EPS) with d.z = 0
P3 heading_easy(const P3* s, size_t n) {
    P3 d = sub(s[n - 1], s[0]);
    d.z = 0;
    const double len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z) + EPS;
    return {d.x / len, d.y / len, d.z / len};
}

// heading from the last 5 vertices (:157-164)
P3 heading_local(const Line& s) { return s.size() > 5 ? heading_easy(s.data() + s.size() - 5, 5) : heading_easy(s.data(), s.size()); }

// dominant eigenvector of the xy scatter matrix of the centred points (:46-56), oriented along first -> last (:69-72, :108-111)
P3 principal_axis(const P3* s, size_t n) {
    double mx = 0, my = 0;
    for (size_t i = 0; i < n; ++i) {
        mx += s[i].x;
        my += s[i].y;
    }
    mx /= (double)n;
    my /= (double)n;
    double a = 0, b = 0, c = 0;      // [[a, b], [b, c]]
    for (size_t i = 0; i < n; ++i) {
        const double dx = s[i].x - mx, dy = s[i].y - my;
        a += dx * dx;
        b += dx * dy;
        c += dy * dy;
    }
    P3 v;
    if (b == 0.0) {
        v = a >= c ? P3{1, 0, 0} : P3{0, 1, 0};
    } else {
        const double h = 0.5 * (a - c), lam = 0.5 * (a + c) + std::sqrt(h * h + b * b);
        // (b, lam - a) and (lam - c, b) are both eigenvectors; take the better conditioned one
        double vx, vy;
        if (a >= c) {
            vx = lam - c;
            vy = b;
        } else {
            vx = b;
            vy = lam - a;
        }
        const double nrm = std::sqrt(vx * vx + vy * vy);
        v = {vx / nrm, vy / nrm, 0};
    }
    const P3 e = heading_easy(s, n);
    if (dot(v, e) < 0) v = {-v.x, -v.y, -v.z};
    return v;
}

// (distance, index) of the vertex of s[0..n) nearest to pt in xy; proj: the distance is measured perpendicular to the
// sequence's first -> last direction through that vertex (:17-31).  First minimum wins (numpy argmin).
double nearest(const P3& pt, const P3* s, size_t n, bool proj, size_t* idx = nullptr) {
    size_t k = 0;
    double best = 0;
    for (size_t i = 0; i < n; ++i) {
        const double dx = s[i].x - pt.x, dy = s[i].y - pt.y;
        const double d2 = dx * dx + dy * dy;
        if (i == 0 || d2 < best) {
            best = d2;
            k = i;
        }
    }
    if (idx) *idx = k;
    if (!proj) return std::sqrt(best);
    const P3 u = heading_easy(s, n), w = sub(pt, s[k]);
    const double cx = u.y * w.z - u.z * w.y, cy = u.z * w.x - u.x * w.z, cz = u.x * w.y - u.y * w.x;
    return std::sqrt(cx * cx + cy * cy + cz * cz);
}

// Weave `fresh` (same heading) into the tail `base` by position along the base's principal axis (:67-104).  Returns the index in
// the base where the overlap starts, or -1 for the reference's IndexError case.
long weave(Line& base, const Line& fresh) {
    const P3 axis = principal_axis(base.data(), base.size());
    std::vector<double> tb(base.size()), tn(fresh.size());
    for (size_t i = 0; i < base.size(); ++i) tb[i] = dot(base[i], axis);
    for (size_t j = 0; j < fresh.size(); ++j) tn[j] = dot(fresh[j], axis);
    std::vector<size_t> over_b, over_n;                 // positions in the ORIGINAL arrays
    const double last_b = tb.back();
    for (size_t i = 0; i < tb.size(); ++i)
        if (tb[i] > tn[0]) over_b.push_back(i);
    for (size_t j = 0; j < tn.size(); ++j)
        if (tn[j] < last_b) over_n.push_back(j);
    for (size_t j : over_n)
        for (size_t i : over_b)                          // stale positions applied to the growing base: the reference's behaviour
            if (tn[j] < tb[i]) {
                base.insert(base.begin() + (long)i, fresh[j]);
                tb.insert(tb.begin() + (long)i, tn[j]);
                break;
            }
    if (over_n.empty()) {
        const long start = (long)base.size();
        base.insert(base.end(), fresh.begin(), fresh.end());
        return start;
    }
    base.insert(base.end(), fresh.begin() + (long)over_n.back() + 1, fresh.end());
    return over_b.empty() ? -1 : (long)over_b[0];
}

// `fresh` runs against the base: append what lies beyond the base's end (walking it backwards), prepend what lies before its
// start (:106-132)
void stitch_reversed(Line& base, const Line& fresh) {
    const P3 axis = principal_axis(base.data(), base.size());
    const double t0 = dot(base.front(), axis), t1 = dot(base.back(), axis);
    std::vector<size_t> ahead, behind;
    for (size_t j = 0; j < fresh.size(); ++j) {
        const double t = dot(fresh[j], axis);
        if (t > t1) ahead.push_back(j);
        if (t < t0) behind.push_back(j);
    }
    for (size_t k = ahead.size(); k-- > 0;) base.push_back(fresh[ahead[k]]);
    for (size_t j : behind) base.insert(base.begin(), fresh[j]);
}

struct Open {
    Line pts;
    size_t roi = 0;      // where the part of the line that may still overlap new tiles starts
    P3 heading{0, 0, 0};
    bool touched = false;
};

struct Merger {
    std::vector<Open> open;
    std::vector<Line> done;
    bool first = true;
    bool finished = false;

    void retire(Open& o) {
        if (o.pts.size() >= 3) done.push_back(std::move(o.pts));
    }

    int add_tile(const std::vector<Line>& lines) {
        if (first) {                                      // the first tile only seeds the open set (:177-191)
            first = false;
            for (const Line& l : lines) {
                Open o;
                o.pts = l;
                o.heading = heading_local(l);
                open.push_back(std::move(o));
            }
            return LM_OK;
        }
        for (Open& o : open) o.touched = false;
        std::vector<P3> heads(lines.size());
        for (size_t t = 0; t < lines.size(); ++t) heads[t] = heading_local(lines[t]);
        for (size_t t = 0; t < lines.size(); ++t) {
            const Line& cand = lines[t];
            long best = -1;
            double best_d = 10;
            for (size_t a = 0; a < open.size(); ++a) {
                const Open& o = open[a];
                const double d = nearest(cand[0], o.pts.data() + o.roi, o.pts.size() - o.roi, true);
                if (d < best_d) {
                    best = (long)a;
                    best_d = d;
                }
            }
            if (best_d < 0.5) {
                Open& o = open[(size_t)best];
                const double cosv = dot(heads[t], o.heading);
                const double back_d = nearest(o.pts.back(), cand.data(), cand.size(), true);
                if (back_d < 0.5 && cosv > 0.7) {
                    Line tail(o.pts.begin() + (long)o.roi, o.pts.end());
                    const long start = weave(tail, cand);
                    if (start < 0) {
                        lm_set_error("merge_lines: candidate starts before every vertex of the open tail yet overlaps its end "
                                     "(the reference raises IndexError at merge_lines.py:104)");
                        return LM_ERR_ARG;
                    }
                    o.pts.resize(o.roi);
                    o.pts.insert(o.pts.end(), tail.begin(), tail.end());
                    o.roi += (size_t)start;
                    o.touched = true;
                    o.heading = heading_local(o.pts);
                    continue;
                }
                if (back_d < 0.5 && cosv < -0.7) {
                    stitch_reversed(o.pts, cand);
                    o.touched = true;
                    o.heading = heading_local(o.pts);
                    continue;
                }
            }
            Open o;                                        // a new line starts here
            o.pts = cand;
            o.heading = heads[t];
            o.touched = true;
            open.push_back(std::move(o));
        }
        // retire what this tile did not touch; the sweep advances its index after an erase, like the reference's pop-while-enumerating
        for (size_t i = 0; i < open.size(); ++i)
            if (!open[i].touched) {
                retire(open[i]);
                open.erase(open.begin() + (long)i);
            }
        return LM_OK;
    }

    void finish() {
        if (finished) return;
        finished = true;
        for (Open& o : open) retire(o);
        open.clear();
    }
};

bool read_lines(const double* pts, const int* lens, int n, std::vector<Line>& out) {
    long at = 0;
    out.resize((size_t)(n > 0 ? n : 0));
    for (int i = 0; i < n; ++i) {
        if (lens[i] < 2) return false;
        out[(size_t)i].resize((size_t)lens[i]);
        for (int k = 0; k < lens[i]; ++k, ++at) out[(size_t)i][(size_t)k] = {pts[3 * at], pts[3 * at + 1], pts[3 * at + 2]};
    }
    return true;
}

}  // namespace

LM_API void* lm_merge_create(void) { return new Merger(); }

LM_API void lm_merge_destroy(void* h) { delete static_cast<Merger*>(h); }

// One tile: n polylines, points concatenated [sum lens][3] (x, y, z in the LAS frame), every line with >= 2 vertices.  n = 0 is a
// tile without usable lines (load_lane_seq returns nothing for files with fewer than two lines): it still retires untouched lines.
LM_API int lm_merge_add_tile(void* h, const double* points, const int* lens, int n) {
    LM_REQUIRE(h && n >= 0 && (n == 0 || (points && lens)), "merge_add_tile: bad arguments");
    Merger* m = static_cast<Merger*>(h);
    LM_REQUIRE(!m->finished, "merge_add_tile: the merger was already finished");
    std::vector<Line> lines;
    LM_REQUIRE(read_lines(points, lens, n, lines), "merge_add_tile: every polyline needs at least 2 vertices");
    return m->add_tile(lines);
}

// Close the map: the lines still open are retired in their order.  Returns the number of merged lines; *total_points their vertex count.
LM_API long lm_merge_finish(void* h, long* total_points) {
    if (!h) return -1;
    Merger* m = static_cast<Merger*>(h);
    m->finish();
    long tot = 0;
    for (const Line& l : m->done) tot += (long)l.size();
    if (total_points) *total_points = tot;
    return (long)m->done.size();
}

LM_API int lm_merge_result(void* h, double* points, int* lens) {
    LM_REQUIRE(h && points && lens, "merge_result: null pointer");
    Merger* m = static_cast<Merger*>(h);
    LM_REQUIRE(m->finished, "merge_result: call lm_merge_finish first");
    long at = 0;
    for (size_t i = 0; i < m->done.size(); ++i) {
        lens[i] = (int)m->done[i].size();
        for (const P3& p : m->done[i]) {
            points[3 * at] = p.x;
            points[3 * at + 1] = p.y;
            points[3 * at + 2] = p.z;
            ++at;
        }
    }
    return LM_OK;
}

// Keep a vertex whenever more than dist_min metres (xy) accumulated since the last kept one (merge_lines.py:133-153); out holds
// at most n + 1 points.  Returns the number kept.
LM_API int lm_downsample_seq(const double* seq, int n, double dist_min, double* out) {
    if (!seq || !out || n < 1) return 0;
    int kept = 0;
    auto keep = [&](int i) {
        out[3 * kept] = seq[3 * i];
        out[3 * kept + 1] = seq[3 * i + 1];
        out[3 * kept + 2] = seq[3 * i + 2];
        ++kept;
    };
    auto step = [&](int i) {      // xy distance to the next vertex; the last vertex "steps" to itself
        const int j = i + 1 < n ? i + 1 : i;
        const double dx = seq[3 * j] - seq[3 * i], dy = seq[3 * j + 1] - seq[3 * i + 1];
        return std::sqrt(dx * dx + dy * dy + 0.0);
    };
    keep(0);
    double acc = 0;
    for (int i = 0; i < n; ++i) {
        acc += step(i);
        if (acc > dist_min) {
            keep(i);
            acc = 0;
        } else if (i == n - 1) {
            const double prev = step(i > 0 ? i - 1 : n - 1);      // numpy's step[i - 1] wraps to the last element for i = 0
            if (prev < 0.05 || i == 0 || acc < 0.05) continue;
            keep(i);
        }
    }
    return kept;
}

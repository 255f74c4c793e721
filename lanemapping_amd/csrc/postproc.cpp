// Host-side (C++) tail of the hot path: endpoint clustering and per-tile polyline assembly.
//
// These stages are tiny, branchy and strictly sequential (72 x 144 doubles per tile), so they run on
// host threads overlapped with the GPU work of the next batch instead of on the GPU.
// They replace ~2.3 s/tile of Python loops in the reference:
//   lm_endp_cluster      <- ColumnProposal2.get_exist_coor_endp_dict :661-688 + cluster_select_topK_pts :903-924
//                           (sklearn DBSCAN(eps=20, min_samples=1) == connected components of the <=20 px graph;
//                            NearestNeighbors(centroid) == arg-min distance, ties -> lowest index)
//   lm_polyline_assemble <- ColumnProposal2.get_lane_map_numpy_with_label :805-861 and
//                           baseline/utils/polyline_utils.py :7-45, :57-164, :167-220, :222-387, :448-608,
//                           get_pred_semantic_lane_coordinates :1091-1115
// Reference quirks are reproduced on purpose (C1 row-0-only occupancy filter, C3 clamp of
// -1 to 0, C4 dropped proposals, C16 stable left-to-right order).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

constexpr int IMG = 1152;
constexpr int BUFF_W = 6;
constexpr int BUFF_D = 24;

struct Lines {
    int n, r;                 // lines, rows
    std::vector<double> v;    // [n][r]
    Lines(int n_, int r_, double fill) : n(n_), r(r_), v((size_t)n_ * r_, fill) {}
    double* row(int i) { return v.data() + (size_t)i * r; }
    const double* row(int i) const { return v.data() + (size_t)i * r; }
};

// polyline_utils.py:167-178 with a stable sort (ties -> lower line index)
Lines order_left_to_right(const Lines& L) {
    std::vector<double> key(L.n, (double)IMG);
    for (int i = 0; i < L.n; ++i) {
        const double* a = L.row(i);
        for (int h = 0; h < L.r; ++h)
            if (a[h] >= 0) {
                key[i] = a[h];
                break;
            }
    }
    std::vector<int> idx(L.n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
    Lines out(L.n, L.r, 0.0);
    for (int i = 0; i < L.n; ++i) std::memcpy(out.row(i), L.row(idx[i]), sizeof(double) * L.r);
    return out;
}

// polyline_utils.py:180-198
void fill_gaps(Lines& L) {
    std::vector<int> pos;
    for (int i = 0; i < L.n; ++i) {
        double* a = L.row(i);
        pos.clear();
        for (int h = 0; h < L.r; ++h)
            if (a[h] > 1e-4) pos.push_back(h);
        if (pos.size() > 1) {
            int cur = -1;
            const int np = (int)pos.size();
            for (int v = pos.front(); v < pos.back(); ++v) {
                if (a[v] < 1e-4) {
                    const int pa = pos[(cur + np) % np], pb = pos[cur + 1];   // cur == -1 cannot reach here
                    const double ratio = (1.0 * v - pa) / (double)(pb - pa);
                    a[v] = (1 - ratio) * a[pa] + a[pb] * ratio;
                } else {
                    ++cur;
                }
            }
        }
    }
}

// polyline_utils.py:222-387 (+ :200-220 for the row-0 occupancy filter)
Lines trace_lines(const Lines& C, const float* seg_rows /*[r][IMG]*/) {
    const int n = C.n, R = C.r;
    Lines S = order_left_to_right(C);
    Lines total(n, R, -1.0);
    std::vector<double> have(n, 0.0);
    std::vector<unsigned char> flag((size_t)R * IMG, 0);
    long nflag = 0;
    for (int i = 0; i < n; ++i)
        for (int h = 0; h < R; ++h) {
            const double c = C.row(i)[h];
            if (c > 0) {
                unsigned char& f = flag[(size_t)h * IMG + (int)c];
                if (!f) {
                    f = 1;
                    ++nflag;
                }
            }
        }
    if (seg_rows && R > 0) {   // occupancy_filter: only image row 0 of the grid is ever filtered (quirk C1)
        unsigned char* f0 = flag.data();
        const int half = 4;
        for (int c = half; c < IMG - half; ++c) {
            int cnt = 0;
            for (int j = 0; j < 2 * half; ++j) cnt += f0[c - half + j];
            if (cnt > 1) {
                int best = -1;
                for (int j = 0; j < 2 * half; ++j)
                    if (f0[c - half + j]) {
                        if (best < 0 || seg_rows[c - half + j] > seg_rows[c - half + best]) best = j;
                    }
                for (int j = 0; j < 2 * half; ++j) f0[c - half + j] = 0;
                f0[c - half + best] = 1;
                nflag -= (cnt - 1);
            }
        }
    }
    auto fl = [&](int h, double col) -> unsigned char& { return flag[(size_t)h * IMG + (int)col]; };
    std::vector<double> plen(n);
    while (nflag > 2 && *std::min_element(have.begin(), have.end()) < 2) {
        Lines piece(n, R, -1.0);
        std::fill(plen.begin(), plen.end(), 0.0);
        for (int i = 0; i < n; ++i) {
            bool started = false;
            int last_h = 0, h = 0, follow = i, step = 1;
            double last_c = 0.0, cur = 0.0;
            const double* Si = S.row(i);
            while (h < R) {
                if (started && (h - last_h > BUFF_D)) break;
                if (!started) {
                    if (Si[h] > 0 && fl(h, Si[h]) > 0) {
                        cur = Si[h];
                        started = true;
                        fl(h, cur) = 0;
                        --nflag;
                        piece.row(i)[h] = cur;
                        plen[i] += 1;
                        last_h = h;
                        last_c = cur;
                        follow = i;
                    }
                    ++h;
                    step = 1;
                    continue;
                }
                double pred = cur;
                if (plen[i] > 1) pred = cur + (cur - last_c) / step;
                double best_d = (double)IMG;
                int best_l = n, best_h = h;
                for (int j = 0; j < n; ++j) {
                    const double sj = S.row(j)[h];
                    if (sj > 0 && fl(h, sj) > 0) {
                        const double d = std::fabs(pred - sj);
                        if (d < best_d) {
                            best_d = d;
                            best_l = j;
                            best_h = h;
                        }
                    }
                }
                const double* Sf = S.row(follow);
                for (int hh = h + 1; hh < R; ++hh) {
                    if (hh - h > BUFF_D) break;
                    if (Sf[hh] > 0 && fl(hh, Sf[hh]) > 0) {
                        const double d = std::fabs(pred - Sf[hh]);
                        if (d < best_d) {
                            best_d = d;
                            best_l = follow;
                            best_h = hh;
                        }
                        break;
                    }
                }
                if (best_d < BUFF_W) {
                    const double nv = S.row(best_l)[best_h];
                    piece.row(i)[best_h] = nv;
                    plen[i] += 1;
                    last_c = cur;
                    cur = nv;
                    fl(best_h, cur) = 0;
                    --nflag;
                    step = best_h - last_h;
                    last_h = best_h;
                    h = best_h + 1;
                    follow = best_l;
                } else {
                    piece.row(i)[h] = -1;
                    ++h;
                    ++step;
                }
            }
        }
        std::vector<int> rows, rj;
        for (int i = 0; i < n; ++i) {
            if (plen[i] <= 2) continue;
            const double* pi = piece.row(i);
            rows.clear();
            for (int h = 0; h < R; ++h)
                if (pi[h] > 0) rows.push_back(h);
            const int s_h = rows.front(), e_h = rows.back();
            const double s_v = pi[s_h], e_v = pi[e_h];
            const double e_next = e_v + (e_v - pi[rows[rows.size() - 2]]);
            bool attached = false;
            for (int j = 0; j < n && !attached; ++j) {
                if (have[j] < 2) continue;
                double* tj = total.row(j);
                rj.clear();
                for (int h = 0; h < R; ++h)
                    if (tj[h] > 0) rj.push_back(h);
                const int js_h = rj.front(), je_h = rj.back();
                const double js_v = tj[js_h], je_v = tj[je_h];
                const double j_next = je_v + (je_v - tj[rj[rj.size() - 2]]);
                const int d1 = s_h - je_h, d2 = js_h - e_h;
                if ((0 < d1 && d1 < BUFF_D && std::fabs(j_next - s_v) < BUFF_W) ||
                    (0 < d2 && d2 < BUFF_D && std::fabs(e_next - js_v) < BUFF_W)) {
                    for (int h : rows) tj[h] = pi[h];
                    have[j] += plen[i];
                    attached = true;
                }
            }
            if (!attached)
                for (int j = 0; j < n; ++j)
                    if (have[j] < 2) {
                        double* tj = total.row(j);
                        for (int h : rows) tj[h] = pi[h];
                        have[j] = plen[i];
                        break;
                    }
        }
    }
    fill_gaps(total);
    return order_left_to_right(total);
}

// polyline_utils.py:7-19 -> (min, max, mean) of |a-b| over rows where both exist; max < 0 when none
void overlap_stats(const double* a, const double* b, int R, double& mn, double& mx, double& mean) {
    mn = 1e300;
    mx = -1.0;
    double sum = 0;
    int cnt = 0;
    bool any_row = false;
    for (int h = 0; h < R; ++h) {
        any_row = true;
        if (a[h] < 0 || b[h] < 0) continue;
        const double d = std::fabs(a[h] - b[h]);
        mn = std::min(mn, d);
        mx = std::max(mx, d);
        sum += d;
        ++cnt;
    }
    (void)any_row;
    if (cnt == 0) {
        mn = -1.0;
        mx = -1.0;
        mean = -1.0;
    } else {
        mean = sum / cnt;
    }
}

// polyline_utils.py:22-45
void align_pair(double* a, double* b, int R) {
    std::vector<int> rows;
    for (int h = 0; h < R; ++h) {
        if (a[h] < 0 || b[h] < 0) continue;
        if (std::fabs(a[h] - b[h]) >= 0.00001) rows.push_back(h);
    }
    for (int r : rows) {
        if (b[r] < a[r]) std::swap(a[r], b[r]);
        if (std::fabs(a[r] - b[r]) < 2.0) {
            const int q = (r - 1 + R) % R;     // python index -1 wraps to the last row
            if (std::fabs(a[r] - a[q]) < std::fabs(b[r] - b[q]) && a[q] > 0 && b[q] > 0)
                b[r] = -1;
            else
                a[r] = -1;
        }
    }
}

int count_pos(const double* a, int R) {
    int c = 0;
    for (int h = 0; h < R; ++h) c += a[h] > 0;
    return c;
}

// polyline_utils.py:57-164
void merge_close_lines(Lines& L, const float* conf_rows) {
    const double thr = 10;
    const int n = L.n, R = L.r;
    auto conf = [&](int h, double col) { return conf_rows[(size_t)h * IMG + (int)col]; };
    for (int a = 0; a < n - 1; ++a) {
        if (count_pos(L.row(a), R) < 2) continue;
        for (int b = a + 1; b < n; ++b) {
            if (count_pos(L.row(b), R) < 2) continue;
            double mn, mx, mean;
            overlap_stats(L.row(a), L.row(b), R, mn, mx, mean);
            if (!(mn >= 0. && mn < thr)) continue;
            bool has_a = false, has_b = false;
            double last_a = 0;
            double* A = L.row(a);
            double* Bv = L.row(b);
            align_pair(A, Bv, R);
            for (int h = 0; h < R; ++h) {
                const double va = A[h], vb = Bv[h];
                if (va < 0 && vb < 0) continue;
                if (va > 0 && vb < 0) continue;
                if (va < 0 && vb > 0) {
                    if (!has_a || std::fabs(last_a - vb) < thr) {
                        A[h] = vb;
                        Bv[h] = -1.;
                        last_a = vb;
                        has_a = true;
                    } else {
                        has_b = true;
                    }
                } else if (va > 0 && vb > 0) {
                    if (std::fabs(vb - va) < thr) {
                        const double hi = conf(h, va) > conf(h, vb) ? va : vb;
                        if (!has_a && !has_b) {
                            A[h] = hi;
                            Bv[h] = -1.;
                            last_a = hi;
                            has_a = true;
                        } else if (has_a && std::fabs(last_a - hi) < thr) {
                            A[h] = hi;
                            Bv[h] = -1.;
                            last_a = hi;
                        } else {
                            A[h] = -1.;
                            Bv[h] = hi;
                            has_b = true;
                        }
                    } else if (!has_a && !has_b) {
                        if (va > vb) {
                            Bv[h] = va;
                            A[h] = vb;
                            has_b = true;
                            last_a = vb;
                            has_a = true;
                        }
                    }
                }
            }
        }
    }
    fill_gaps(L);
    for (int a = 0; a < n - 1; ++a) {
        const int na = count_pos(L.row(a), R);
        if (na < 2) {
            std::fill(L.row(a), L.row(a) + R, -1.);
            continue;
        }
        for (int b = a + 1; b < n; ++b) {
            const int nb = count_pos(L.row(b), R);
            if (nb < 2) {
                std::fill(L.row(b), L.row(b) + R, -1.);
                continue;
            }
            double mn, mx, mean;
            overlap_stats(L.row(a), L.row(b), R, mn, mx, mean);
            if (mx >= 0. && (mx < thr * 1.5 || mean < thr * 0.8)) {
                if (na < nb)
                    std::fill(L.row(a), L.row(a) + R, -1.);
                else
                    std::fill(L.row(b), L.row(b) + R, -1.);
            }
        }
    }
}

}  // namespace

LM_API int lm_endp_cluster(const int* topk_idx, int n_avail, int Wc, int clip, int k0, int k_step, int k_max,
                           int radius, int min_clusters, int* out_hw, int max_out, int* n_out, int* k_used) {
    LM_REQUIRE(topk_idx && out_hw && n_out && k_used && Wc > 0 && k0 >= 1 && k_step >= 1, "endp_cluster: bad args");
    std::vector<int> ph, pw, parent;
    auto find = [&](int a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    const long r2 = (long)radius * radius;
    int K = k0, ncl = 0;
    for (;;) {
        LM_REQUIRE(K <= n_avail, "endp_cluster: need the %d best scores but only %d were provided", K, n_avail);
        for (int i = (int)ph.size(); i < K; ++i) {
            LM_REQUIRE(topk_idx[i] >= 0, "endp_cluster: invalid index at rank %d", i);
            ph.push_back(topk_idx[i] / Wc);
            pw.push_back(topk_idx[i] % Wc);
            parent.push_back(i);
            ++ncl;
            for (int j = 0; j < i; ++j) {
                const long dh = ph[j] - ph[i], dw = pw[j] - pw[i];
                if (dh * dh + dw * dw <= r2) {
                    const int ra = find(j), rb = find(i);
                    if (ra != rb) {
                        parent[std::max(ra, rb)] = std::min(ra, rb);
                        --ncl;
                    }
                }
            }
        }
        if (ncl > min_clusters || K > k_max) break;
        K += k_step;
    }
    *k_used = K;
    std::vector<int> root(K);
    for (int i = 0; i < K; ++i) root[i] = find(i);
    int cnt = 0;
    for (int r = 0; r < K; ++r) {
        if (root[r] != r) continue;            // clusters in first-seen order
        double sh = 0, sw = 0;
        int m = 0;
        for (int i = r; i < K; ++i)
            if (root[i] == r) {
                sh += ph[i];
                sw += pw[i];
                ++m;
            }
        const double ch = sh / m, cw = sw / m;
        int best = -1;
        double bd = 0;
        for (int i = r; i < K; ++i)
            if (root[i] == r) {
                const double dh = ph[i] - ch, dw = pw[i] - cw;
                const double d = dh * dh + dw * dw;
                if (best < 0 || d < bd) {
                    best = i;
                    bd = d;
                }
            }
        LM_REQUIRE(cnt < max_out, "endp_cluster: more than %d clusters", max_out);
        out_hw[2 * cnt] = ph[best] + clip;
        out_hw[2 * cnt + 1] = pw[best] + clip;
        ++cnt;
    }
    *n_out = cnt;
    return LM_OK;
}

LM_API int lm_polyline_assemble(const float* prop_conf /*[P][2]*/, const float* prop_v_ext /*[P][R]*/,
                                const double* cls_offset /*[P][R]*/, const float* bi_seg_rows /*[R][1152]*/,
                                const int* endp_hw, int n_endp, int P, int R, float obj_thre, int min_vertices,
                                double* out_lanes /*[P][R][2]*/, int* endp_keep /*[n_endp]*/) {
    LM_REQUIRE(prop_conf && prop_v_ext && cls_offset && bi_seg_rows && out_lanes, "polyline_assemble: null pointer");
    LM_REQUIRE(R * 8 == IMG && P >= 10 && (n_endp == 0 || (endp_hw && endp_keep)), "polyline_assemble: bad shapes");
    // --- :812-837 existence gating, coordinates at image scale, semantic seed map (rows 8h+3 only) ---
    std::vector<float> vex((size_t)P * R);
    Lines C(P, R, 0.0);
    for (int p = 0; p < P; ++p) {
        const bool off = prop_conf[2 * p + 1] < obj_thre || p < 4 || p >= P - 6;   // quirk C4
        for (int h = 0; h < R; ++h) {
            float e = off ? 0.f : prop_v_ext[(size_t)p * R + h];
            e = e > 0.5f ? e : -1.f;
            vex[(size_t)p * R + h] = e;
            double c = cls_offset[(size_t)p * R + h] / (double)R * (double)IMG;
            if (e == -1.f) c = -1;
            if (c < 0) c = 0;                                                      // quirk C3
            if (c > IMG - 1) c = IMG - 1;
            C.row(p)[h] = c;
        }
    }
    std::vector<unsigned char> sem_map((size_t)R * IMG, 0);
    for (int p = 0; p < P; ++p)
        for (int h = 0; h < R; ++h) {
            const double c = C.row(p)[h];
            if (c > 0) sem_map[(size_t)h * IMG + (int)c] = (unsigned char)vex[(size_t)p * R + h];
        }
    Lines L = trace_lines(C, bi_seg_rows);                                          // :847
    merge_close_lines(L, bi_seg_rows);                                              // :848
    // --- :1091-1115 per-vertex semantics ---
    std::vector<double> S((size_t)P * R, 0.0);
    for (int i = 0; i < P; ++i) {
        const double* a = L.row(i);
        for (int r = 0; r < R - 1; ++r) {
            const int c1 = (int)a[r], c2 = (int)a[r + 1];
            if (c1 < 0 || c2 < 0) continue;
            const double colour = (sem_map[(size_t)r * IMG + c1] == 2 || sem_map[(size_t)(r + 1) * IMG + c2] == 2) ? 2 : 1;
            S[(size_t)i * R + r] = colour;
            if (r == R - 2 && c2 > 0) S[(size_t)i * R + r + 1] = colour;
        }
    }
    // --- polyline_utils.py:448-586 run-length semantic smoothing + endpoint pruning ---
    for (int k = 0; k < n_endp; ++k) endp_keep[k] = 1;
    std::vector<double> all_r, all_c;
    std::vector<std::pair<int, int>> runs;
    for (int i = 0; i < P; ++i) {
        const double* a = L.row(i);
        double* s = S.data() + (size_t)i * R;
        std::vector<int> vid;
        for (int h = 0; h < R; ++h)
            if (a[h] > 0.) vid.push_back(h);
        if (vid.size() <= 1) continue;
        for (int h : vid) {
            all_r.push_back(8.0 * h + 3.0);
            all_c.push_back(a[h]);
        }
        runs.clear();
        runs.push_back({(int)s[0], 1});
        for (int h = 1; h < R; ++h) {
            if ((int)s[h] == runs.back().first)
                runs.back().second += 1;
            else
                runs.push_back({(int)s[h], 1});
        }
        for (int v = 5; v < 20; v += 3) {
            size_t q = 1;
            while (q + 1 < runs.size()) {
                if (runs[q - 1].first > 0 && runs[q - 1].first != runs[q].first && runs[q + 1].first == runs[q - 1].first &&
                    runs[q].second < v && runs[q - 1].second - runs[q].second >= 0 && runs[q + 1].second - runs[q].second >= 0) {
                    runs[q - 1].second += runs[q].second + runs[q + 1].second;
                    runs.erase(runs.begin() + q, runs.begin() + q + 2);
                    q = 1;
                } else {
                    ++q;
                }
            }
        }
        int start = 0, best_cnt = 0;
        for (auto& rn : runs) {
            for (int h = start; h < start + rn.second; ++h) s[h] = rn.first;
            start += rn.second;
            if (rn.first > 0 && rn.second > best_cnt) best_cnt = rn.second;
        }
        if (best_cnt > 130)
            for (int k = 0; k < n_endp; ++k) {
                const double eh = endp_hw[2 * k], ew = endp_hw[2 * k + 1];
                for (int h : vid) {
                    const double dh = eh - (8.0 * h + 3.0), dw = ew - a[h];
                    if (dh * dh + dw * dw <= 64.0) {
                        endp_keep[k] = 0;
                        break;
                    }
                }
            }
    }
    if (!all_r.empty())
        for (int k = 0; k < n_endp; ++k) {
            const double eh = endp_hw[2 * k], ew = endp_hw[2 * k + 1];
            double best = 1e300;
            for (size_t q = 0; q < all_r.size(); ++q) {
                const double dh = all_r[q] - eh, dw = all_c[q] - ew;
                best = std::min(best, dh * dh + dw * dw);
            }
            if (std::sqrt(best) > 10) endp_keep[k] = 0;
        }
    // --- :589-608 drop short lines, pack [P][R][2] ---
    for (int i = 0; i < P; ++i) {
        const double* a = L.row(i);
        const bool keep = count_pos(a, R) >= min_vertices;
        for (int h = 0; h < R; ++h) {
            out_lanes[((size_t)i * R + h) * 2 + 0] = keep ? a[h] : -1.;
            out_lanes[((size_t)i * R + h) * 2 + 1] = keep ? S[(size_t)i * R + h] : 0.;
        }
    }
    return LM_OK;
}

// polyline_utils.py:610-638 (renew_semantic_map): cv2.line(map, pt1, pt2, color, thickness=1) = OpenCV's 8-connected Bresenham.
// cv2 is not in the image and its source is not part of /root/reference (a third-party dependency, opencv-python, version not pinned by
// the reference), so this restates the published algorithm of OpenCV 4.x `LineIterator` as `cv::line` drives it for thickness 1,
// LINE_8, shift 0 (modules/imgproc/src/drawing.cpp: Line() -> LineIterator(img, pt1, pt2, 8, leftToRight = true)):
//   * the segment is walked from its LEFT end point (end points swapped when pt2.x < pt1.x), so the pixels do not depend on the order
//     of the end points;
//   * major axis = the longer of |dx|, |dy| (ties: x), count = major + 1 pixels, both end points drawn;
//   * err = major - 2 minor; per step: a minor-axis move is added when err < 0 (then err += 2 major - 2 minor, else err -= 2 minor).
//     A tie (err == 0) keeps the minor coordinate - the textbook all-octant form (err = dx - dy, e2 = 2 err) used here until round 4
//     takes the diagonal there and differs from OpenCV in up to major / 2 pixels of a segment.
// PARITY UNPINNED against OpenCV itself (no reference-held vector); pinned to the hand-derived table of
// tests/test_metrics_io_cpu.py::test_line8_opencv_table and to the independently written oracle (oracle/postproc_ref.py _line8).
static void line8_opencv(unsigned char* out, int x1, int y1, int x2, int y2, unsigned char colour) {
    if (x2 < x1) {                                       // leftToRight
        std::swap(x1, x2);
        std::swap(y1, y2);
    }
    const int dx = x2 - x1, ady = std::abs(y2 - y1), sy = y2 < y1 ? -1 : 1;
    const bool ymajor = ady > dx;
    const int major = ymajor ? ady : dx, minor = ymajor ? dx : ady;
    int err = major - 2 * minor, x = x1, y = y1;
    for (int k = 0; k <= major; ++k) {
        if ((unsigned)y < (unsigned)IMG && (unsigned)x < (unsigned)IMG) out[(size_t)y * IMG + x] = colour;
        const bool both = err < 0;
        err += -2 * minor + (both ? 2 * major : 0);
        if (ymajor) {
            y += sy;
            if (both) ++x;
        } else {
            ++x;
            if (both) y += sy;
        }
    }
}

LM_API int lm_raster_polylines(const double* lanes /*[P][R][2]*/, int P, int R, unsigned char* out /*[1152][1152]*/) {
    LM_REQUIRE(lanes && out && R * 8 == IMG, "raster_polylines: bad args");
    std::memset(out, 0, (size_t)IMG * IMG);
    for (int i = 0; i < P; ++i)
        for (int r = 0; r < R - 1; ++r) {
            const int c1 = (int)lanes[((size_t)i * R + r) * 2], c2 = (int)lanes[((size_t)i * R + r + 1) * 2];
            if (c1 < 0 || c2 < 0) continue;
            const int s1 = (int)lanes[((size_t)i * R + r) * 2 + 1], s2 = (int)lanes[((size_t)i * R + r + 1) * 2 + 1];
            line8_opencv(out, c1, r * 8 + 3, c2, (r + 1) * 8 + 3, (s1 == 2 || s2 == 2) ? 2 : 1);
        }
    return LM_OK;
}

// one segment of the rasteriser above on an IMG x IMG map (test hook of the OpenCV table)
LM_API int lm_line8(unsigned char* out /*[1152][1152]*/, int x1, int y1, int x2, int y2, int colour) {
    LM_REQUIRE(out, "line8: null pointer");
    line8_opencv(out, x1, y1, x2, y2, (unsigned char)colour);
    return LM_OK;
}

// smooth_cls_line_per_batch alone (polyline_utils.py:222-387): used by the RowRef head (config 4), which calls it with
// no segmentation confidence (seg_rows == NULL -> no occupancy filter) on its 12 lane rows.
LM_API int lm_trace_lines(const double* cols /*[n][R]*/, int n, int R, const float* seg_rows /*[R][1152] or NULL*/, double* out) {
    LM_REQUIRE(cols && out && n >= 1 && R * 8 == IMG, "trace_lines: bad args");
    Lines C(n, R, 0.0);
    std::memcpy(C.v.data(), cols, sizeof(double) * (size_t)n * R);
    Lines L = trace_lines(C, seg_rows);
    std::memcpy(out, L.v.data(), sizeof(double) * (size_t)n * R);
    return LM_OK;
}

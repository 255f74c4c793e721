"""Oracle (test infrastructure): per-tile polyline assembly in NumPy f64.

Restates, with the reference's quirks kept:
  assemble_tile              <- heads/polyline_fpn_vit_vertex_2.py:805-861
  order_left_to_right        <- utils/polyline_utils.py:167-178   (stable order, quirk C16)
  thin_row0                  <- utils/polyline_utils.py:200-220   (row 0 only, quirk C1)
  trace_lines                <- utils/polyline_utils.py:222-387
  fill_gaps                  <- utils/polyline_utils.py:180-198
  overlap_stats / align_pair <- utils/polyline_utils.py:7-45
  merge_close_lines          <- utils/polyline_utils.py:57-164
  vertex_semantics           <- heads/polyline_fpn_vit_vertex_2.py:1091-1115
  smooth_semantics           <- utils/polyline_utils.py:448-586
  drop_short                 <- utils/polyline_utils.py:589-608
  raster_semantic_map        <- utils/polyline_utils.py:610-638 (cv2.line replaced by an own
                                OpenCV's 8-connected Bresenham restated: parity vs OpenCV itself unpinned)
  lanes_to_json_records      <- utils/io_utils.py:58-93 and the packing at
                                heads/polyline_fpn_vit_vertex_2.py:997-1000
"""
import numpy as np

IMG = 1152
BUFF_W = 6
BUFF_D = 24


def order_left_to_right(L):
    key = np.full(L.shape[0], float(IMG))
    for i in range(L.shape[0]):
        nz = np.nonzero(L[i] >= 0)[0]
        if nz.size:
            key[i] = L[i, nz[0]]
    return L[np.argsort(key, kind='stable')]


def thin_row0(flag, conf, half=4):
    out = flag.copy()
    if flag.shape[0] == 0:
        return out
    for c in range(half, flag.shape[1] - half):
        win = out[0, c - half:c + half]
        if win.sum() > 1:
            cand = np.nonzero(win > 0)[0]
            best = cand[0]
            for j in cand:
                if conf[0, c - half + j] > conf[0, c - half + best]:
                    best = j
            out[0, c - half:c + half] = 0
            out[0, c - half + best] = 1.0
    return out


def fill_gaps(L):
    for i in range(L.shape[0]):
        pos = np.nonzero(L[i] > 1e-4)[0]
        if pos.size > 1:
            cur = -1
            for v in range(pos[0], pos[-1]):
                if L[i, v] < 1e-4:
                    a, b = pos[cur], pos[cur + 1]
                    r = (1.0 * v - a) / (b - a)
                    L[i, v] = (1 - r) * L[i, a] + L[i, b] * r
                else:
                    cur += 1
    return L


def trace_lines(C, seg_conf):
    """C [n_line, n_row] f64 column px (0 = none), seg_conf [1152,1152] -> traced lines [n_line,n_row] (-1 = none)."""
    n_line, n_row = C.shape
    S = order_left_to_right(C)
    total = np.full_like(C, -1.0)
    have = np.zeros(n_line)
    flag = np.zeros((n_row, IMG))
    for i in range(n_line):
        r = np.nonzero(C[i] > 0)[0]
        flag[r, C[i, r].astype(int)] = 1
    if seg_conf is not None:
        flag = thin_row0(flag, seg_conf[3:IMG:8, :])
    while flag.sum() > 2 and have.min() < 2:
        piece = np.full_like(C, -1.0)
        plen = np.zeros(n_line)
        for i in range(n_line):
            started = False
            last_h = 0
            h = 0
            follow = i
            last_c = 0.0
            cur = 0.0
            step = 1
            while h < n_row:
                if started and (h - last_h > BUFF_D):
                    break
                if not started:
                    if S[i, h] > 0 and flag[h, int(S[i, h])] > 0:
                        cur = S[i, h]
                        started = True
                        flag[h, int(cur)] = 0
                        piece[i, h] = cur
                        plen[i] += 1
                        last_h = h
                        last_c = cur
                        follow = i
                    h += 1
                    step = 1
                    continue
                pred = cur
                if plen[i] > 1:
                    pred = cur + (cur - last_c) / step
                best_d, best_l, best_h = float(IMG), n_line, h
                for j in range(n_line):                       # any line's vertex on this row
                    if S[j, h] > 0 and flag[h, int(S[j, h])] > 0:
                        d = abs(pred - S[j, h])
                        if d < best_d:
                            best_d, best_l, best_h = d, j, h
                for hh in range(h + 1, n_row):                # first flagged vertex of the followed line
                    if hh - h > BUFF_D:
                        break
                    if S[follow, hh] > 0 and flag[hh, int(S[follow, hh])] > 0:
                        d = abs(pred - S[follow, hh])
                        if d < best_d:
                            best_d, best_l, best_h = d, follow, hh
                        break
                if best_d < BUFF_W:
                    piece[i, best_h] = S[best_l, best_h]
                    plen[i] += 1
                    last_c = cur
                    cur = S[best_l, best_h]
                    flag[best_h, int(cur)] = 0
                    step = best_h - last_h
                    last_h = best_h
                    h = best_h + 1
                    follow = best_l
                else:
                    piece[i, h] = -1
                    h += 1
                    step += 1
        for i in range(n_line):
            if plen[i] <= 2:
                continue
            rows = np.nonzero(piece[i] > 0)[0]
            s_h, e_h = rows[0], rows[-1]
            s_v, e_v = piece[i, s_h], piece[i, e_h]
            e_next = e_v + (e_v - piece[i, rows[-2]])
            attached = False
            for j in range(n_line):
                if have[j] >= 2:
                    rj = np.nonzero(total[j] > 0)[0]
                    js_h, je_h = rj[0], rj[-1]
                    js_v, je_v = total[j, js_h], total[j, je_h]
                    j_next = je_v + (je_v - total[j, rj[-2]])
                    if (0 < (s_h - je_h) < BUFF_D and abs(j_next - s_v) < BUFF_W) or \
                            (0 < (js_h - e_h) < BUFF_D and abs(e_next - js_v) < BUFF_W):
                        total[j, rows] = piece[i, rows]
                        have[j] += plen[i]
                        attached = True
                        break
            if not attached:
                for j in range(n_line):
                    if have[j] < 2:
                        total[j, rows] = piece[i, rows]
                        have[j] = plen[i]
                        break
    total = fill_gaps(total)
    return order_left_to_right(total)


def overlap_stats(a, b):
    d = np.abs(a - b)
    d[a < 0] = -1
    d[b < 0] = -1
    mx = d.max()
    if mx < 0.:
        return -1., mx, -1.
    ok = d[d >= 0]
    return ok.min(), mx, ok.mean()


def align_pair(a, b):
    d = np.abs(a - b)
    d[a < 0] = -1
    d[b < 0] = -1
    for r in np.nonzero(d >= 0.00001)[0]:
        if b[r] < a[r]:
            a[r], b[r] = b[r], a[r]
        if abs(a[r] - b[r]) < 2.0:
            if abs(a[r] - a[r - 1]) < abs(b[r] - b[r - 1]) and a[r - 1] > 0 and b[r - 1] > 0:
                b[r] = -1
            else:
                a[r] = -1
    return a, b


def merge_close_lines(L, conf, thr=10):
    n_line, n_row = L.shape
    for a in range(n_line - 1):
        if np.count_nonzero(L[a] > 0) < 2:
            continue
        for b in range(a + 1, n_line):
            if np.count_nonzero(L[b] > 0) < 2:
                continue
            mn, _, _ = overlap_stats(L[a], L[b])
            if not (mn >= 0. and mn < thr):
                continue
            last_a = None
            last_b = None
            L[a], L[b] = align_pair(L[a], L[b])
            for h in range(n_row):
                va, vb = L[a, h], L[b, h]
                if va < 0 and vb < 0:
                    continue
                elif va > 0 and vb < 0:
                    continue
                elif va < 0 and vb > 0:
                    if last_a is None or abs(last_a - vb) < thr:
                        L[a, h] = vb
                        L[b, h] = -1.
                        last_a = L[a, h]
                    else:
                        last_b = vb
                elif va > 0 and vb > 0:
                    if abs(vb - va) < thr:
                        hi = va if conf[h * 8 + 3, int(va)] > conf[h * 8 + 3, int(vb)] else vb
                        if (last_a is None) and (last_b is None):
                            L[a, h] = hi
                            L[b, h] = -1.
                            last_a = hi
                        elif abs(last_a - hi) < thr:      # last_a None here raises in the reference too
                            L[a, h] = hi
                            L[b, h] = -1.
                            last_a = hi
                        else:
                            L[a, h] = -1.
                            L[b, h] = hi
                            last_b = hi
                    elif (last_a is None) and (last_b is None):
                        if va > vb:
                            L[b, h], L[a, h] = va, vb
                            last_b = L[b, h]
                            last_a = L[a, h]
    L = fill_gaps(L)
    for a in range(n_line - 1):
        na = np.count_nonzero(L[a] > 0)
        if na < 2:
            L[a] = -1.
            continue
        for b in range(a + 1, n_line):
            nb = np.count_nonzero(L[b] > 0)
            if nb < 2:
                L[b] = -1.
                continue
            _, mx, mean = overlap_stats(L[a], L[b])
            if mx >= 0. and (mx < thr * 1.5 or mean < thr * 0.8):
                if na < nb:
                    L[a] = -1.
                else:
                    L[b] = -1.
    return L


def vertex_semantics(L, sem_map):
    out = np.zeros_like(L)
    n_line, n_row = L.shape
    for i in range(n_line):
        for r in range(n_row - 1):
            c1, c2 = int(L[i, r]), int(L[i, r + 1])
            if c1 < 0 or c2 < 0:
                continue
            colour = 2 if (sem_map[r * 8 + 3, c1] == 2 or sem_map[(r + 1) * 8 + 3, c2] == 2) else 1
            out[i, r] = colour
            if r == n_row - 2 and c2 > 0:
                out[i, r + 1] = colour
    return out


def smooth_semantics(V, endp, max_void=20):
    """V [n_line,n_row,2] (col, semantic); endp [1152,1152] modified in place."""
    n_line, n_row, _ = V.shape
    eh, ew = np.nonzero(endp > 0)
    E = np.stack([eh, ew], axis=1).astype(np.float64)
    rows_px = np.arange(3, IMG, 8).astype(np.float64)
    all_v = []
    for i in range(n_line):
        vid = np.nonzero(V[i, :, 0] > 0.)[0]
        if vid.size <= 1:
            continue
        pts = np.stack([rows_px[vid], V[i, vid, 0]], axis=1)
        all_v.append(pts)
        runs = [[int(V[i, 0, 1]), 1]]
        for r in range(1, n_row):
            if V[i, r, 1] == runs[-1][0] and True:
                runs[-1][1] += 1
            else:
                runs.append([int(V[i, r, 1]), 1])
        # NB: the reference compares against the tracked "current" semantic, which equals runs[-1][0]
        void = 5
        while void < max_void:
            s = 1
            while s < len(runs) - 1:
                if runs[s - 1][0] > 0 and runs[s - 1][0] != runs[s][0] and runs[s + 1][0] == runs[s - 1][0] \
                        and runs[s][1] < void and runs[s - 1][1] - runs[s][1] >= 0 and runs[s + 1][1] - runs[s][1] >= 0:
                    runs[s - 1][1] += runs[s][1] + runs[s + 1][1]
                    del runs[s]
                    del runs[s]
                    s = 1
                else:
                    s += 1
            void += 3
        start = 0
        for sem, cnt in runs:
            V[i, start:start + cnt, 1] = sem
            start += cnt
        best_cnt = 0
        for sem, cnt in runs:
            if sem > 0 and cnt > best_cnt:
                best_cnt = cnt
        if best_cnt > 130 and len(E):
            d2 = ((E[:, None, :] - pts[None, :, :]) ** 2).sum(2)
            hit = (np.sqrt(d2) <= 8).any(axis=1)
            endp[eh[hit], ew[hit]] = 0.
    if all_v and len(E):
        A = np.concatenate(all_v, axis=0)
        for k in range(len(E)):
            d = np.sqrt(((A - E[k]) ** 2).sum(1)).min()
            if d > 10:
                endp[eh[k], ew[k]] = 0
    return V, endp


def drop_short(V, min_v=8):
    for i in range(V.shape[0]):
        if np.count_nonzero(V[i, :, 0] > 0.) < min_v:
            V[i, :, 0] = -1.
            V[i, :, 1] = 0.
    return V


def _line8(img, x0, y0, x1, y1, colour):
    """cv2.line(img, (x0, y0), (x1, y1), colour, thickness=1): OpenCV's 8-connected Bresenham as cv::line runs it (LineIterator with
    leftToRight=True; OpenCV 4.x modules/imgproc/src/drawing.cpp - a third-party dependency of the reference, opencv-python, version not
    pinned there).  Written from the error-term definition, not from the product's loop: pixel k of the walk from the LEFT end point has
    minor-axis offset = the number of j in 1..k with  major - 2 * minor * j + 2 * major * (moves so far) < 0  evaluated step by step."""
    if x1 < x0:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, y1 - y0
    sy = -1 if dy < 0 else 1
    major, minor = (abs(dy), dx) if abs(dy) > dx else (dx, abs(dy))
    ymajor = abs(dy) > dx
    moved = 0
    for k in range(major + 1):
        x, y = (x0 + moved, y0 + sy * k) if ymajor else (x0 + k, y0 + sy * moved)
        if 0 <= y < img.shape[0] and 0 <= x < img.shape[1]:
            img[y, x] = colour
        if major - 2 * minor * (k + 1) + 2 * major * moved < 0:          # err after k steps, before the decision of step k + 1
            moved += 1


def raster_semantic_map(V):
    img = np.zeros((IMG, IMG))
    for i in range(V.shape[0]):
        for r in range(V.shape[1] - 1):
            c1, c2 = int(V[i, r, 0]), int(V[i, r + 1, 0])
            if c1 < 0 or c2 < 0:
                continue
            colour = 2 if (int(V[i, r, 1]) == 2 or int(V[i, r + 1, 1]) == 2) else 1
            _line8(img, c1, r * 8 + 3, c2, (r + 1) * 8 + 3, colour)
    return img


def assemble_tile(prop_conf1, prop_v_ext, cls_offset, bi_seg, endp, obj_thre=0.3, row_size=144,
                  want_raster=False):
    """One tile.  prop_conf1 [72] f32 (existence prob), prop_v_ext [72,144] f32 in {0,1,2},
    cls_offset [72,144] f64, bi_seg [1152,1152] f32, endp [1152,1152] f32 in {0,1}.
    Returns (cls_offset_smooth [72,144,2] f64, endp_by_cls [1152,1152], semantic_line or None)."""
    ext = np.array(prop_v_ext, dtype=np.float32, copy=True)
    ext[np.asarray(prop_conf1, dtype=np.float32) < np.float32(obj_thre)] = 0.        # :812-813
    ext[0:4] = 0.                                                                     # :815 (quirk C4)
    ext[-6:] = 0.                                                                     # :816
    vex = np.where(ext > 0.5, ext, -1)                                                # :819
    C = np.asarray(cls_offset, dtype=np.float64) / row_size * IMG                     # :830
    C = np.where(vex == -1, -1, C)
    C[C < 0] = 0                                                                      # :832 (quirk C3)
    C[C > IMG - 1] = IMG - 1
    sem_map = np.zeros((IMG, IMG))
    for i in range(C.shape[0]):                                                       # :835-837
        r = np.nonzero(C[i] > 0)[0]
        sem_map[r * 8 + 3, C[i, r].astype(int)] = vex[i, r]
    seg = np.asarray(bi_seg)
    L = trace_lines(C, seg)                                                           # :847
    L = merge_close_lines(L, seg)                                                     # :848
    S = vertex_semantics(L, sem_map)                                                  # :854
    V = np.stack([L, S], axis=2)
    E = np.array(endp, dtype=np.float32, copy=True)
    V, E = smooth_semantics(V, E, 20)                                                 # :856
    V = drop_short(V, 8)                                                              # :857
    return V, E, (raster_semantic_map(V) if want_raster else None)


def lanes_to_json_records(V):
    """[72,144,2] -> list of dicts as save_lane_seq_2d writes them (io_utils.py:58-93); vertex =
    (row px = 3+8i, col, semantic) (heads/...vertex_2.py:997-1000)."""
    recs = []
    rows_px = np.arange(3, IMG, 8).astype(np.float64)
    for i in range(V.shape[0]):
        keep = V[i, :, 0] > 0
        pv = np.stack([rows_px[keep], V[i, keep, 0], V[i, keep, 1]], axis=1)
        if pv.shape[0] < 2:
            continue
        recs.append({'seq_len': int(pv.shape[0]), 'seq': pv.tolist(),
                     'init_vertex': pv[0].tolist(), 'end_vertex': pv[-1].tolist()})
    return recs

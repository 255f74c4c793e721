"""Oracle (test infrastructure, PARITY UNPINNED): Lee-Kashyap-Chu 3-D thinning applied to a 2-D image as a one-slice volume.

skimage is absent from the build container, so `skimage.morphology.skeletonize(method='lee')` - what the reference's
eval_metric_line_segmentor calls (baseline/utils/metric_utils.py:430,453) - cannot be imported; this is a slow, literal restatement
of the published algorithm (Lee, Kashyap, Chu, CVGIP 56(6), 1994) in its 3-D form: 26-neighbourhoods, Euler characteristic of the
cubical complex of the 3x3x3 block, simple-point test by 26-connected labelling of the neighbours.  The product
(csrc/skeleton.cpp) uses the 2-D reduction with 256-entry tables; the two are compared on small images.
"""
import itertools

import numpy as np

_OFFS = [o for o in itertools.product((-1, 0, 1), repeat=3) if o != (0, 0, 0)]


def _euler(cells):
    """Euler characteristic of the union of closed unit cubes at the integer positions `cells` (V - E + F - C)."""
    V, E, F = set(), set(), set()
    for (z, y, x) in cells:
        for dz, dy, dx in itertools.product((0, 1), repeat=3):
            V.add((z + dz, y + dy, x + dx))
        for a in range(3):                      # edges along axis a
            for d1, d2 in itertools.product((0, 1), repeat=2):
                p = [z, y, x]
                o = [i for i in range(3) if i != a]
                p[o[0]] += d1
                p[o[1]] += d2
                E.add((a, tuple(p)))
        for a in range(3):                      # faces normal to axis a
            for d in (0, 1):
                p = [z, y, x]
                p[a] += d
                F.add((a, tuple(p)))
    return len(V) - len(E) + len(F) - len(cells)


def _neigh(vol, p, r, c):
    return [(o, vol[p + o[0], r + o[1], c + o[2]]) for o in _OFFS]


def _is_endpoint(nb):
    return sum(1 for _, v in nb if v) == 1


def _euler_invariant(nb):
    cells = [o for o, v in nb if v]
    return _euler(cells + [(0, 0, 0)]) == _euler(cells)


def _is_simple(nb):
    cells = [o for o, v in nb if v]
    if not cells:
        return False
    seen, stack = {cells[0]}, [cells[0]]
    while stack:
        a = stack.pop()
        for b in cells:
            if b not in seen and max(abs(a[i] - b[i]) for i in range(3)) <= 1:
                seen.add(b)
                stack.append(b)
    return len(seen) == len(cells)


def skeletonize_lee_ref(image):
    img = (np.asarray(image) != 0).astype(np.uint8)
    vol = np.pad(img[None], 1)
    H, W = img.shape
    step = {4: (0, -1, 0), 3: (0, 1, 0), 2: (0, 0, 1), 1: (0, 0, -1)}
    unchanged = 0
    while unchanged < 4:
        unchanged = 0
        for border in (4, 3, 2, 1):             # one-slice volume: the two z directions are skipped
            dz, dy, dx = step[border]
            cand = []
            for r in range(1, H + 1):
                for c in range(1, W + 1):
                    if not vol[1, r, c] or vol[1 + dz, r + dy, c + dx]:
                        continue
                    nb = _neigh(vol, 1, r, c)
                    if _is_endpoint(nb) or not _euler_invariant(nb) or not _is_simple(nb):
                        continue
                    cand.append((r, c))
            no_change = True
            for r, c in cand:
                if _is_simple(_neigh(vol, 1, r, c)):
                    vol[1, r, c] = 0
                    no_change = False
            if no_change:
                unchanged += 1
    return vol[1, 1:-1, 1:-1].copy()

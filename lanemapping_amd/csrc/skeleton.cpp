// Lee-Kashyap-Chu thinning of a 2-D binary image - the skeletonisation step of the reference's semantic-line F1
// (baseline/utils/metric_utils.py:415-481 calls skimage.morphology.skeletonize(method='lee')).  SURVEY §8f row f3, host C++.
//
// PARITY UNPINNED: skimage is absent from the build container, so this restates the published algorithm (T.-C. Lee, R. L. Kashyap,
// C.-N. Chu, "Building skeleton models via 3-D medial surface/axis thinning algorithms", CVGIP 56(6), 1994) in the form skimage
// applies to a 2-D image (a one-slice volume): repeated sub-iterations over the border directions N, S, E, W; in each one the object
// pixels whose neighbour in that direction is background and that are (1) not an end point (exactly one neighbour), (2) Euler
// invariant and (3) simple (their object neighbours form one connected set) are collected in raster order, then re-checked for
// simplicity one by one - deletions of the same sub-iteration are visible to later candidates - and deleted.  Stops after a full
// cycle without a deletion.
// For a one-slice volume the 26-neighbourhood reduces to the 8 in-plane neighbours.  The Euler test is derived here from first
// principles instead of Lee's octant table: deleting p must not change the Euler characteristic of the union of closed unit squares
// of the 3x3 neighbourhood (vertices - edges + faces of the cubical complex), tabulated once for the 256 neighbour patterns.
#include "common.h"

#include <cstring>
#include <vector>

namespace {

// neighbour bit k <-> (dr, dc): raster order without the centre
const int DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
const int DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

// Euler characteristic of the union of closed unit squares of the set cells (3x3 grid given as 9 flags, row major)
int euler_of(const bool (&cell)[9]) {
    bool vtx[4][4] = {}, eh[4][3] = {}, ev[3][4] = {};      // vertices, horizontal edges [row 0..3][col 0..2], vertical edges [row 0..2][col 0..3]
    int faces = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (cell[r * 3 + c]) {
                ++faces;
                vtx[r][c] = vtx[r][c + 1] = vtx[r + 1][c] = vtx[r + 1][c + 1] = true;
                eh[r][c] = eh[r + 1][c] = true;
                ev[r][c] = ev[r][c + 1] = true;
            }
    int v = 0, e = 0;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) v += vtx[r][c];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c) e += eh[r][c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) e += ev[r][c];
    return v - e + faces;
}

struct Tables {
    bool euler_invariant[256];      // deleting the centre keeps the Euler characteristic
    bool simple[256];               // the object neighbours form exactly one 8-connected set (no neighbours: not simple)
    int count[256];
    Tables() {
        for (int m = 0; m < 256; ++m) {
            bool with[9], without[9];
            int k = 0, n = 0;
            for (int i = 0; i < 9; ++i) {
                if (i == 4) {
                    with[i] = true;
                    without[i] = false;
                    continue;
                }
                with[i] = without[i] = (m >> k) & 1;
                n += (m >> k) & 1;
                ++k;
            }
            count[m] = n;
            euler_invariant[m] = euler_of(with) == euler_of(without);
            // connected components of the neighbours under 8-adjacency (the centre removed)
            int label[8];
            for (int i = 0; i < 8; ++i) label[i] = ((m >> i) & 1) ? i : -1;
            bool changed = true;
            while (changed) {
                changed = false;
                for (int a = 0; a < 8; ++a)
                    for (int b = 0; b < 8; ++b)
                        if (label[a] >= 0 && label[b] >= 0 && label[a] != label[b]) {
                            const int dr = DR[a] - DR[b], dc = DC[a] - DC[b];
                            if (dr >= -1 && dr <= 1 && dc >= -1 && dc <= 1) {
                                const int lo = label[a] < label[b] ? label[a] : label[b];
                                label[a] = label[b] = lo;
                                changed = true;
                            }
                        }
            }
            int comps = 0;
            for (int i = 0; i < 8; ++i) comps += label[i] == i;
            simple[m] = comps == 1;
        }
    }
};

inline int pattern(const unsigned char* p, long ld) {      // p -> padded image centre
    return (p[-ld - 1] ? 1 : 0) | (p[-ld] ? 2 : 0) | (p[-ld + 1] ? 4 : 0) | (p[-1] ? 8 : 0) | (p[1] ? 16 : 0) | (p[ld - 1] ? 32 : 0) |
           (p[ld] ? 64 : 0) | (p[ld + 1] ? 128 : 0);
}

}  // namespace

// img [H][W] u8, nonzero = object; thinned in place to 0 / 1.  Returns the number of deleted pixels, -1 on bad arguments.
LM_API long lm_skeletonize_lee_2d(unsigned char* img, int H, int W) {
    if (!img || H < 1 || W < 1) return -1;
    static const Tables T;
    const long ld = W + 2;
    std::vector<unsigned char> pad((size_t)(H + 2) * ld, 0);
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) pad[(size_t)(r + 1) * ld + c + 1] = img[(long)r * W + c] ? 1 : 0;
    const long step[4] = {-ld, ld, 1, -1};          // border directions in skimage's order: N (4), S (3), E (2), W (1)
    std::vector<long> cand;
    long deleted = 0;
    int unchanged = 0;
    while (unchanged < 4) {
        unchanged = 0;
        for (int d = 0; d < 4; ++d) {
            cand.clear();
            for (int r = 1; r <= H; ++r) {
                const unsigned char* row = pad.data() + (long)r * ld;
                for (int c = 1; c <= W; ++c) {
                    if (!row[c] || row[c + step[d]]) continue;
                    const int m = pattern(row + c, ld);
                    if (T.count[m] == 1 || !T.euler_invariant[m] || !T.simple[m]) continue;
                    cand.push_back((long)r * ld + c);
                }
            }
            bool no_change = true;
            for (long at : cand)
                if (T.simple[pattern(pad.data() + at, ld)]) {       // sequential re-check: earlier deletions of this pass count
                    pad[(size_t)at] = 0;
                    no_change = false;
                    ++deleted;
                }
            if (no_change) ++unchanged;
        }
    }
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) img[(long)r * W + c] = pad[(size_t)(r + 1) * ld + c + 1];
    return deleted;
}

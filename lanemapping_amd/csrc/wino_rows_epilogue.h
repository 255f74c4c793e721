// Epilogue of wino_rows_split_kernel (conv_wino.hip), #included INSIDE the kernel body: it uses the kernel's local names (acc[16] =
// [j][tile half][channel half], smem, wave, lane, frow, fhalf, n0, p, g, ts / sn / oy0 / ox0 run table, img_pix0, bi, t0).
// Wave w holds ROW w of the 4 x 4 products.  Output (a, b) of the 2 x 2 = sum_w At[a][w] * (sum_j At[b][j] * M[w][j]),
// At = [[1, 1, 1, 0], [0, 1, -1, -1]]: every wave folds its own four xi for both output columns (column pass) and puts the two terms
// into the exchange area (2 x 4 rows x 4 blocks x 4.5 KB = 144 KB, one round); the wave that owns a (tile half, channel half) block
// combines three row terms per output in ascending row order, then BN scale / shift, residual (loads issued before the combine), ReLU,
// 16-byte stores, optional GroupNorm partial sums - the tail of wino_pipe_epilogue.h.
    float* const xch = smem;                           // [b][row w][block][32 tiles x RELD]
    constexpr int LPR = 8, RPI = 8, NP = 4;
    const int oms = wave >> 1, ons = wave & 1;         // the block this wave finishes
    const int c4 = (lane & 7) * 4;
    const int n = n0 + ons * 32 + c4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) {
                if (p.scale) sc[e] = p.scale[n + e];
                if (p.shift) sh[e] = p.shift[n + e];
            }
    }
    const bool vec = (n + 3 < p.Cout) && ((p.ldy & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
    f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};
    int pix0[NP];
    unsigned vmask = 0;                                // 3 bits per row: tile exists | a = 1 inside | b = 1 inside
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int tl = oms * 32 + pass * RPI + lane / LPR;
        int nn = sn[0], oy = oy0[0], oxb = ox0[0], tb = 0;
#pragma unroll
        for (int k = 1; k < INSEG; ++k)
            if (tl >= ts[k]) {
                nn = sn[k]; oy = oy0[k]; oxb = ox0[k]; tb = ts[k];
            }
        const int ox = oxb + 2 * (tl - tb) * g.dil;
        pix0[pass] = img_pix0 + oy * g.W + ox;
        if (nn > 0 && oy < g.H && ox < g.W)
            vmask |= (1u | (oy + g.dil < g.H ? 2u : 0u) | (ox + g.dil < g.W ? 4u : 0u)) << (3 * pass);
    }
    const int step_a = g.dil * g.W, step_b = g.dil;
    // one exchange round: every wave folds its row for both output columns b (f_b = sum_j At[b][j] * M[w][j]) and publishes the terms;
    // the owner of a block combines output (a, b) = sum_w At[a][w] * f_b[w]: rows 0 1 2 for a = 0, rows 1 -2 -3 for a = 1
    __syncthreads();                                   // every wave is done with the patch buffers
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float* const st = xch + ((b * 4 + wave) * 4 + blk) * (32 * RELD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float o = b == 0 ? (acc[blk][r] + acc[4 + blk][r]) + acc[8 + blk][r] : (acc[4 + blk][r] - acc[8 + blk][r]) - acc[12 + blk][r];
                st[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * RELD + frow] = o;
            }
        }
    LM_TICK(9)
    const float* const rd = xch + wave * (32 * RELD) + (lane / LPR) * RELD + c4;      // + (b * 4 + row w) * 4 blocks, + pass * RPI rows
    __syncthreads();
    LM_TICK(10)
    if (n < p.Cout) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                // residual (BasicBlock identity): the four 16-byte loads of this output position go out before its combine
                f32x4 rpre[NP];
                if (vec && p.res) {
#pragma unroll
                    for (int pass = 0; pass < NP; ++pass) {
                        const unsigned vm = vmask >> (3 * pass);
                        const bool ok = (vm & 1u) && (!a || (vm & 2u)) && (!b || (vm & 4u));
                        const long pix = pix0[pass] + a * step_a + b * step_b;
                        rpre[pass] = ok ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                // the twelve reads of the position first, unconditionally (a row without a tile reads finite stale data that is never stored)
                f32x4 vv[NP];
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
                    const float* const src = rd + pass * (RPI * RELD) + (b * 4 + a) * (4 * 32 * RELD);      // rows a, a + 1, a + 2
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4 * 32 * RELD);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + 8 * 32 * RELD);
#pragma unroll
                    for (int e = 0; e < 4; ++e) vv[pass][e] = a == 0 ? (v0[e] + v1[e]) + v2[e] : (v0[e] - v1[e]) - v2[e];
                }
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
                    const unsigned vm = vmask >> (3 * pass);
                    if (!(vm & 1u) || (a && !(vm & 2u)) || (b && !(vm & 4u))) continue;
                    const long pix = pix0[pass] + a * step_a + b * step_b;
                    f32x4 v = vv[pass];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p.scale ? v[e] * sc[e] + sh[e] : v[e] + sh[e];
                    if (p.gn_part) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gs[e] += v[e];
                            gq[e] = fmaf(v[e], v[e], gq[e]);
                        }
                    }
                    if (vec) {
                        if (p.res) {
                            const f32x4 rres = rpre[pass];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += rres[e];
                        }
                        if (p.act == LM_ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + n) = v;
                    } else {
                        for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                            float u = v[e];
                            if (p.res) u += p.res[pix * p.ldr + n + e];
                            if (p.act == LM_ACT_RELU) u = fmaxf(u, 0.f);
                            p.y[pix * p.ldy + n + e] = u;
                        }
                    }
                }
            }
    }
    if (p.gn_part && n < p.Cout) {   // fixed-order reduction over the 8 lanes that share a channel quad, then one writer lane
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gs[e] += __shfl_xor(gs[e], o);
                gq[e] += __shfl_xor(gq[e], o);
            }
        if (lane < LPR) {
            const long chunk = t0 / 32 + oms;
            double* o = p.gn_part + (((long)bi * (g.Tpad / 32) + chunk) * p.Cout + n) * 2;
            for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                o[2 * e] = (double)gs[e];
                o[2 * e + 1] = (double)gq[e];
            }
        }
    }

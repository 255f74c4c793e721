// Epilogue of wino_pipe_kernel (conv_wino.hip), #included INSIDE the kernel body (a file of its own since a split-precision kernel of the
// same geometry shared it in round 3; that kernel was superseded by wino_rows_split_kernel): it uses the kernel's local
// names (acc[16] accumulators, smem, wave, lane, frow, fhalf, n0, wn0, p, g, ts / sn / oy0 / ox0 run table, img_pix0, bi, t0).
// Fold of the 16 Winograd products into the 2x2 outputs in ascending xi (exact +-1 coefficients), wave-private LDS transposes, BN scale /
// shift, residual (its loads issued before the fold of each output position), ReLU, 16-byte stores, optional GroupNorm partial sums.
    // --- epilogue: the wide kernel's (fold in ascending xi, wave-private LDS transposes, 16-byte stores)
    constexpr int ELD = 32 + 4;
    float* stage = smem + wave * (32 * ELD);
    constexpr int LPR = 8, RPI = 8, NP = 4;
    const int c4 = (lane & 7) * 4;
    const int n = n0 + wn0 + c4;
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
        const int tl = pass * RPI + lane / LPR;
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
    __syncthreads();                                   // every wave is done with the patch / V buffers
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // residual (BasicBlock identity): the four 16-byte loads of this output position go out BEFORE its fold (~1.3 k cycles of VALU
            // work) - in the pass loop each of them was a memory round trip of its own in front of a store (-3 % on the residual layers;
            // all sixteen up front spill)
            f32x4 rpre[NP];
            if (vec && p.res && n < p.Cout) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
                    const unsigned vm = vmask >> (3 * pass);
                    const bool ok = (vm & 1u) && (!a || (vm & 2u)) && (!b || (vm & 4u));
                    const long pix = pix0[pass] + a * step_a + b * step_b;
                    rpre[pass] = ok ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const float c = wino_fold_coef(2 * a + b, xi);
                if (c == 0.f) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = fmaf(acc[xi][r], c, o[r]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * ELD + frow] = o[r];
            __builtin_amdgcn_wave_barrier();
            if (n >= p.Cout) continue;
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int row = pass * RPI + lane / LPR;
                const unsigned vm = vmask >> (3 * pass);
                if (!(vm & 1u) || (a && !(vm & 2u)) || (b && !(vm & 4u))) continue;
                const long pix = pix0[pass] + a * step_a + b * step_b;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * ELD + c4);
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
                        const f32x4 rr = rpre[pass];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rr[e];
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
    if (p.gn_part && n < p.Cout) {   // fixed-order reduction over the 8 lanes that share a channel quad, then one writer lane
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gs[e] += __shfl_xor(gs[e], o);
                gq[e] += __shfl_xor(gq[e], o);
            }
        if (lane < LPR) {
            const long chunk = t0 / 32;
            double* o = p.gn_part + (((long)bi * (g.Tpad / 32) + chunk) * p.Cout + n) * 2;
            for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                o[2 * e] = (double)gs[e];
                o[2 * e + 1] = (double)gq[e];
            }
        }
    }

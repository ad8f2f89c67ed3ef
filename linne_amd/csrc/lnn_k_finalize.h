/* lnn_k_finalize.h -- k_finalize: best regulariser, quantiser, int32 FIR cascade.
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_FINALIZE_H_INCLUDED
#define LNN_K_FINALIZE_H_INCLUDED

/* ------------------------------------------------------------------------------------------------
 * finalize per channel-frame: best regulariser, quantisation, int32 FIR cascade
 * ---------------------------------------------------------------------------------------------- */
#define FIN_THREADS 256
__global__ __launch_bounds__(FIN_THREADS) void k_finalize(Plan p)
{
    __shared__ int32_t s_coef[LNN_MAXL][LNN_MAXP];
    __shared__ uint32_t s_rshift[LNN_MAXL], s_units[LNN_MAXL], s_best;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t n = c.n, S = p.S;
    const size_t ocf = (size_t)p.frame_map[cf / p.C] * p.C + cf % p.C;     /* the caller's channel-frame behind row cf of the class-sorted chunk */
    int32_t *rec = p.prm + ocf * LINNE_AMD_PARAM_WORDS;
    double *st = p.stats + ocf * LINNE_AMD_STAT_WORDS;

    if (tid == 0) {     /* linne_network.c:618-626 */
        double min_loss = (double)FLT_MAX; uint32_t best = 0;
        for (uint32_t r = 0; r < p.R; r++) { const double l = p.jloss[(size_t)cf * p.R + r]; if (l < min_loss) { min_loss = l; best = r; } }
        s_best = best;
        st[LINNE_AMD_ST_TAIL] = p.jtail[(size_t)cf * p.R + best];
        st[LINNE_AMD_ST_BEST] = p.af_best ? (double)p.af_best[cf] : (double)best;       /* -a N: this plan is the final pass of the winner */
        st[LINNE_AMD_ST_LOSS] = p.af_loss ? p.af_loss[cf] : p.jloss[(size_t)cf * p.R + best];
    }
    /* the record is written completely, whatever the caller's buffer held: words this preset does not use are zero */
    for (uint32_t i = LINNE_AMD_PRM_UNITS + tid; i < LINNE_AMD_PARAM_WORDS; i += FIN_THREADS) rec[i] = 0;
    __syncthreads();
    const uint32_t job = cf * p.R + s_best;
    if (tid < p.L) {    /* lpc.c:981-1040 over all units of the layer together */
        const uint32_t l = tid, P = p.P[l];
        const double *d = p.lparams + ((size_t)job * LNN_MAXL + l) * LNN_MAXP;
        double mx = 0.0;
        for (uint32_t k = 0; k < P; k++) if (mx < fabs(d[k])) mx = fabs(d[k]);
        uint32_t rshift;
        if (mx <= 0.0078125) {                              /* 2^-(8-1) */
            rshift = 8;
            for (uint32_t k = 0; k < P; k++) s_coef[l][k] = 0;
        } else {
            int ndigit; (void)frexp(mx, &ndigit);
            rshift = (uint32_t)(7 - ndigit);
            const double sc = ldexp(1.0, (int)rshift);       /* pow(2.0, rshift), exact */
            double qerr = 0.0;
            for (int32_t k = (int32_t)P - 1; k >= 0; k--) {
                qerr += d[k] * sc;
                int32_t q = (int32_t)round_away(qerr);
                if (q >= 128) q = 127; else if (q < -128) q = -128;
                qerr -= (double)q;
                s_coef[l][k] = q;
            }
        }
        s_rshift[l] = rshift;
        s_units[l] = p.lunits[(size_t)job * LNN_MAXL + l];
        rec[LINNE_AMD_PRM_UNITS + l] = (int32_t)s_units[l];
        rec[LINNE_AMD_PRM_RSHIFT + l] = (int32_t)rshift;
        for (uint32_t k = 0; k < P; k++) rec[LINNE_AMD_PRM_COEF + p.coef_off[l] + k] = s_coef[l][k];
    }
    __syncthreads();
    /* FIR cascade (linne_encoder.c:687-696, linne_lpc_predict.c:7-38) on the n valid samples; each layer streams the
     * channel through an LDS tile (1024 samples + 128 of history) so the tap loop reads LDS, not global memory */
    __shared__ int32_t xt[LNN_MAXP + 4 * FIN_THREADS];
    int32_t *src = p.xint + (size_t)cf * S, *dst = p.xtmp + (size_t)cf * S;
    for (uint32_t l = 0; l < p.L; l++) {
        const uint32_t units = s_units[l], np = p.P[l] / units, ns = n / units, rs = s_rshift[l];
        const uint32_t half = 1u << ((rs - 1u) & 31u);
        int32_t *out = (l + 1 == p.L) ? (p.resid + ocf * S) : dst;
        /* the tile's samples are requested one tile ahead (registers), so that a block does not sit out a trip to memory per tile */
        constexpr int NPF = (LNN_MAXP + 4 * FIN_THREADS + FIN_THREADS - 1) / FIN_THREADS;
        int32_t pf[NPF];
        auto prefetch = [&](uint32_t s0_) {
#pragma unroll
            for (int m = 0; m < NPF; m++) {
                const uint32_t i = tid + (uint32_t)m * FIN_THREADS;
                const int64_t g = (int64_t)s0_ - LNN_MAXP + i;
                pf[m] = (i < LNN_MAXP + 4 * FIN_THREADS && g >= 0 && g < (int64_t)n) ? src[g] : 0;
            }
        };
        prefetch(0);
        for (uint32_t s0 = 0; s0 < n; s0 += 4 * FIN_THREADS) {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < NPF; m++) { const uint32_t i = tid + (uint32_t)m * FIN_THREADS; if (i < LNN_MAXP + 4 * FIN_THREADS) xt[i] = pf[m]; }
            if (s0 + 4 * FIN_THREADS < n) prefetch(s0 + 4 * FIN_THREADS);
            __syncthreads();
            {   /* a lane owns 4 consecutive samples: when they sit in one unit past its first np samples (the usual case) the
                 * taps slide a 4-wide register window over the tile, one coefficient and one new sample per tap for four
                 * multiply-adds (int32 wrap-around: any order); otherwise sample by sample */
                const uint32_t e0 = 4 * tid, sb = s0 + e0;
                const uint32_t nsd = ns ? ns : 1u;
                const uint32_t unit0 = sb / nsd, loc0 = sb - unit0 * nsd;
                const bool quad = (sb + 3 < n) && (ns >= np) && (unit0 < units) && (loc0 >= np) && (loc0 + 3 < ns);
                if (quad) {
                    const int32_t *cc = s_coef[l] + unit0 * np;
                    const int32_t *xx = xt + LNN_MAXP + e0 - np;      /* -> x[sb - np] */
                    uint32_t p0 = half, p1 = half, p2 = half, p3 = half;
                    uint32_t w0 = (uint32_t)xx[0], w1 = (uint32_t)xx[1], w2 = (uint32_t)xx[2];
                    for (uint32_t k = 0; k < np; k++) {
                        const uint32_t ck = (uint32_t)cc[k], w3 = (uint32_t)xx[k + 3];
                        p0 += ck * w0; p1 += ck * w1; p2 += ck * w2; p3 += ck * w3;
                        w0 = w1; w1 = w2; w2 = w3;
                    }
                    const int32_t *xv = xt + LNN_MAXP + e0;
                    int4 o;
                    o.x = (int32_t)((uint32_t)xv[0] + (uint32_t)((int32_t)p0 >> (rs & 31u)));
                    o.y = (int32_t)((uint32_t)xv[1] + (uint32_t)((int32_t)p1 >> (rs & 31u)));
                    o.z = (int32_t)((uint32_t)xv[2] + (uint32_t)((int32_t)p2 >> (rs & 31u)));
                    o.w = (int32_t)((uint32_t)xv[3] + (uint32_t)((int32_t)p3 >> (rs & 31u)));
                    *(int4 *)(out + sb) = o;                          /* sb is a multiple of 4, rows are 16-byte aligned */
                } else {
                    for (uint32_t j = 0; j < 4; j++) {
                        const uint32_t e = e0 + j, s = s0 + e;
                        if (s >= n) continue;
                        int32_t v = xt[LNN_MAXP + e];
                        const uint32_t unit = s / nsd;
                        if (ns >= np && unit < units) {
                            const uint32_t loc = s - unit * ns;
                            if (loc >= np) {
                                uint32_t pred = half;
                                const int32_t *cc = s_coef[l] + unit * np;
                                const int32_t *xx = xt + LNN_MAXP + e - np;
                                for (uint32_t k = 0; k < np; k++) pred += (uint32_t)cc[k] * (uint32_t)xx[k];
                                v = (int32_t)((uint32_t)v + (uint32_t)((int32_t)pred >> (rs & 31u)));
                            }
                        }
                        out[s] = v;
                    }
                }
            }
        }
        if (l + 1 == p.L) for (uint32_t s = n + tid; s < S; s += FIN_THREADS) out[s] = 0;
        __syncthreads();
        if (l + 1 < p.L) { int32_t *t = src; src = dst; dst = t; }
    }
}


#endif

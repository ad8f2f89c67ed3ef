/* lnn_k_finalize.h -- k_quantize (best regulariser, quantiser, records) and k_fir_cascade (the int32 FIR cascade).
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_FINALIZE_H_INCLUDED
#define LNN_K_FINALIZE_H_INCLUDED

/* ------------------------------------------------------------------------------------------------
 * finalize per channel-frame: best regulariser, quantisation, int32 FIR cascade
 * ---------------------------------------------------------------------------------------------- */
#define FIN_THREADS 256
#define FIN_TILE (4 * FIN_THREADS)          /* output samples per block of the cascade */
#define FIN_HALO (LNN_MAXL * LNN_MAXP)      /* >= the taps of all layers together: what a tile recomputes in front of itself */

/* k_quantize, per channel-frame: the best regulariser (linne_encoder.c:618-626), the quantiser (lpc.c:981-1040 over all units of a
 * layer together), the parameter record and the statistics record.  One wave; lanes 0 .. L-1 quantise a layer each. */
__global__ __launch_bounds__(64) void k_quantize(Plan p)
{
    __shared__ uint32_t s_best;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
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
    uint32_t used = LINNE_AMD_PRM_COEF;
    for (uint32_t l = 0; l < p.L; l++) used += p.P[l];
    for (uint32_t i = LINNE_AMD_PRM_UNITS + tid; i < LINNE_AMD_PARAM_WORDS; i += 64u)
        if ((i >= LINNE_AMD_PRM_UNITS + p.L && i < LINNE_AMD_PRM_RSHIFT) || (i >= LINNE_AMD_PRM_RSHIFT + p.L && i < LINNE_AMD_PRM_COEF) || i >= used) rec[i] = 0;
    __syncthreads();
    const uint32_t job = cf * p.R + s_best;
    /* the winner's coefficients go to LDS with all lanes' loads in flight together: the quantiser below is a serial recurrence, and
     * read from global memory it paid a trip per coefficient (35 us for a block-at-a-time call) */
    __shared__ double s_d[LNN_MAXL][LNN_MAXP];
    for (uint32_t i = tid; i < p.L * LNN_MAXP; i += 64u) { const uint32_t l = i / LNN_MAXP, k = i % LNN_MAXP; if (k < p.P[l]) s_d[l][k] = p.lparams[((size_t)job * LNN_MAXL + l) * LNN_MAXP + k]; }
    __syncthreads();
    if (tid < p.L) {    /* lpc.c:981-1040 over all units of the layer together */
        const uint32_t l = tid, P = p.P[l];
        const double *d = s_d[l];
        int32_t *cq = rec + LINNE_AMD_PRM_COEF + p.coef_off[l];
        double mx = 0.0;
        for (uint32_t k = 0; k < P; k++) if (mx < fabs(d[k])) mx = fabs(d[k]);
        uint32_t rshift;
        if (mx <= 0.0078125) {                              /* 2^-(8-1) */
            rshift = 8;
            for (uint32_t k = 0; k < P; k++) cq[k] = 0;
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
                cq[k] = q;
            }
        }
        rec[LINNE_AMD_PRM_UNITS + l] = (int32_t)p.lunits[(size_t)job * LNN_MAXL + l];
        rec[LINNE_AMD_PRM_RSHIFT + l] = (int32_t)rshift;
    }
}

/* k_fir_cascade: the int32 FIR cascade (linne_encoder.c:687-696, linne_lpc_predict.c:7-38) of a channel-frame, a block per tile of
 * 1024 outputs.  The cascade has no recurrence -- layer l's output at s is a function of layer l - 1's outputs at s - np .. s of the
 * same unit -- so a tile recomputes what it needs of the layers in front of it: the input with sum P taps of history goes into LDS
 * once, every layer works LDS to LDS (its output range shrinks by its own taps), and only the residual is written.  Traffic: the
 * channel in (plus 14 % of halo), the residual out -- the form that streamed every layer through xint / xtmp moved 250 KB per
 * channel-frame and kept a channel-frame in ONE block (two blocks busy on a 256-CU chip for a stereo block-at-a-time call).
 *
 * Layers of >= 16 taps per unit run on the MATRIX unit.  The coefficients are 8-bit (the quantiser's range), the samples are cut
 * into four signed base-256 digits (sp_digits), and sum_k c[k] x[s - np + k] for 64 consecutive outputs is one Toeplitz product
 * v_mfma_i32_16x16x64_i8 per 64 window samples: A = the digit planes of four 16-output groups' windows (rows 4 q + b), B = the
 * unit's coefficients as a Toeplitz band (the same for every group: it depends on relative positions only), C[4 q + b][i] = plane b
 * of output i of group q -- which is exactly the lane (i, q)'s four accumulator registers.  The planes recombine by shifts modulo
 * 2^32, what the reference's wrap-around int32 sum holds.  148 multiply-adds per sample on v_mul_lo_u32 (the cascade was bound by
 * them: 4 ms per 31 008 channel-frames) become 0.3 instructions per output and layer.  Layers of fewer taps, chunks that straddle
 * a unit boundary or the frame's ends, take the integer multiplier as before. */
#define FIN_FRONT 192u                       /* buffer entries in front of the history: the matrix windows reach back 64 KS - 16 samples from a chunk */
#define FIN_BUF (FIN_FRONT + ((FIN_HALO + 63u) & ~63u) + FIN_TILE)
__global__ __launch_bounds__(FIN_THREADS, 4) void k_fir_cascade(Plan p)
{
    __shared__ __attribute__((aligned(16))) int32_t s_coef[LNN_MAXL][LNN_MAXP];
    __shared__ __attribute__((aligned(16))) int32_t bufs[2][FIN_BUF];
    __shared__ __attribute__((aligned(16))) int8_t dig[2][4][FIN_BUF];          /* digit planes of a layer's input, when that layer takes the matrix path */
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t n = c.n, S = p.S, L = p.L;
    const size_t ocf = (size_t)p.frame_map[cf / p.C] * p.C + cf % p.C;
    const int32_t *rec = p.prm + ocf * LINNE_AMD_PARAM_WORDS;
    int32_t *out = p.resid + ocf * S;
    uint32_t H = 0;
    for (uint32_t l = 0; l < L; l++) H += p.P[l];            /* taps of all layers: the history a tile needs (<= FIN_HALO) */
    /* the layers' unit counts and shifts go to LDS once (read from the record where they are needed they cost every layer a trip to memory) */
    __shared__ uint32_t l_units[LNN_MAXL], l_rs[LNN_MAXL], l_matrix[LNN_MAXL + 1], s_toep_unit[LNN_MAXL];
    __shared__ lnn_v4i s_toep[LNN_MAXL][3][64];
    if (tid <= LNN_MAXL) {
        const uint32_t l = tid;
        if (l < LNN_MAXL) s_toep_unit[l] = 0xFFFFFFFFu;
        uint32_t m = 0;
        if (l < L) {
            const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + l];
            l_units[l] = units; l_rs[l] = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + l];
            /* which layers take the matrix path (never layer 0: its input has no digit planes) */
            if (l > 0 && units != 0 && p.P[l] <= 128u) { const uint32_t np = p.P[l] / units; m = (np >= 16u && n / units >= np) ? 1u : 0u; }
        }
        l_matrix[l] = m;
    }
    const uint32_t ORG = FIN_FRONT + ((H + 63u) & ~63u);      /* buffer index of a tile's first sample s0: a multiple of 64 */
    const int32_t *src = p.xint + (size_t)cf * S;
    /* A block takes the tiles blockIdx.y, blockIdx.y + gridDim.y, ... of its channel-frame (batches: gridDim.y = 1, the block walks the
     * channel-frame and pays its start-up -- class, map and parameter loads, each a dependent trip to memory -- once; block-at-a-time
     * calls: a tile per block).  The next tile's samples are requested before this one is worked on. */
    const uint32_t ntile = (S + FIN_TILE - 1u) / FIN_TILE;
    constexpr int NPF = (FIN_HALO + FIN_TILE + FIN_THREADS - 1) / FIN_THREADS;
    int32_t pf[NPF];
    auto fetch = [&](uint32_t s0_) {                          /* buffer index i <-> sample s0 - ORG + i; loaded: [ORG - H, ORG + FIN_TILE) */
#pragma unroll
        for (int m = 0; m < NPF; m++) {
            const uint32_t i = tid + (uint32_t)m * FIN_THREADS;
            const int64_t g = (int64_t)s0_ - H + i;
            pf[m] = (i < H + FIN_TILE && g >= 0 && g < (int64_t)n) ? src[g] : 0;
        }
    };
    if (blockIdx.y < ntile && blockIdx.y * FIN_TILE < n) fetch(blockIdx.y * FIN_TILE);
    for (uint32_t i = tid; i < L * LNN_MAXP; i += FIN_THREADS) { const uint32_t l = i / LNN_MAXP, k = i % LNN_MAXP; if (k < p.P[l]) s_coef[l][k] = rec[LINNE_AMD_PRM_COEF + p.coef_off[l] + k]; }
    for (uint32_t tile = blockIdx.y; tile < ntile; tile += gridDim.y) {
    const uint32_t s0 = tile * FIN_TILE;
    if (s0 >= n) {                                            /* behind the frame's end: zeros (linne_encoder.c:613-621 padded the input) */
        for (uint32_t i = tid; i < FIN_TILE; i += FIN_THREADS) if (s0 + i < S) out[s0 + i] = 0;
        continue;
    }
    __syncthreads();                                          /* the tile before is through with the buffers */
#pragma unroll
    for (int m = 0; m < NPF; m++) { const uint32_t i = tid + (uint32_t)m * FIN_THREADS; if (i < H + FIN_TILE) bufs[0][ORG - H + i] = pf[m]; }
    { const uint32_t tn = tile + gridDim.y; if (tn < ntile && tn * FIN_TILE < n) fetch(tn * FIN_TILE); }
    __syncthreads();
    uint32_t lead = H;                                        /* samples in front of s0 that the current input buffer holds valid */
#pragma unroll 1
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t units = l_units[l], rs = l_rs[l];
        const uint32_t np = p.P[l] / (units ? units : 1u), ns = units ? n / units : 0u;
        const uint32_t half = 1u << ((rs - 1u) & 31u);
        const bool last = (l + 1 == L), digits_out = l_matrix[l + 1] != 0;
        const int32_t *in = bufs[l & 1u];
        int32_t *ob = bufs[(l & 1u) ^ 1u];
        int8_t (*dgo)[FIN_BUF] = dig[(l & 1u) ^ 1u];
        const uint32_t olead = lead - p.P[l];                 /* this layer's outputs start olead samples in front of s0 */
        const uint32_t count = olead + FIN_TILE;
        const uint32_t nsd = ns ? ns : 1u;
        /* one output by the positional rule (the general form: any unit layout, the frame's ends) */
        auto scalar_out = [&](uint32_t idx) -> int32_t {
            const int64_t s64 = (int64_t)s0 - ORG + idx;
            if (s64 < 0 || s64 >= (int64_t)n) return 0;
            const uint32_t s = (uint32_t)s64, unit = s / nsd;
            int32_t v = in[idx];
            if (units && ns >= np && unit < units) {
                const uint32_t loc = s - unit * ns;
                if (loc >= np) {
                    uint32_t pred = half;
                    const int32_t *cc = s_coef[l] + unit * np;
                    const int32_t *xx = in + idx - np;
                    for (uint32_t k = 0; k < np; k++) pred += (uint32_t)cc[k] * (uint32_t)xx[k];
                    v = (int32_t)((uint32_t)v + (uint32_t)((int32_t)pred >> (rs & 31u)));
                }
            }
            return v;
        };
        auto put_digits = [&](uint32_t idx, int32_t v) {
            const uint32_t dg = sp_digits(v);
            dgo[0][idx] = (int8_t)dg; dgo[1][idx] = (int8_t)(dg >> 8); dgo[2][idx] = (int8_t)(dg >> 16); dgo[3][idx] = (int8_t)(dg >> 24);
        };
        if (l_matrix[l] != 0) {
            /* chunks of 64 outputs at buffer indices that are multiples of 64; wave w takes the chunks w, w + 4, ... */
            const uint32_t KS = (np + 15u + 63u) / 64u;                         /* 64-sample steps of a chunk's windows (1 .. 3) */
            const int8_t (*dgi)[FIN_BUF] = dig[l & 1u];
            const uint32_t first = (ORG - olead) & ~63u, nchunk = (ORG + FIN_TILE - first) / 64u;
            const uint32_t i = lane & 15u, g4 = lane >> 4, qa = (lane & 15u) >> 2, ba = lane & 3u;
            /* the Toeplitz band of a unit: window element m of a group meets tap m + 16 - 64 KS - i + np of output i */
            auto build_band = [&](uint32_t unit, lnn_v4i *frag) {
                const int32_t *cc = s_coef[l] + unit * np;
#pragma unroll
                for (uint32_t st = 0; st < 3u; st++) {
                    uint32_t w[4] = { 0u, 0u, 0u, 0u };
                    if (st < KS) {
#pragma unroll
                        for (uint32_t e = 0; e < 16u; e++) {
                            const int32_t kx = (int32_t)(64u * st + 16u * g4 + e + 16u + np) - (int32_t)(64u * KS + i);
                            const int32_t cv = (kx >= 0 && (uint32_t)kx < np) ? cc[kx] : 0;
                            w[e >> 2] |= ((uint32_t)cv & 0xFFu) << (8u * (e & 3u));
                        }
                    }
                    frag[st][0] = (int)w[0]; frag[st][1] = (int)w[1]; frag[st][2] = (int)w[2]; frag[st][3] = (int)w[3];
                }
            };
            /* The band of the unit the tile starts in lives in LDS, shared by the waves, and stays there from tile to tile (a block
             * that walks its channel-frame builds it once per unit: building it costs as many instructions as five chunks of
             * outputs).  A chunk in another unit -- a tile with a unit boundary in it -- builds its own in registers. */
            const uint32_t u0 = ((s0 > olead) ? (s0 - olead) / nsd : 0u) < units ? ((s0 > olead) ? (s0 - olead) / nsd : 0u) : units - 1u;
            {
                const uint32_t held = s_toep_unit[l];
                __syncthreads();
                if (held != u0) {
                    if (wave == 0) { lnn_v4i fr[3]; build_band(u0, fr); s_toep[l][0][lane] = fr[0]; s_toep[l][1][lane] = fr[1]; s_toep[l][2][lane] = fr[2]; }
                    if (tid == 0) s_toep_unit[l] = u0;
                }
                __syncthreads();
            }
            lnn_v4i bfrag[3];
            uint32_t have_unit = 0xFFFFFFFFu;
            for (uint32_t ck = wave; ck < nchunk; ck += FIN_THREADS / 64u) {
                const uint32_t bc = first + 64u * ck;
                const int64_t sa = (int64_t)s0 - ORG + bc, sz = sa + 63;         /* first and last sample of the chunk */
                const uint32_t idx = bc + 16u * g4 + i;                           /* my output */
                int32_t o;
                bool mat = sa >= 0 && sz < (int64_t)n;
                uint32_t unit = 0;
                if (mat) { unit = (uint32_t)sa / nsd; mat = unit < units && (uint32_t)sz / nsd == unit; }
                if (mat) {
                    if (unit != have_unit) {
                        if (unit == u0) { bfrag[0] = s_toep[l][0][lane]; bfrag[1] = s_toep[l][1][lane]; bfrag[2] = s_toep[l][2][lane]; }
                        else build_band(unit, bfrag);
                        have_unit = unit;
                    }
                    lnn_v4i acc4 = { 0, 0, 0, 0 };
                    const uint32_t w0 = bc + 16u * qa + 16u - 64u * KS + 16u * g4;   /* my 16 bytes of group qa's window, plane ba, step 0 */
#pragma unroll
                    for (uint32_t st = 0; st < 3u; st++) {
                        if (st < KS) {
                            const lnn_v4i a = *(const lnn_v4i *)&dgi[ba][w0 + 64u * st];
                            acc4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bfrag[st], acc4, 0, 0, 0);
                        }
                    }
                    const uint32_t sum = (uint32_t)acc4[0] + ((uint32_t)acc4[1] << 8) + ((uint32_t)acc4[2] << 16) + ((uint32_t)acc4[3] << 24);
                    const uint32_t s = (uint32_t)((int64_t)s0 - ORG + idx), loc = s - unit * ns;
                    const int32_t v = in[idx];
                    o = (loc >= np) ? (int32_t)((uint32_t)v + (uint32_t)((int32_t)(half + sum) >> (rs & 31u))) : v;
                } else o = scalar_out(idx);
                if (last) { const int64_t s64 = (int64_t)s0 - ORG + idx; if (s64 >= (int64_t)s0 && s64 < (int64_t)S) out[s64] = o; }
                else { ob[idx] = o; if (digits_out) put_digits(idx, o); }
            }
        } else {
        /* a lane owns 4 consecutive samples: when they sit in one unit past its first np samples (the usual case) the taps slide a
         * 4-wide register window over the buffer, one coefficient and one new sample per tap for four multiply-adds (int32
         * wrap-around: any order); otherwise sample by sample */
        for (uint32_t e0 = 4u * tid; e0 < count; e0 += 4u * FIN_THREADS) {
            const int64_t sb64 = (int64_t)s0 - olead + e0;    /* first of my 4 samples (may lie before sample 0 in the first tile) */
            const uint32_t bi = ORG - olead + e0;             /* its buffer index */
            int32_t o[4];
            bool quad = false;
            if (sb64 >= 0 && sb64 + 3 < (int64_t)n && units && ns >= np) {
                const uint32_t sb = (uint32_t)sb64, unit0 = sb / nsd, loc0 = sb - unit0 * nsd;
                quad = (unit0 < units) && (loc0 >= np) && (loc0 + 3 < ns);
                if (quad) {
                    const int32_t *cc = s_coef[l] + unit0 * np;
                    const int32_t *xx = in + bi - np;             /* -> x[sb - np] */
                    uint32_t p0 = half, p1 = half, p2 = half, p3 = half;
                    if ((np & 3u) == 0 && ((bi - np) & 3u) == 0) {
                        /* four taps a trip: one 16-byte read of samples, one of coefficients (a broadcast) for 16 multiply-adds */
                        const int4 *xx4 = (const int4 *)xx, *cc4 = (const int4 *)cc;
                        int4 xa = xx4[0];
                        for (uint32_t k4 = 0; k4 < (np >> 2); k4++) {
                            const int4 xb = xx4[k4 + 1], c4 = cc4[k4];
                            const uint32_t v0 = (uint32_t)xa.x, v1 = (uint32_t)xa.y, v2 = (uint32_t)xa.z, v3 = (uint32_t)xa.w, v4 = (uint32_t)xb.x, v5 = (uint32_t)xb.y, v6 = (uint32_t)xb.z;
                            const uint32_t c0 = (uint32_t)c4.x, c1 = (uint32_t)c4.y, c2 = (uint32_t)c4.z, c3 = (uint32_t)c4.w;
                            p0 += c0 * v0 + c1 * v1 + c2 * v2 + c3 * v3;
                            p1 += c0 * v1 + c1 * v2 + c2 * v3 + c3 * v4;
                            p2 += c0 * v2 + c1 * v3 + c2 * v4 + c3 * v5;
                            p3 += c0 * v3 + c1 * v4 + c2 * v5 + c3 * v6;
                            xa = xb;
                        }
                    } else {
                        uint32_t w0 = (uint32_t)xx[0], w1 = (uint32_t)xx[1], w2 = (uint32_t)xx[2];
                        for (uint32_t k = 0; k < np; k++) {
                            const uint32_t ck = (uint32_t)cc[k], w3 = (uint32_t)xx[k + 3];
                            p0 += ck * w0; p1 += ck * w1; p2 += ck * w2; p3 += ck * w3;
                            w0 = w1; w1 = w2; w2 = w3;
                        }
                    }
                    const int32_t *xv = in + bi;
                    o[0] = (int32_t)((uint32_t)xv[0] + (uint32_t)((int32_t)p0 >> (rs & 31u)));
                    o[1] = (int32_t)((uint32_t)xv[1] + (uint32_t)((int32_t)p1 >> (rs & 31u)));
                    o[2] = (int32_t)((uint32_t)xv[2] + (uint32_t)((int32_t)p2 >> (rs & 31u)));
                    o[3] = (int32_t)((uint32_t)xv[3] + (uint32_t)((int32_t)p3 >> (rs & 31u)));
                }
            }
            if (!quad) { for (uint32_t j = 0; j < 4; j++) o[j] = (e0 + j < count) ? scalar_out(bi + j) : 0; }
            if (last) {                                       /* olead = 0: e0 is the offset inside the tile */
                const uint32_t sb = s0 + e0;
                if (sb + 3 < S && (S & 3u) == 0) { int4 q; q.x = o[0]; q.y = o[1]; q.z = o[2]; q.w = o[3]; *(int4 *)(out + sb) = q; }      /* (samples behind n are zero) */
                else for (uint32_t j = 0; j < 4; j++) if (sb + j < S) out[sb + j] = o[j];
            } else {
                for (uint32_t j = 0; j < 4; j++) if (e0 + j < count) { ob[bi + j] = o[j]; if (digits_out) put_digits(bi + j, o[j]); }
            }
        }
        }
        lead = olead;
        __syncthreads();
    }
    }
}


#endif

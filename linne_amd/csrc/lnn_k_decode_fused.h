/* lnn_k_decode_fused.h -- the END of the synthesis cascade in one launch: layer 0 (linne_lpc_synthesize.c:8-83), the two-stage
 * de-emphasis (linne_utility.c:215-241) and MS -> LR (linne_utility.c:135-147), the channel-frame crossing HBM once where
 * k_synth_rows8 + k_deemph_lr streamed it in and out twice (k_deemph_lr alone ran at the HBM's pace: 0.52 ms per 31 008
 * channel-frames for 8 instructions a sample).
 * Part of the single translation unit lnn_device.hip (included there behind lnn_k_decode_rows.h); not a stand-alone header.
 *
 * A block owns 64 channel-frames and walks them in tiles of 64 samples that live in LDS, four tiles in flight, one barrier per tile:
 *   waves 9, 10   load tile t (16-byte groups, 16 B per lane: 8 instructions per wave and tile) and store tile t - 3 (MS -> LR on the
 *                 way out when FUSE_MS: the frame's channels 0 and 1 are neighbouring rows of the block);
 *   waves 0 .. 7  layer 0 on tile t - 1, IN PLACE: k_synth_rows8's recurrence, eight channel-frames per wave on the halves of the DPP
 *                 rows, blocks of 8 outputs (shift, subtract, two broadcasts, the multiply-adds into this block's sum and the next's);
 *   wave 8        the de-emphasis on tile t - 2, in place: lanes = channel-frames, a scalar recurrence each (its state in registers).
 * Tile layout: channel-frame r = 8 w + q of the block (w: its producer wave, q: its place in that wave) sits in tile row
 * rho = 8 q + w, rows 65 words apart: bank = rho + sample.  The producers' lanes (q, i) at sample 8 k + i hit banks 8 q + w + 8 k + i
 * (all 64 distinct), the de-emphasis lanes (lane = rho, one sample) banks rho + s (distinct): neither form conflicts.
 * Frames of any length and unit count take this kernel (a lane is `pred` or not, as in k_synth_rows8); what lies behind a frame's
 * end is loaded, worked on and never stored. */
#ifndef LNN_K_DECODE_FUSED_H_INCLUDED
#define LNN_K_DECODE_FUSED_H_INCLUDED

#define SF_PROD 8                          /* producer waves (layer 0): eight channel-frames each */
#define SF_WAVES (SF_PROD + 3)             /* + the de-emphasis wave + two waves that load and store */
#define SF_STRIDE 65u                      /* words from a tile row to the next */
#define SF_TILE (64u * SF_STRIDE)
#define SF_STEPS SF_STEP(0) SF_STEP(1) SF_STEP(2) SF_STEP(3) SF_STEP(4) SF_STEP(5) SF_STEP(6) SF_STEP(7)
__device__ __forceinline__ uint32_t sf_rho(uint32_t r) { return 8u * (r & 7u) + (r >> 3); }      /* tile row of the block's channel-frame r (its own inverse) */

template <bool FUSE_MS>
__global__ __launch_bounds__(64 * SF_WAVES, 6) void k_synth_l0_de(DecPlan p)      /* (two blocks per CU: 66.6 KB of LDS each, 80 registers) */
{
    __shared__ int32_t tile[4][SF_TILE];
    __shared__ int8_t call[64][16];                                /* layer 0's coefficients of each channel-frame (unit u's at u np) */
    __shared__ uint32_t nlen[64];                                  /* the channel-frames' lengths (what the storing waves may write) */
    const uint32_t lane = threadIdx.x & 63u, S = p.S, C = p.C;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t row0 = blockIdx.x * 64u, nrows = p.F * C;
    const uint32_t nv = (nrows - row0 < 64u) ? nrows - row0 : 64u;
    /* tiles of the block: the longest of its channel-frames (every wave computes it: the barriers below are counted by it) */
    uint32_t ntiles;
    {
        uint32_t rl = row0 + lane; if (rl >= nrows) rl = nrows - 1u;
        uint32_t nmax = p.nsmp[rl / C];
        if (wave == (uint32_t)SF_PROD) nlen[lane] = nmax;          /* (read behind the first barrier) */
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)nmax, o); nmax = other > nmax ? other : nmax; }
        ntiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)((nmax + 63u) / 64u));
    }
    if (wave < (uint32_t)SF_PROD) {
        /* ---- layer 0: k_synth_rows8<4> on tiles in LDS ---- */
        constexpr int PB = 4, JB = 8 - PB;
        const uint32_t i = lane & 7u, q = lane >> 3, r = 8u * wave + q;
        uint32_t cf = row0 + r;
        const bool have = cf < nrows;
        if (!have) cf = nrows - 1u;
        const uint32_t n = have ? p.nsmp[cf / C] : 0u;
        const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
        const uint32_t P = p.P[0];
        const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + 0], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + 0];
        const uint32_t np = units ? P / units : 0u, ns = units ? n / units : 0u;
        const bool skip = (units == 0 || np == 0 || ns < np);      /* linne_decoder.c: such a layer leaves the data unchanged */
        const uint32_t half = 1u << ((rs - 1u) & 31u), sh_ = rs & 31u;
        const int32_t *crec = rec + LINNE_AMD_PRM_COEF + p.coef_off[0];
        call[r][i] = (i < P && units) ? (int8_t)crec[i] : (int8_t)0;
        call[r][i + 8u] = 0;
        int32_t *const trow = &tile[0][0] + sf_rho(r) * SF_STRIDE + i;      /* my sample of block k of tile b: trow[b SF_TILE + 8 k] */
        uint32_t unit = 0, m_event = 0u, half_l = 0u;
        bool pred = false;
        int32_t ccA[8], ccB[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { ccA[j] = 0; ccB[j] = 0; }
        int32_t yprev = 0;
        uint32_t accB = 0;                                         /* what the previous block adds to this block's sums */
        const uint32_t nblk = 8u * ntiles;
        __syncthreads();                                           /* iteration 0: tile 0 arrives */
        uint32_t m = 0;
#pragma unroll 1
        while (m < nblk) {
            {   /* a lane's class changes here: every lane's class in this block, for how many blocks it keeps it, its registers */
                const uint32_t t = 8u * m + i;
                unit = skip ? units : t / (ns ? ns : 1u);
                uint32_t ahead = 0xFFFFFFFFu;                      /* behind the last unit (or a skipped layer): copied to the end */
                pred = false;
                if (unit < units) {
                    const uint32_t tl = t - unit * ns;
                    pred = tl >= np;
                    ahead = ((pred ? ns : np) - tl + 7u) / 8u;
                }
                m_event = (ahead == 0xFFFFFFFFu) ? ahead : m + ahead;
                const uint32_t cb0 = unit * np + np - i;           /* tap of distance d of my unit: call[cb0 + i - d]; all of them lie in the unit */
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int32_t da = (int32_t)i - j; const uint32_t db = 8u + i - (uint32_t)j;
                    const bool va = pred && da >= 1 && (uint32_t)da <= np, vb = pred && db <= np;
                    ccA[j] = va ? (int32_t)call[r][va ? cb0 + (uint32_t)j : 0u] : 0;
                    ccB[j] = vb ? (int32_t)call[r][vb ? cb0 + (uint32_t)j - 8u : 0u] : 0;
                }
                half_l = pred ? half : 0u;
                accB = 0;                                          /* with the registers as they are now (lanes that kept their class get the value they had) */
#define SF_STEP(J) if ((J) >= JB) { const int32_t sv = half_bcast<J>(yprev); accB += sp_mul8(ccB[J], sv & 0xFFFF, sv >> 16); }
                SF_STEPS
#undef SF_STEP
            }
#pragma unroll 1
            do {
                int32_t *const cell = trow + ((m >> 3) & 3u) * SF_TILE + 8u * (m & 7u);
                const int32_t res = *cell;
                const uint32_t acc0 = half_l + accB;
                uint32_t acc = acc0, nb = 0;
                /* speculation as in k_synth_rows8: every output of the block fits 24 bits */
#define SF_STEP(J) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); const int32_t sv = half_bcast<J>(y); \
                acc += (uint32_t)__mul24(ccA[J], sv); \
                if ((J) >= JB) asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(nb) : "v"(ccB[J]), "v"(sv)); }
                SF_STEPS
#undef SF_STEP
                int32_t yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
                const bool fits = (((int32_t)((uint32_t)yout << 8) >> 8) == yout);
                if (!__all(fits)) {
                    acc = acc0; nb = 0;
#define SF_STEP(J) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); const int32_t sv = half_bcast<J>(y); \
                const int32_t sl = sv & 0xFFFF, shh = sv >> 16; acc += sp_mul8(ccA[J], sl, shh); if ((J) >= JB) nb += sp_mul8(ccB[J], sl, shh); }
                    SF_STEPS
#undef SF_STEP
                    yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
                }
                *cell = yout;
                yprev = yout;
                accB = nb;
                m++;
                if ((m & 7u) == 0u) __syncthreads();               /* the tile is through layer 0 (iteration m / 8 ends) */
            } while (m < nblk && __all(m < m_event));
        }
        __syncthreads(); __syncthreads();                          /* iterations ntiles + 1, ntiles + 2: the last tiles leave */
        return;
    }
    if (wave == (uint32_t)SF_PROD) {
        /* ---- de-emphasis: lane = tile row rho, i.e. the block's channel-frame sf_rho(lane) ---- */
        uint32_t cf = row0 + sf_rho(lane);
        if (cf >= nrows) cf = nrows - 1u;
        const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
        const int32_t c0e = rec[LINNE_AMD_PRM_PCOEF + 0], c1e = rec[LINNE_AMD_PRM_PCOEF + 1];
        int32_t zp = rec[LINNE_AMD_PRM_PREV + 1], yp = rec[LINNE_AMD_PRM_PREV + 0];
        int32_t *const trow = &tile[0][0] + lane * SF_STRIDE;
        /* the one serial piece of a tile -- 64 dependent steps -- sets the block's pace if it has to queue for issue slots behind the
         * eight producer waves: it goes first */
        __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (uint32_t t = 0; t < ntiles + 3u; t++) {
            if (t >= 2u && t - 2u < ntiles) {
                int32_t *tl = trow + ((t - 2u) & 3u) * SF_TILE;
#pragma unroll 16
                for (uint32_t s = 0; s < 64u; s++) {               /* (behind a frame's end the state runs on: nothing of it is stored) */
                    const int32_t z = (int32_t)((uint32_t)tl[s] + (uint32_t)mulshr5(zp, c1e));
                    const int32_t y = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                    zp = z; yp = y;
                    tl[s] = y;
                }
            }
            __syncthreads();
        }
        return;
    }
    {
        /* ---- load and store: an instruction moves 16 bytes per lane = 64 samples of four channel-frames.  Instruction kk of wave
         * 9 + h takes the block's channel-frames r = 8 (4 h + lane / 16) + kk: their tile rows rho = 8 kk + 4 h + lane / 16 lie 8 rows
         * apart from one instruction to the next (constant offsets from one address register), a lane's four samples of the four
         * rows hit 64 different banks, and a frame's channels 0 and 1 (r and r ^ 1) are the SAME lane's instructions kk and kk ^ 1:
         * MS -> LR needs no second read ---- */
        const uint32_t h = wave - (uint32_t)SF_PROD - 1u, rq = lane >> 4, i4 = 4u * (lane & 15u);
        const uint32_t rbase = 8u * (4u * h + rq);                 /* my rows: rbase + kk */
        int32_t *blk = p.data + (size_t)row0 * S;
        int32_t *const tl0 = &tile[0][0] + (4u * h + rq) * SF_STRIDE + i4;      /* row kk of tile b: tl0[b SF_TILE + 8 kk SF_STRIDE] */
        __builtin_amdgcn_s_setprio(2);                             /* (the next tile's loads should be on their way early) */
        lnn_v4i pre[8];
        auto issue = [&](uint32_t t) {
            const uint32_t s0 = t * 64u + i4;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const uint32_t r = rbase + (uint32_t)kk, rr = (r < nv) ? r : nv - 1u;
                pre[kk] = (s0 < S) ? *(const lnn_v4i *)(blk + (rr * S + s0)) : lnn_v4i{ 0, 0, 0, 0 };      /* (S is a multiple of 4; a 32-bit offset from the block's uniform base) */
            }
        };
        auto commit = [&](uint32_t t) {
            int32_t *tb = tl0 + (t & 3u) * SF_TILE;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                int32_t *w = tb + 8u * (uint32_t)kk * SF_STRIDE;
                w[0] = pre[kk][0]; w[1] = pre[kk][1]; w[2] = pre[kk][2]; w[3] = pre[kk][3];
            }
        };
        /* FUSE_MS: channel of row rbase + kk (C is a power of two; row0 a multiple of 64 >= C) */
        const bool ms_rows = FUSE_MS && ((rbase & (C - 1u) & ~7u) == 0u);       /* my rows hold channels 0 .. 7 of a frame (C <= 8: always) */
        if (ntiles) issue(0);
#pragma unroll 1
        for (uint32_t t = 0; t < ntiles + 3u; t++) {
            if (t >= 3u) {                   /* tile t - 3 leaves */
                const uint32_t to = t - 3u, s0 = to * 64u + i4;
                const int32_t *tb = tl0 + (to & 3u) * SF_TILE;
#pragma unroll
                for (int kp = 0; kp < 4; kp++) {                   /* rows 2 kp and 2 kp + 1 */
                    const int32_t *wa = tb + 8u * (uint32_t)(2 * kp) * SF_STRIDE, *wb = wa + 8u * SF_STRIDE;
                    lnn_v4i va = { wa[0], wa[1], wa[2], wa[3] }, vb = { wb[0], wb[1], wb[2], wb[3] };
                    if (FUSE_MS && ms_rows && ((uint32_t)(2 * kp) & (C - 1u)) == 0u) {      /* (va, vb) = (mid, side) of a frame: linne_utility.c:135-147 */
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t l = (uint32_t)va[j] - (uint32_t)(vb[j] >> 1);
                            va[j] = (int32_t)l; vb[j] = (int32_t)((uint32_t)vb[j] + l);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const uint32_t r = rbase + (uint32_t)(2 * kp + e);
                        const lnn_v4i v = e ? vb : va;
                        if (r < nv) {
                            const uint32_t nr = nlen[r];
                            int32_t *gp = blk + (r * S + s0);
                            if (s0 + 3u < nr) *(lnn_v4i *)gp = v;
                            else { if (s0 < nr) gp[0] = v[0]; if (s0 + 1u < nr) gp[1] = v[1]; if (s0 + 2u < nr) gp[2] = v[2]; }
                        }
                    }
                }
            }
            if (t < ntiles) { commit(t); if (t + 1u < ntiles) issue(t + 1u); }
            __syncthreads();
        }
    }
}
#undef SF_STEPS

#endif

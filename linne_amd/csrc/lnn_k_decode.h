/* lnn_k_decode.h -- decode kernels: k_synth_small, k_synth_big, k_synthesize, k_ms_to_lr.
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_DECODE_H_INCLUDED
#define LNN_K_DECODE_H_INCLUDED

/* ------------------------------------------------------------------------------------------------
 * decode: synthesis cascade + de-emphasis per channel-frame (one wavefront), MS->LR per frame
 * ---------------------------------------------------------------------------------------------- */
struct DecPlan {
    uint32_t C, S, L, ms, F;
    uint32_t P[LNN_MAXL], coef_off[LNN_MAXL];
    int32_t *data; const int32_t *prm; const uint32_t *nsmp;
};

/* wrap-around sum of one int per lane over the 64-lane wavefront (associative, so a DPP tree is exact) */
__device__ __forceinline__ int32_t wave_sum_i32(int32_t v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   /* row_shr:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   /* row_bcast:15 -> rows 1,3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   /* row_bcast:31 -> rows 2,3 */
    return __builtin_amdgcn_readlane(v, 63);
}

/* One wavefront per channel-frame, data streamed through registers in 64-sample chunks (coalesced loads/stores):
 *   - the 128-slot history ring of the recurrence lives in two registers per lane (slot = sample index mod 128);
 *     the chunk being reconstructed IS one of them, so finished samples are already where the next steps need them
 *     and the chunk is stored from that register;
 *   - taps are spread over the lanes, the rotating zero-extended coefficient ring sits in 512 B of LDS, the int32
 *     dot product is reduced with the DPP tree (wrap-around addition is associative);
 *   - the residual of the step is picked from the loaded chunk with a scalar readlane; the next chunk's load is in
 *     flight meanwhile.
 * No per-channel LDS staging, so occupancy is limited by registers only. */
__global__ __launch_bounds__(64) void k_synthesize(DecPlan p, uint32_t only_layer, uint32_t deemph)
{
    __shared__ int32_t cpad[128];
    const uint32_t cf = blockIdx.x, lane = threadIdx.x;
    const uint32_t n = p.nsmp[cf / p.C], S = p.S;
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    int32_t *g = p.data + (size_t)cf * S;
    /* linne_decoder.c:503-509: layers in reverse order; linne_lpc_synthesize.c:8-83: units are independent */
    for (int32_t l = (int32_t)p.L - 1; l >= 0; l--) {
        if (only_layer != 0xFFFFFFFFu && (uint32_t)l != only_layer) continue;
        const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + l], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + l];
        const uint32_t np = p.P[l] / (units ? units : 1u), ns = n / (units ? units : 1u);
        const uint32_t half = 1u << ((rs - 1u) & 31u);
        if (units == 0 || np == 0 || ns < np) continue;
        for (uint32_t unit = 0; unit < units; unit++) {
            int32_t *x = g + (size_t)unit * ns;
            __syncthreads();                /* previous unit/layer: its stores are issued, cpad is free */
            for (uint32_t j = lane; j < 128; j += 64) cpad[j] = (j >= 128 - np) ? rec[LINNE_AMD_PRM_COEF + p.coef_off[l] + unit * np + (j - (128 - np))] : 0;
            __syncthreads();
            /* history: slot m holds x[t'] with t' = m (mod 128); the first np samples pass through unchanged */
            int32_t h0 = (lane < np) ? x[lane] : 0, h1 = (lane + 64 < np) ? x[lane + 64] : 0;
            const uint32_t c_first = np & ~63u;
            int32_t vin = (c_first + lane < ns) ? x[c_first + lane] : 0;
            for (uint32_t c0 = c_first; c0 < ns; c0 += 64) {
                const int32_t cur = vin;
                if (c0 + 64 < ns) vin = (c0 + 64 + lane < ns) ? x[c0 + 64 + lane] : 0;       /* prefetch the next chunk */
                const uint32_t t_begin = (c0 > np) ? c0 : np, t_end = (c0 + 64 < ns) ? (c0 + 64) : ns;
                const bool odd = (c0 >> 6) & 1u;
                int32_t hr = odd ? h1 : h0;                 /* the register this chunk is reconstructed into */
                for (uint32_t t = t_begin; t < t_end; t++) {
                    const int32_t ca = cpad[(lane - t) & 127u], cb = cpad[(lane + 64u - t) & 127u];
                    const int32_t ha = odd ? h0 : hr, hb = odd ? hr : h1;
                    const int32_t acc = (int32_t)((uint32_t)ha * (uint32_t)ca + (uint32_t)hb * (uint32_t)cb);
                    const uint32_t pred = half + (uint32_t)wave_sum_i32(acc);
                    const int32_t res = __builtin_amdgcn_readlane(cur, (int)(t - c0));
                    const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)pred >> (rs & 31u)));
                    if (lane == (t & 63u)) hr = y;
                }
                if (odd) h1 = hr; else h0 = hr;
                if (c0 + lane >= t_begin && c0 + lane < t_end) x[c0 + lane] = hr;
            }
        }
    }
    __syncthreads();
    /* two-stage de-emphasis (linne_utility.c:215-241): stage-2 inverse then stage-1 inverse, fused; the recurrence
     * itself is scalar (wave-uniform), chunks of 64 samples move through a register */
    if (n > 0 && deemph) {
        const int32_t c0e = rec[LINNE_AMD_PRM_PCOEF + 0], c1e = rec[LINNE_AMD_PRM_PCOEF + 1];
        int32_t zp = rec[LINNE_AMD_PRM_PREV + 1], yp = rec[LINNE_AMD_PRM_PREV + 0];
        int32_t vin = (lane < n) ? g[lane] : 0;
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const int32_t cur = vin;
            if (c0 + 64 < n) vin = (c0 + 64 + lane < n) ? g[c0 + 64 + lane] : 0;
            const uint32_t cnt = (n - c0 < 64) ? (n - c0) : 64;
            int32_t outv = cur;
            for (uint32_t i = 0; i < cnt; i++) {
                const int32_t b = __builtin_amdgcn_readlane(cur, (int)i);
                const int32_t z = (int32_t)((uint32_t)b + (uint32_t)mulshr5(zp, c1e));
                const int32_t y = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                zp = z; yp = y;
                if (lane == i) outv = y;
            }
            if (lane < cnt) g[c0 + lane] = outv;
        }
    }
}

/* Synthesis of a SHORT layer (order <= 16), lanes = channel-frames: a wavefront reconstructs the same layer of 64
 * channel-frames, each lane running its own recurrence (linne_lpc_synthesize.c:8-83) with its unit's coefficients
 * (zero-extended to PL taps) and the last PL outputs in registers -- the time loop is unrolled over one turn of that
 * history ring, so no register moves.  The int32 dot product is evaluated in FP64: coefficients are 8-bit, so
 * |sum c*y| < 2^45 and every FMA is exact; the sum is then reduced modulo 2^32, which is what the reference's wrap-around
 * int32 accumulation holds.  Samples travel in 64 x 64 tiles transposed through LDS (coalesced loads and stores, next
 * tile prefetched into registers).  DEEMPH fuses the two de-emphasis stages (linne_utility.c:215-241), a scalar
 * recurrence per lane, behind layer 0. */
#define SYN_T 64
template <int PL, bool DEEMPH>
__global__ __launch_bounds__(64) void k_synth_small(DecPlan p, uint32_t layer)
{
    __shared__ int32_t tile[SYN_T][65];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64, S = p.S;
    const uint32_t nrows = p.F * p.C;
    uint32_t cf = row0 + lane;
    const bool have = cf < nrows;
    if (!have) cf = nrows - 1;
    const uint32_t n = p.nsmp[cf / p.C];
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + layer], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + layer];
    const uint32_t np = (units ? PL / units : 0u), ns = (units ? n / units : 0u);
    const bool skip = (units == 0 || np == 0 || ns < np);         /* linne_decoder.c: such a layer leaves the data unchanged */
    const uint32_t half = 1u << ((rs - 1u) & 31u);
    const int32_t *crec = rec + LINNE_AMD_PRM_COEF + p.coef_off[layer];
    double c[PL], h[PL];
#pragma unroll
    for (int k = 0; k < PL; k++) { c[k] = 0.0; h[k] = 0.0; }
    uint32_t tl = 0, unit = 0;                                   /* place inside the current unit; its index */
    bool fresh = true;                                           /* the unit's coefficients are not loaded yet */
    int32_t zp = 0, yp = 0, c0e = 0, c1e = 0;
    if (DEEMPH) { c0e = rec[LINNE_AMD_PRM_PCOEF + 0]; c1e = rec[LINNE_AMD_PRM_PCOEF + 1]; zp = rec[LINNE_AMD_PRM_PREV + 1]; yp = rec[LINNE_AMD_PRM_PREV + 0]; }
    /* wave-uniform number of tiles: the longest frame of the block */
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)nmax, o); nmax = other > nmax ? other : nmax; }
    const uint32_t ntiles = (nmax + SYN_T - 1) / SYN_T;
    int32_t pre[64];
    auto issue = [&](uint32_t t) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const uint32_t row = (row0 + r < nrows) ? row0 + r : nrows - 1, sidx = t * SYN_T + lane;
            pre[r] = (sidx < S) ? p.data[(size_t)row * S + sidx] : 0;
        }
    };
    if (ntiles) issue(0);
    for (uint32_t t = 0; t < ntiles; t++) {
#pragma unroll
        for (int r = 0; r < 64; r++) tile[lane][r] = pre[r];      /* transposed: tile[sample][row] */
        if (t + 1 < ntiles) issue(t + 1);
        __syncthreads();
#pragma unroll 1
        for (uint32_t s0 = 0; s0 < SYN_T; s0 += PL) {
            /* One turn of the history ring = PL steps.  If no lane of the wave meets a unit boundary inside the turn (the
             * common case: boundaries are hundreds of samples apart), the turn runs as straight-line code, so the FMAs on
             * older outputs of step k+1 overlap the tail of step k; otherwise the turn takes the careful path. */
            const bool calm = !__any((!skip && unit < units) && (fresh || tl + PL > ns));
            if (calm) {
                int32_t rin[PL];                                 /* the turn's residuals, read before any output is stored */
#pragma unroll
                for (int tt = 0; tt < PL; tt++) rin[tt] = tile[s0 + tt][lane];
#pragma unroll
                for (int tt = 0; tt < PL; tt++) {
                    const uint32_t sidx = t * SYN_T + s0 + tt;
                    const int32_t res = rin[tt];
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                    for (int k = 0; k < PL; k++) {
                        const double prod_h = h[(tt + k) % PL];
                        if ((k & 3) == 0) a0 = __builtin_fma(c[k], prod_h, a0);
                        else if ((k & 3) == 1) a1 = __builtin_fma(c[k], prod_h, a1);
                        else if ((k & 3) == 2) a2 = __builtin_fma(c[k], prod_h, a2);
                        else a3 = __builtin_fma(c[k], prod_h, a3);
                    }
                    const double acc = (a0 + a1) + (a2 + a3);
                    const uint32_t pred = half + (uint32_t)__double2loint(acc + 6755399441055744.0);
                    const bool predict = !skip && (tl + (uint32_t)tt >= np) && unit < units;
                    int32_t y = predict ? (int32_t)((uint32_t)res - (uint32_t)((int32_t)pred >> (rs & 31u))) : res;
                    h[tt] = (double)y;
                    if (DEEMPH) {
                        const int32_t z = (int32_t)((uint32_t)y + (uint32_t)mulshr5(zp, c1e));
                        const int32_t yy = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                        const bool live = sidx < n;
                        zp = live ? z : zp; yp = live ? yy : yp;
                        y = yy;
                    }
                    tile[s0 + tt][lane] = y;
                }
                tl += PL;                                        /* tl + PL <= ns for predicting lanes; == ns closes the unit */
                if (!skip && unit < units && tl == ns) { tl = 0; unit++; fresh = true; }
                continue;
            }
#pragma unroll
            for (int tt = 0; tt < PL; tt++) {                    /* sample index = tt (mod PL): ring slot tt is the oldest */
                const uint32_t sidx = t * SYN_T + s0 + tt;
                if (fresh && !skip && unit < units) {            /* first sample of a unit: its zero-extended coefficients */
#pragma unroll
                    for (int k = 0; k < PL; k++) c[k] = ((uint32_t)k >= PL - np) ? (double)crec[unit * np + ((uint32_t)k - (PL - np))] : 0.0;
                }
                fresh = false;
                const int32_t res = tile[s0 + tt][lane];
                /* four partial sums (exact integers: any order) keep the FMA chain short */
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int k = 0; k < PL; k++) {
                    const double prod_h = h[(tt + k) % PL];
                    if ((k & 3) == 0) a0 = __builtin_fma(c[k], prod_h, a0);
                    else if ((k & 3) == 1) a1 = __builtin_fma(c[k], prod_h, a1);
                    else if ((k & 3) == 2) a2 = __builtin_fma(c[k], prod_h, a2);
                    else a3 = __builtin_fma(c[k], prod_h, a3);
                }
                const double acc = (a0 + a1) + (a2 + a3);
                const uint32_t sum32 = (uint32_t)__double2loint(acc + 6755399441055744.0);   /* acc mod 2^32: |acc| < 2^45, so adding 1.5 * 2^52 leaves the integer in the low mantissa bits, two's complement */
                const uint32_t pred = half + sum32;
                int32_t y = res;
                if (!skip && tl >= np && unit < units) y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)pred >> (rs & 31u)));
                h[tt] = (double)y;
                tl++;
                if (tl == ns) { tl = 0; unit++; fresh = true; }
                if (DEEMPH) {
                    const int32_t z = (int32_t)((uint32_t)y + (uint32_t)mulshr5(zp, c1e));
                    const int32_t yy = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                    if (sidx < n) { zp = z; yp = yy; }
                    y = yy;
                }
                tile[s0 + tt][lane] = y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const uint32_t row = row0 + r, sidx = t * SYN_T + lane;
            if (row < nrows && sidx < p.nsmp[row / p.C]) p.data[(size_t)row * S + sidx] = tile[lane][r];
        }
        __syncthreads();
    }
}

/* Synthesis of a LONG layer (order 32..128), four lanes per channel-frame: a wavefront reconstructs the same layer of 16
 * channel-frames; lane g of a channel-frame owns the taps [TP g, TP g + TP) of the zero-extended coefficient vector and the TP
 * outputs they multiply, BOTH in registers: its window of the history slides by one sample per step -- the value that leaves
 * lane g + 1's window enters lane g's (one DPP quad move), the newest output enters lane 3's -- and the time loop is unrolled
 * over one turn of the TP-slot ring, so every ring index is a compile-time constant (k_synth_small's scheme, four lanes wide).
 * The int32 dot product is evaluated in FP64 (exact, see k_synth_small), the four partial sums of a channel-frame meet through two
 * DPP quad permutes, and every lane of the quad finishes the step redundantly.  All taps but the newest are summed one step
 * ahead (`part`), so only one multiply-add, the quad sum and the integer tail are on the sample-to-sample critical path.
 * (The first form kept the history in an LDS ring: 32 LDS reads per lane and sample, 8.5 ms per 31 008 channel-frames, bound by
 * the LDS's bandwidth.) */
#define SYB_T 64
template <int PL>
__global__ __launch_bounds__(64) void k_synth_big(DecPlan p, uint32_t layer)
{
    constexpr int TP = PL / 4;                                   /* taps per lane */
    static_assert(SYB_T % TP == 0, "a tile is whole turns of the ring");
    __shared__ int32_t tile[16][SYB_T];
    const uint32_t lane = threadIdx.x, cfl = lane >> 2, g = lane & 3u, S = p.S;
    const uint32_t nrows = p.F * p.C, row0 = blockIdx.x * 16;
    uint32_t cf = row0 + cfl;
    if (cf >= nrows) cf = nrows - 1;
    const uint32_t n = p.nsmp[cf / p.C];
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + layer], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + layer];
    const uint32_t np = (units ? PL / units : 0u), ns = (units ? n / units : 0u);
    const bool skip = (units == 0 || np == 0 || ns < np);
    const uint32_t half = 1u << ((rs - 1u) & 31u);
    const int32_t *crec = rec + LINNE_AMD_PRM_COEF + p.coef_off[layer];
    double c[TP], h[TP];                                         /* my taps; my window of the outputs: element j (oldest first) at h[(turn position + j) % TP] */
#pragma unroll
    for (int j = 0; j < TP; j++) { c[j] = 0.0; h[j] = 0.0; }
    uint32_t tl = 0, unit = 0;
    bool fresh = true;
    double ynew = 0.0, part = 0.0;                               /* the previous step's output; the partial sum made ahead */
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)nmax, o); nmax = other > nmax ? other : nmax; }
    const uint32_t ntiles = (nmax + SYB_T - 1) / SYB_T;
    int32_t pre[16];
    auto issue = [&](uint32_t t) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t row = (row0 + r < nrows) ? row0 + r : nrows - 1, sidx = t * SYB_T + lane;
            pre[r] = (sidx < S) ? p.data[(size_t)row * S + sidx] : 0;
        }
    };
    if (ntiles) issue(0);
    __syncthreads();
    for (uint32_t t = 0; t < ntiles; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) tile[r][lane] = pre[r];
        if (t + 1 < ntiles) issue(t + 1);
        __syncthreads();
        /* one step: the oldest element of my window sits at h[TT_].  CALM_: no lane meets a unit's first sample in this turn, the
         * taps stay as they are and `TL_` is the place inside the unit */
#define SYB_STEP(TT_, S_, CALM_, TL_) { \
                const int32_t res = tile[cfl][S_]; \
                /* window element j multiplies tap j; everything but the newest output (lane 3's last element) was summed during the \
                 * previous step (`part`): only that one product is on this step's critical path */ \
                double acc = (g == 3u) ? __builtin_fma(c[TP - 1], ynew, part) : part; \
                {   /* quad sum: lanes 4q .. 4q+3 */ \
                    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc), 0xB1, 0xf, 0xf, true);     /* quad_perm: 1,0,3,2 */ \
                    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc), 0xB1, 0xf, 0xf, true); \
                    acc += __hiloint2double(hi, lo); \
                    lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc), 0x4E, 0xf, 0xf, true);         /* quad_perm: 2,3,0,1 */ \
                    hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc), 0x4E, 0xf, 0xf, true); \
                    acc += __hiloint2double(hi, lo); \
                } \
                /* the window slides: my oldest element leaves for lane g - 1, lane g + 1's oldest enters as my newest (lane 3: the \
                 * slot is filled with this step's output below).  Then next step's partial sum over the elements 0 .. TP-2 of the \
                 * NEW window (h[TT_ + 1 ..]) -- lanes 0..2 take their newest element in too: it is known already */ \
                { \
                    const double out = h[TT_]; \
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(out), 0xF9, 0xf, 0xf, true);     /* quad_perm: 1,2,3,3: from lane g + 1 */ \
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(out), 0xF9, 0xf, 0xf, true); \
                    h[TT_] = __hiloint2double(hi, lo);           /* (lane 3: overwritten with ynew once it is known) */ \
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0; \
                    _Pragma("unroll") for (int j = 0; j < TP - 1; j++) { \
                        const double hv = h[(TT_ + 1 + j) % TP]; \
                        if ((j & 3) == 0) a0 = __builtin_fma(c[j], hv, a0); \
                        else if ((j & 3) == 1) a1 = __builtin_fma(c[j], hv, a1); \
                        else if ((j & 3) == 2) a2 = __builtin_fma(c[j], hv, a2); \
                        else a3 = __builtin_fma(c[j], hv, a3); \
                    } \
                    const double hl = (g == 3u) ? 0.0 : h[TT_]; \
                    a3 = __builtin_fma(c[TP - 1], hl, a3); \
                    part = (a0 + a1) + (a2 + a3); \
                } \
                const uint32_t sum32 = (uint32_t)__double2loint(acc + 6755399441055744.0);      /* acc mod 2^32 (see k_synth_small) */ \
                const uint32_t pred = half + sum32; \
                int32_t y = res; \
                if (!skip && (TL_) >= np && unit < units) y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)pred >> (rs & 31u))); \
                ynew = (double)y; \
                if (g == 3u) h[TT_] = ynew;                      /* the newest element of lane 3's window */ \
                if (g == 0) tile[cfl][S_] = y; }
#pragma unroll 1
        for (uint32_t sb = 0; sb < SYB_T; sb += TP) {
            /* One turn of the ring = TP steps.  If no lane of the wave meets a unit boundary inside the turn (the common case),
             * the turn runs as straight-line code; otherwise it takes the careful path (k_synth_small's scheme). */
            const bool calm = !__any((!skip && unit < units) && (fresh || tl + (uint32_t)TP > ns));
            if (calm) {
#pragma unroll
                for (int tt = 0; tt < TP; tt++) SYB_STEP(tt, sb + (uint32_t)tt, true, tl + (uint32_t)tt)
                tl += TP;                                        /* tl + TP <= ns for predicting lanes; == ns closes the unit */
                if (!skip && unit < units && tl == ns) { tl = 0; unit++; fresh = true; }
                continue;
            }
#pragma unroll
            for (int tt = 0; tt < TP; tt++) {
                if (fresh && !skip && unit < units) {            /* first sample of a unit: my taps of its zero-extended coefficients */
#pragma unroll
                    for (int j = 0; j < TP; j++) {
                        const uint32_t k = (uint32_t)TP * g + (uint32_t)j;
                        c[j] = (k >= PL - np) ? (double)crec[unit * np + (k - (PL - np))] : 0.0;
                    }
                    /* (`part`, made ahead with the unit before's taps, is not used: a unit's first np samples are copied) */
                }
                fresh = false;
                SYB_STEP(tt, sb + (uint32_t)tt, false, tl)
                tl++;
                if (tl == ns) { tl = 0; unit++; fresh = true; }
            }
        }
#undef SYB_STEP
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t row = row0 + r, sidx = t * SYB_T + lane;
            if (row < nrows && sidx < p.nsmp[row / p.C]) p.data[(size_t)row * S + sidx] = tile[r][lane];
        }
        __syncthreads();
    }
}

/* MS -> LR (linne_utility.c:135-147) */
__global__ void k_ms_to_lr(DecPlan p)
{
    const uint32_t f = blockIdx.x, s = blockIdx.y * blockDim.x + threadIdx.x;      /* grid = (frames, sample tiles) */
    if (s >= p.nsmp[f]) return;
    int32_t *m = p.data + (size_t)f * p.C * p.S, *sd = m + p.S;
    const uint32_t l = (uint32_t)m[s] - (uint32_t)(sd[s] >> 1);
    m[s] = (int32_t)l;
    sd[s] = (int32_t)((uint32_t)sd[s] + l);
}



/* the decoded PCM of <= 16-bit audio as int16 (half the D2H bytes); a sample outside the int16 range -- only a stream no encoder
 * wrote can decode to one -- raises *flag, and the host takes the int32 path.  Only a frame's own samples count (what lies behind a
 * short frame's end is nobody's data). */
__global__ void k_narrow16(const int32_t *src, int16_t *dst, uint64_t count, uint32_t *flag, const uint32_t *nsmp, uint32_t C, uint32_t S)
{
    uint32_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const int32_t v = src[i];
        const uint32_t s = (uint32_t)(i % S), f = (uint32_t)(i / ((uint64_t)C * S));
        if ((v < -32768 || v > 32767) && s < nsmp[f]) bad = 1;
        dst[i] = (int16_t)v;
    }
    if (bad) atomicOr(flag, 1u);
}

#endif

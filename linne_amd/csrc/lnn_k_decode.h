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
            if (row < nrows && sidx < (uint32_t)__builtin_amdgcn_readlane((int)n, r)) p.data[(size_t)row * S + sidx] = tile[lane][r];      /* (lane r holds row r's length: no division by the channel count and no load per row) */
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
            if (row < nrows && sidx < (uint32_t)__builtin_amdgcn_readlane((int)n, 4 * r)) p.data[(size_t)row * S + sidx] = tile[r][lane];    /* (lanes 4 r .. 4 r + 3 hold row r's length) */
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------------------------------
 * k_synth_pipe: the LATENCY form of the synthesis -- what LINNEDecoder_DecodeBlock waits for (tools/linne_player decodes a block
 * per audio callback, linne_player.c:66-118; linne_decoder.c:503-522).  One block per channel-frame, one WAVE PER STAGE of the
 * cascade: the layers in decode order (L-1 .. 0), then the two de-emphasis stages; the stages work IN PLACE on the channel-frame's
 * samples in LDS and follow each other at a distance of 16 samples (a progress counter per stage), so the call takes the time of
 * the slowest stage, not the sum.
 *
 * A layer stage walks its units in blocks of 16 outputs.  y[t] = r[t] - ((half + sum_d c_d y[t - d]) >> rshift) is split by the
 * distance d of a tap (linne_lpc_synthesize.c:27-33 written with d = np - ord):
 *   d <= i            (i = place in the block) the block's own outputs: the serial part.  Lane i holds output i's sum; when y_j is
 *                     final it is read with v_readlane and every later lane adds its tap times y_j -- one dependent multiply-add
 *                     per sample instead of a wave-wide reduction (k_synthesize: ~290 cycles per sample, this: ~50);
 *   i < d <= i + 16   the previous block's outputs: added in the same steps into the NEXT block's sums (second coefficient set);
 *   d > i + 16        older samples (only layers of more than 16 taps): a Toeplitz product on the matrix unit,
 *                     v_mfma_i32_16x16x64_i8 -- A = the last 64 KS outputs as four planes of signed base-256 digits (rows), B = the
 *                     unit's 8-bit coefficients laid out as a Toeplitz matrix (constant per unit), issued a block ahead and summed
 *                     modulo 2^32 (the reference's wrap-around int32 arithmetic is associative, the planes recombine by shifts).
 * Coefficients are 8-bit by format (the stream codes them with a 256-symbol Huffman code; k_synth_small / k_synth_big rely on the
 * same range).  Frames too long for the LDS image stay with k_synthesize. */
#define SP_RING 256u
#define SP_RINGP 272u                   /* bytes from a digit plane to the next: the four planes' 16-byte reads of a window (one lane each) hit different banks */
struct SpShared { uint32_t prog[LNN_MAXL + 1]; uint32_t pad[4]; int8_t ring[LNN_MAXL][4][SP_RINGP]; };

__device__ __forceinline__ uint32_t sp_load_prog(const uint32_t *q) { return __hip_atomic_load(q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sp_publish(uint32_t *q, uint32_t v) { __hip_atomic_store(q, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sp_wait(const uint32_t *q, uint32_t need) { if (q) while (sp_load_prog(q) < need) __builtin_amdgcn_s_sleep(1); }

/* c * s modulo 2^32 for an 8-bit c and any 32-bit s, on the full-rate 24-bit multiplier: s = sl + 2^16 sh */
__device__ __forceinline__ uint32_t sp_mul8(int32_t c, int32_t sl, int32_t sh) { return (uint32_t)__mul24(c, sl) + ((uint32_t)__mul24(c, sh) << 16); }

/* the four signed base-256 digits d_b in [-128, 127] of v (v = sum d_b 256^b modulo 2^32), byte b of the result: adding 128 to each
 * of the three low digits makes them the unsigned bytes of v + 0x00808080 (the carries run as they must), and x - 128 = x ^ 0x80 */
__device__ __forceinline__ uint32_t sp_digits(int32_t v) { return ((uint32_t)v + 0x00808080u) ^ 0x00808080u; }

template <int KS>       /* 64-sample steps of history summed on the matrix unit: 0 for layers of <= 16 taps, 1 up to 80, 2 up to 144 */
__device__ void sp_layer_stage(int32_t *buf, const uint32_t *pin, uint32_t *pout, int8_t (*ring)[SP_RINGP], const int32_t *coef,
        uint32_t P, uint32_t units, uint32_t rs, uint32_t n, uint32_t lane)
{
    const uint32_t np = units ? P / units : 0u, ns = units ? n / units : 0u;
    if (units == 0 || np == 0 || ns < np) { sp_wait(pin, n); sp_publish(pout, n); return; }        /* (what k_synthesize skips) */
    const uint32_t i = lane & 15u, g = lane >> 4;
    const uint32_t half = 1u << ((rs - 1u) & 31u), sh_ = rs & 31u;
    for (uint32_t unit = 0; unit < units; unit++) {
        const uint32_t base = unit * ns;
        const int32_t *cu = coef + (size_t)unit * np;
        /* tap of distance d: cu[np - d] (linne_lpc_synthesize.c:30: coef[ord] meets data[smpl + ord], the output is data[smpl + np]) */
        int32_t ccA[16], ccB[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int32_t da = (int32_t)i - j, db = 16 + (int32_t)i - j;
            ccA[j] = (da >= 1 && (uint32_t)da <= np) ? cu[np - (uint32_t)da] : 0;
            ccB[j] = ((uint32_t)db <= np) ? cu[np - (uint32_t)db] : 0;
        }
        lnn_v4i bfrag[KS ? KS : 1];
        if (KS) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                uint32_t w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t k = 64u * s + 16u * g + e, d = 64u * KS + 16u - k + i;      /* window element k lies d samples before output i */
                    const int32_t c = (d <= np) ? cu[np - d] : 0;
                    w[e >> 2] |= ((uint32_t)c & 0xFFu) << (8 * (e & 3));
                }
                bfrag[s][0] = (int)w[0]; bfrag[s][1] = (int)w[1]; bfrag[s][2] = (int)w[2]; bfrag[s][3] = (int)w[3];
            }
        }
        /* the unit's first np samples pass through (linne_lpc_synthesize.c:26): they are this stage's outputs as they are */
        sp_wait(pin, base + np);
        if (KS) {       /* their digits: sample t of the unit sits at ring index (t - np + 128) mod 256 */
            for (uint32_t t = lane; t < np; t += 64u) {
                const uint32_t dg = sp_digits(buf[base + t]);
                const uint32_t ix = (t - np + 128u) & (SP_RING - 1u);
                ring[0][ix] = (int8_t)dg; ring[1][ix] = (int8_t)(dg >> 8); ring[2][ix] = (int8_t)(dg >> 16); ring[3][ix] = (int8_t)(dg >> 24);
            }
        }
        sp_publish(pout, base + np);
        /* what the 16 samples in front of the first block add to its sums (distances i + 1 .. i + 16) */
        uint32_t nxt = 0;
        {
            const int32_t tprev = (int32_t)np - 16 + (int32_t)i;
            const int32_t yp = (lane < 16u && tprev >= 0) ? buf[base + (uint32_t)tprev] : 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int32_t sv = __builtin_amdgcn_readlane(yp, j);
                nxt += sp_mul8(ccB[j], sv & 0xFFFF, sv >> 16);
            }
        }
        auto window = [&](uint32_t m) -> uint32_t {             /* matrix-unit part of block m's sums: the 64 KS samples that end 16 before it */
            lnn_v4i acc4 = { 0, 0, 0, 0 };
            if (KS) {
                const uint32_t w0 = 128u + 16u * m - 16u - 64u * KS;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    lnn_v4i a = { 0, 0, 0, 0 };
                    if (i < 4u) a = *(const lnn_v4i *)&ring[i][(w0 + 64u * s + 16u * g) & (SP_RING - 1u)];
                    acc4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bfrag[s], acc4, 0, 0, 0);
                }
            }
            return (uint32_t)acc4[0] + ((uint32_t)acc4[1] << 8) + ((uint32_t)acc4[2] << 16) + ((uint32_t)acc4[3] << 24);     /* lanes 0 .. 15: planes 0 .. 3 of column i */
        };
        const uint32_t nblk = (ns - np + 15u) / 16u;
        uint32_t mcur = window(0);
        for (uint32_t m = 0; m < nblk; m++) {
            const uint32_t t0 = np + 16u * m, cnt = (ns - t0 < 16u) ? (ns - t0) : 16u;
            sp_wait(pin, base + t0 + cnt);
            const int32_t res = (lane < cnt) ? buf[base + t0 + i] : 0;
            const uint32_t mnext = (m + 1u < nblk) ? window(m + 1u) : 0u;      /* issued now, needed a block later */
            const uint32_t acc0 = half + mcur + nxt;
            uint32_t acc = acc0;
            nxt = 0;
            /* Speculation: every output of the block fits 24 bits (any audio of <= 23 bits does) -- then c * y is ONE full-rate
             * 24-bit multiply-add per sum, and a step of the dependent chain is shift, subtract, v_readlane, multiply-add.  An
             * output outside that range (loud 24-bit material, a damaged stream) is seen after the block, and the block is done
             * again with the multiplication split in halves (exact modulo 2^32 for any value). */
            /* (a lone wave issues an instruction every ~9 cycles whatever its dependences, so a step costs what it counts in
             * instructions: shift, subtract, v_readlane, two multiply-adds and the three wait states the scalar result needs) */
            /* (round 3: y_j reaches the lanes through a DPP row_newbcast -- lanes 0 .. 15 are the block -- not through v_readlane: no
             * trip through a scalar register and its wait states) */
#define SP_STEP(J) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); \
                const int32_t sv = __builtin_amdgcn_update_dpp(0, y, 0x150 + (J), 0xf, 0xf, true); \
                acc += (uint32_t)__mul24(ccA[J], sv); nxt += (uint32_t)__mul24(ccB[J], sv); }
            SP_STEP(0) SP_STEP(1) SP_STEP(2) SP_STEP(3) SP_STEP(4) SP_STEP(5) SP_STEP(6) SP_STEP(7)
            SP_STEP(8) SP_STEP(9) SP_STEP(10) SP_STEP(11) SP_STEP(12) SP_STEP(13) SP_STEP(14) SP_STEP(15)
#undef SP_STEP
            /* lane i's sum has not changed since step i (ccA[j] = 0 for j >= i): its output, whatever the step */
            int32_t yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
            const bool fits = (lane >= 16u) || (((int32_t)((uint32_t)yout << 8) >> 8) == yout);
            if (!__all(fits)) {
                acc = acc0; nxt = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
                    const int32_t sv = __builtin_amdgcn_readlane(y, j);
                    const int32_t sl = sv & 0xFFFF, sh = sv >> 16;
                    acc += sp_mul8(ccA[j], sl, sh);
                    nxt += sp_mul8(ccB[j], sl, sh);
                }
                yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
            }
            if (lane < cnt) buf[base + t0 + i] = yout;
            sp_publish(pout, base + t0 + cnt);
            if (KS && lane < 16u) {
                const uint32_t dg = sp_digits(yout);
                const uint32_t ix = (128u + 16u * m + i) & (SP_RING - 1u);
                ring[0][ix] = (int8_t)dg; ring[1][ix] = (int8_t)(dg >> 8); ring[2][ix] = (int8_t)(dg >> 16); ring[3][ix] = (int8_t)(dg >> 24);
            }
            mcur = mnext;
        }
    }
    sp_wait(pin, n);            /* what lies behind the last unit passes through */
    sp_publish(pout, n);
}

__global__ __launch_bounds__(64 * (LNN_MAXL + 1)) void k_synth_pipe(DecPlan p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sp_dyn[];
    SpShared *sh = (SpShared *)sp_dyn;
    int32_t *buf = (int32_t *)(sp_dyn + sizeof(SpShared));
    const uint32_t cf = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t n = p.nsmp[cf / p.C], S = p.S, L = p.L;
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    int32_t *g = p.data + (size_t)cf * S;
    if (tid <= LNN_MAXL) sh->prog[tid] = 0u;
    for (uint32_t t = tid; t < n; t += blockDim.x) buf[t] = g[t];
    __syncthreads();
    if (wave < L) {
        const uint32_t l = L - 1u - wave, P = p.P[l];
        const uint32_t *pin = wave ? &sh->prog[wave - 1u] : (const uint32_t *)0;
        const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + l], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + l];
        const int32_t *coef = rec + LINNE_AMD_PRM_COEF + p.coef_off[l];
        if (P <= 16u) sp_layer_stage<0>(buf, pin, &sh->prog[wave], sh->ring[wave], coef, P, units, rs, n, lane);
        else if (P <= 64u) sp_layer_stage<1>(buf, pin, &sh->prog[wave], sh->ring[wave], coef, P, units, rs, n, lane);
        else sp_layer_stage<2>(buf, pin, &sh->prog[wave], sh->ring[wave], coef, P, units, rs, n, lane);
    } else if (wave == L) {
        /* two-stage de-emphasis (linne_utility.c:215-241) behind the last layer stage, 64 samples at a time; the recurrences are
         * wave-uniform (scalar) */
        const uint32_t *pin = &sh->prog[L - 1u];
        const int32_t c0e = rec[LINNE_AMD_PRM_PCOEF + 0], c1e = rec[LINNE_AMD_PRM_PCOEF + 1];
        int32_t zp = rec[LINNE_AMD_PRM_PREV + 1], yp = rec[LINNE_AMD_PRM_PREV + 0];
        for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
            const uint32_t cnt = (n - c0 < 64u) ? (n - c0) : 64u;
            sp_wait(pin, c0 + cnt);
            const int32_t cur = (lane < cnt) ? buf[c0 + lane] : 0;
            int32_t outv = cur;
            if (cnt == 64u) {                 /* whole chunk: lane indices are constants, the results go back with v_writelane */
#pragma unroll
                for (int k = 0; k < 64; k++) {
                    const int32_t b = __builtin_amdgcn_readlane(cur, k);
                    const int32_t z = (int32_t)((uint32_t)b + (uint32_t)mulshr5(zp, c1e));
                    const int32_t y = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                    zp = z; yp = y;
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(outv) : "s"(y), "n"(k));
                }
            } else for (uint32_t k = 0; k < cnt; k++) {
                const int32_t b = __builtin_amdgcn_readlane(cur, (int)k);
                const int32_t z = (int32_t)((uint32_t)b + (uint32_t)mulshr5(zp, c1e));
                const int32_t y = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                zp = z; yp = y;
                if (lane == k) outv = y;
            }
            if (lane < cnt) buf[c0 + lane] = outv;
        }
    }
    __syncthreads();
    for (uint32_t t = tid; t < n; t += blockDim.x) g[t] = buf[t];
}
#define SP_LDS_BYTES(n_) (sizeof(SpShared) + sizeof(int32_t) * (size_t)(n_))

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
/* the same for audio of 17 .. 24 bits: packed little-endian 3-byte samples (three quarters of the D2H bytes) */
__global__ void k_narrow24(const int32_t *src, uint8_t *dst, uint64_t count, uint32_t *flag, const uint32_t *nsmp, uint32_t C, uint32_t S)
{
    uint32_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const int32_t v = src[i];
        const uint32_t s = (uint32_t)(i % S), f = (uint32_t)(i / ((uint64_t)C * S));
        if ((v < -8388608 || v > 8388607) && s < nsmp[f]) bad = 1;
        dst[3u * i] = (uint8_t)v; dst[3u * i + 1u] = (uint8_t)(v >> 8); dst[3u * i + 2u] = (uint8_t)(v >> 16);
    }
    if (bad) atomicOr(flag, 1u);
}
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

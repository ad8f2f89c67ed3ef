/* lnn_k_search.h -- k_search_long: the certified unit-count search of the LONG layer (64 or 128 taps) in one pass over the window.
 * Part of the single translation unit lnn_device.hip (included there, after lnn_k_fir.h); not a stand-alone header.
 *
 * Replaces k_fir2<2, false, true> (lnn_k_fir.h) for the frames it takes (search_long_takes: every trial present, the analysis
 * length a multiple of the 2048-sample tile -- all full 10240-sample frames); k_fir2 keeps the rest (ragged tails).  Same
 * outputs: per-wave partial sums of |trial residual| for every trial (linne_network.c:318-335, order-free: the certificate of
 * k_select decides whether they settle the argmin), max |x| and coefficient norms for the certificate, and the one-unit trial's
 * forward output (linne_network.c:165-210), bit for bit.
 *
 * What the counters said about k_fir2<2> (profiles/r02_sq_*.json): VALU issue is the limit (61-73 % busy), a third of the VALU
 * instructions are not multiply-adds, 66 % of the LDS-active cycles are bank conflicts, and the 4 trials of <= 8 taps cost 15 %
 * of the time for 4 % of the multiply-adds (a window fill, an epilogue and a wave reduction per trial).  Hence:
 *   - ONE pass over the 128-tap window feeds the three big trials.  Trial u (order P/u) looks back P/u samples: its window is
 *     the tail of the one-unit trial's.  While the one-unit chain walks taps 0 .. P-1, the two-unit trial joins at tap P/2 and
 *     the four-unit trial at tap 3P/4, on the same window registers: the window is read from LDS once, not once per trial.
 *   - a wave's 512 samples lie inside one unit for u <= 4 (the analysis length is a multiple of 2048), so the big trials'
 *     coefficients are wave-uniform: broadcast LDS reads, free of bank conflicts.
 *   - the tile image in LDS is padded (2 doubles behind every 8 samples): lane l's window is 80 bytes from lane l-1's, window
 *     reads are conflict-free, and with 8 taps per unrolled pair of steps every offset is a compile-time constant.
 *   - the five small trials (16, 8, 4, 2, 1 taps) run from ONE 24-sample register window.
 * Two forms (template TWO; LINNE_AMD_SEARCH_TWO, default 1).  TWO = false: ONE pass, the joining trials ride on the one-unit
 * trial's window registers (122 VGPRs: four waves per SIMD).  TWO = true: a pass per big trial -- the one-unit trial over the whole
 * window, its output and search term, then the two-unit trial over the window's last half and the four-unit trial over its last
 * quarter, each with the ring to itself (84 VGPRs: five waves per SIMD; the window's second half is read from LDS twice, which costs
 * less than the fifth wave gains: 24.9 ms against 25.7).
 * The one-unit chain is the forward pass's (predict from 0.0, separate multiply and add, taps in order); everything else runs
 * on fused multiply-adds inside the certificate's slack (see k_select).  Zero history in front of sample 0 stands in for the
 * reference's skipped taps: adding +-0.0 products first leaves a chain's bits unchanged.
 */
#ifndef LNN_K_SEARCH_H_INCLUDED
#define LNN_K_SEARCH_H_INCLUDED

#define SL_XPAD(i) ((i) + 2 * ((i) >> 3))

template <int P, bool TWO>
__global__ __launch_bounds__(FIR_THREADS, TWO ? 5 : 4) void k_search_long(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int NT = (P == 128) ? 8 : 7;                 /* trials: u = 1 .. P (P = 128: u <= 128) */
    constexpr int NBIG = NT - 5;                           /* orders P, P/2 (, P/4): >= 32 taps */
    constexpr int HS = 16 + 8 + 4 + 2 + 1;                 /* taps of the small trials */
    __shared__ __attribute__((aligned(16))) double xs[SL_XPAD(P + FIR_TILE + 8) + 2];
    __shared__ __attribute__((aligned(16))) double hsm[5][LNN_MAXP];          /* coefficients of the five small trials (16 .. 1 taps), all units */
    __shared__ __attribute__((aligned(16))) double hbg[3][LNN_MAXP + 8];      /* coefficients of the big trials, all units (+8: the loop reads one step ahead) */
    const uint32_t job = blockIdx.x, tid = threadIdx.x, s0 = blockIdx.y * FIR_TILE;
    const DevClass &c = job_class(p, job);
    const uint32_t na = c.na;
    if (!search_long_takes(p, layer, c) || s0 >= na) return;
    const double *x = p.sig + ((size_t)job * 2 + cur) * p.S;
    const double *const hglob = p.tcoef + (size_t)job * LNN_MAXT * LNN_MAXP;
    /* stage the tile (P samples of history; zeros in front of sample 0) and the small trials' coefficients */
    if (s0 >= (uint32_t)P) {
        for (uint32_t i = 2 * tid; i < P + FIR_TILE + 8; i += 2 * FIR_THREADS) {
            const uint32_t g = s0 - P + i;
            lnn_d2 v; v.x = 0.0; v.y = 0.0;
            if (g + 1 < na) v = *(const lnn_d2 *)(x + g); else if (g < na) v.x = x[g];
            *(lnn_d2 *)(xs + SL_XPAD(i)) = v;
        }
    } else {
        for (uint32_t i = tid; i < P + FIR_TILE + 8; i += FIR_THREADS) {
            const int64_t g = (int64_t)s0 - P + i;
            xs[SL_XPAD(i)] = (g >= 0 && g < (int64_t)na) ? x[g] : 0.0;
        }
    }
    for (uint32_t i = tid; i < 5u * P; i += FIR_THREADS) { const uint32_t tt = i / P, k = i % P; hsm[tt][k] = hglob[(size_t)(NBIG + tt) * LNN_MAXP + k]; }
    for (uint32_t i = tid; i < (uint32_t)NBIG * P; i += FIR_THREADS) { const uint32_t tt = i / P, k = i % P; hbg[tt][k] = hglob[(size_t)tt * LNN_MAXP + k]; }
    if (blockIdx.y == 0 && tid < (uint32_t)NT) {            /* per trial: the largest L1 norm of a unit's coefficients (search_slack) */
        const uint32_t u = 1u << tid, np = P >> tid;
        double mx = 0.0;
        for (uint32_t un = 0; un < u; un++) { double a = 0.0; for (uint32_t k = 0; k < np; k++) a += fabs(hglob[(size_t)tid * LNN_MAXP + un * np + k]); mx = fmax(mx, a); }
        p.thsum[(size_t)job * LNN_MAXT + tid] = mx;
    }
    __syncthreads();

    const uint32_t s = s0 + FIR_SPL * tid, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));      /* wave-uniform, and known to be */
    const double *xc = xs + SL_XPAD(P + FIR_SPL * tid);      /* -> x[s]; x[s + j] = xc[j], 0 <= j < 8 */
    const size_t part = blockIdx.y * (FIR_THREADS / 64) + wave;
    const bool last_lane = (tid & 63u) == 63u;
    {   /* max |x| of the wave's samples (search_slack) */
        double mx = 0.0;
#pragma unroll
        for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); mx = fmax(mx, fmax(fabs(v.x), fabs(v.y))); }
        mx = wave_max_f64_lane63(mx);
        if (last_lane) p.txmax[(size_t)job * p.npart + part] = mx;
    }

    /* ---------------- the big trials: one pass over the window ---------------- */
    {
        const uint32_t ws = s0 + wave * 64u * FIR_SPL;                           /* first sample of the wave: wave-uniform */
        /* The coefficients come from LDS by broadcast reads (every lane of the wave reads the same address: its unit's
         * coefficients).  Scalar loads from global memory were tried first: the co-resident blocks' coefficient sets (3 KB
         * each) do not fit the scalar cache, every 64-byte line was a miss, and the waves sat on s_waitcnt 60 % of the time. */
        const double *c0 = hbg[0];                                                                  /* u = 1 */
        const double *c1 = hbg[1] + (size_t)(ws / (na >> 1)) * (P / 2);                             /* u = 2: the wave's unit */
        const double *c2 = hbg[2] + (size_t)(ws / (na >> 2)) * (P / 4);                             /* u = 4 (P = 128 only) */
        double a0[FIR_SPL], a1[FIR_SPL], a2[FIR_SPL];
#pragma unroll
        for (int j = 0; j < FIR_SPL; j++) a0[j] = 0.0;
        const double *xw = xc - (P >> 3) * 10;                                   /* -> x[s - P] in the padded image */
        double w[16];
#pragma unroll
        for (int j = 0; j < 12; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xw + SL_XPAD(j)); w[j] = v.x; w[j + 1] = v.y; }
        /* Step G (of four: the ring position) handles taps q .. q+3 of the one-unit trial.  The new samples q+12 .. q+15 sit at
         * padded xw + 14 (G even) or xw + 20 (G odd); xw moves on one group after every odd step.  Software pipeline: the samples
         * and the one-unit trial's coefficients of the NEXT step are requested before this step's multiply-adds issue (HN); the
         * joining trials' coefficients of this step are requested first thing and used behind the one-unit trial's 64
         * instructions. */
#define SL_LOAD(G, Q, HN0, HN1) \
            const lnn_d2 na_ = *(const lnn_d2 *)(xw + ((G & 1) ? 20 : 14)), nb_ = *(const lnn_d2 *)(xw + ((G & 1) ? 22 : 16)); \
            if (G & 1) xw += 10; \
            HN0 = *(const lnn_d2 *)(c0 + (Q) + 4); HN1 = *(const lnn_d2 *)(c0 + (Q) + 6);
#define SL_RING(G) \
            w[(4 * G + 12) % 16] = na_.x; w[(4 * G + 13) % 16] = na_.y; w[(4 * G + 14) % 16] = nb_.x; w[(4 * G + 15) % 16] = nb_.y;
#define SL_T0(G, HC0, HC1) { const double hh_[4] = { HC0.x, HC0.y, HC1.x, HC1.y }; \
            _Pragma("unroll") for (int kk = 0; kk < 4; kk++) { \
            _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) a0[j] = a0[j] + hh_[kk] * w[(4 * G + kk + j) % 16]; } }
#define SL_TF(G, ACC, H0, H1) { const double hh_[4] = { H0.x, H0.y, H1.x, H1.y }; \
            _Pragma("unroll") for (int kk = 0; kk < 4; kk++) { \
            _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) ACC[j] = __builtin_fma(hh_[kk], w[(4 * G + kk + j) % 16], ACC[j]); } }
#define SL_STEP1(G, Q, HC0, HC1, HN0, HN1) { SL_LOAD(G, Q, HN0, HN1) SL_T0(G, HC0, HC1) SL_RING(G) }
#define SL_STEP2(G, Q, K1, HC0, HC1, HN0, HN1) { \
            const lnn_d2 p0_ = *(const lnn_d2 *)(c1 + (K1)), p1_ = *(const lnn_d2 *)(c1 + (K1) + 2); \
            SL_LOAD(G, Q, HN0, HN1) SL_T0(G, HC0, HC1) SL_TF(G, a1, p0_, p1_) SL_RING(G) }
#define SL_STEP3(G, Q, K1, K2, HC0, HC1, HN0, HN1) { \
            const lnn_d2 p0_ = *(const lnn_d2 *)(c1 + (K1)), p1_ = *(const lnn_d2 *)(c1 + (K1) + 2); \
            const lnn_d2 r0_ = *(const lnn_d2 *)(c2 + (K2)), r1_ = *(const lnn_d2 *)(c2 + (K2) + 2); \
            SL_LOAD(G, Q, HN0, HN1) SL_T0(G, HC0, HC1) SL_TF(G, a1, p0_, p1_) SL_TF(G, a2, r0_, r1_) SL_RING(G) }
        lnn_d2 ha0 = *(const lnn_d2 *)(c0), ha1 = *(const lnn_d2 *)(c0 + 2), hb0, hb1;
        uint32_t q = 0;
        if (TWO) {
            /* two passes over the window: the one-unit trial alone (its accumulators, the ring and the coefficient pairs fit 96
             * registers: a fifth wave per SIMD), its output and search term, then the joining trials over the window's second half */
#pragma unroll 1
            for (; q < (uint32_t)P; q += 16) {
                SL_STEP1(0, q, ha0, ha1, hb0, hb1)      SL_STEP1(1, q + 4, hb0, hb1, ha0, ha1)
                SL_STEP1(2, q + 8, ha0, ha1, hb0, hb1)  SL_STEP1(3, q + 12, hb0, hb1, ha0, ha1)
            }
            {
                double *dst = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S + s;
                double ps0 = 0.0;
#pragma unroll
                for (int j = 0; j < FIR_SPL; j += 2) {
                    const lnn_d2 xv = *(const lnn_d2 *)(xc + j);
                    lnn_d2 o; o.x = (s + j == 0) ? xv.x : (xv.x + a0[j]); o.y = xv.y + a0[j + 1];
                    *(lnn_d2 *)(dst + j) = o;
                    ps0 += (s + j == 0) ? 0.0 : fabs(o.x); ps0 += fabs(o.y);
                }
                ps0 = wave_sum_f64_lane63(ps0);
                if (last_lane) p.tsum[((size_t)job * LNN_MAXT + 0) * p.npart + part] = ps0;
            }
#define SL_LOADB(G) \
            const lnn_d2 na_ = *(const lnn_d2 *)(xw + ((G & 1) ? 20 : 14)), nb_ = *(const lnn_d2 *)(xw + ((G & 1) ? 22 : 16)); \
            if (G & 1) xw += 10;
#define SL_STEPB(G, ACC, CP, K) { const lnn_d2 p0_ = *(const lnn_d2 *)(CP + (K)), p1_ = *(const lnn_d2 *)(CP + (K) + 2); \
            SL_LOADB(G) SL_TF(G, ACC, p0_, p1_) SL_RING(G) }
            /* the two-unit trial: the last P/2 samples of the window */
            xw = xc - ((P / 2) >> 3) * 10;                                      /* -> x[s - P/2] */
#pragma unroll
            for (int j = 0; j < 12; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xw + SL_XPAD(j)); w[j] = v.x; w[j + 1] = v.y; }
#pragma unroll
            for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); a1[j] = v.x; a1[j + 1] = v.y; }
#pragma unroll 1
            for (uint32_t k1 = 0; k1 < (uint32_t)(P / 2); k1 += 16) {
                SL_STEPB(0, a1, c1, k1)  SL_STEPB(1, a1, c1, k1 + 4)  SL_STEPB(2, a1, c1, k1 + 8)  SL_STEPB(3, a1, c1, k1 + 12)
            }
            {
                double ps1 = 0.0;
#pragma unroll
                for (int j = 0; j < FIR_SPL; j += 2) { ps1 += (s + j == 0) ? 0.0 : fabs(a1[j]); ps1 += fabs(a1[j + 1]); }
                ps1 = wave_sum_f64_lane63(ps1);
                if (last_lane) p.tsum[((size_t)job * LNN_MAXT + 1) * p.npart + part] = ps1;
            }
            if (NBIG == 3) {                                                    /* the four-unit trial: the last P/4 */
                xw = xc - ((P / 4) >> 3) * 10;
#pragma unroll
                for (int j = 0; j < 12; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xw + SL_XPAD(j)); w[j] = v.x; w[j + 1] = v.y; }
#pragma unroll
                for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); a2[j] = v.x; a2[j + 1] = v.y; }
#pragma unroll 1
                for (uint32_t k2 = 0; k2 < (uint32_t)(P / 4); k2 += 16) {
                    SL_STEPB(0, a2, c2, k2)  SL_STEPB(1, a2, c2, k2 + 4)  SL_STEPB(2, a2, c2, k2 + 8)  SL_STEPB(3, a2, c2, k2 + 12)
                }
                double ps2 = 0.0;
#pragma unroll
                for (int j = 0; j < FIR_SPL; j += 2) { ps2 += (s + j == 0) ? 0.0 : fabs(a2[j]); ps2 += fabs(a2[j + 1]); }
                ps2 = wave_sum_f64_lane63(ps2);
                if (last_lane) p.tsum[((size_t)job * LNN_MAXT + 2) * p.npart + part] = ps2;
            }
#undef SL_LOADB
#undef SL_STEPB
        } else {
        for (; q < (uint32_t)(P / 2); q += 16) {             /* the one-unit trial alone */
            SL_STEP1(0, q, ha0, ha1, hb0, hb1)      SL_STEP1(1, q + 4, hb0, hb1, ha0, ha1)
            SL_STEP1(2, q + 8, ha0, ha1, hb0, hb1)  SL_STEP1(3, q + 12, hb0, hb1, ha0, ha1)
        }
#pragma unroll
        for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); a1[j] = v.x; a1[j + 1] = v.y; }
        for (; q < (uint32_t)(NBIG == 3 ? 3 * P / 4 : P); q += 16) {             /* the two-unit trial joins */
            const uint32_t k1 = q - P / 2;
            SL_STEP2(0, q, k1, ha0, ha1, hb0, hb1)          SL_STEP2(1, q + 4, k1 + 4, hb0, hb1, ha0, ha1)
            SL_STEP2(2, q + 8, k1 + 8, ha0, ha1, hb0, hb1)  SL_STEP2(3, q + 12, k1 + 12, hb0, hb1, ha0, ha1)
        }
        if (NBIG == 3) {
#pragma unroll
            for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); a2[j] = v.x; a2[j + 1] = v.y; }
            for (; q < (uint32_t)P; q += 16) {                /* the four-unit trial joins */
                const uint32_t k1 = q - P / 2, k2 = q - 3 * P / 4;
                SL_STEP3(0, q, k1, k2, ha0, ha1, hb0, hb1)              SL_STEP3(1, q + 4, k1 + 4, k2 + 4, hb0, hb1, ha0, ha1)
                SL_STEP3(2, q + 8, k1 + 8, k2 + 8, ha0, ha1, hb0, hb1)  SL_STEP3(3, q + 12, k1 + 12, k2 + 12, hb0, hb1, ha0, ha1)
            }
        }
        }
#undef SL_LOAD
#undef SL_RING
#undef SL_T0
#undef SL_TF
#undef SL_STEP1
#undef SL_STEP2
#undef SL_STEP3
        if (!TWO) {
            double xo[FIR_SPL];
#pragma unroll
            for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); xo[j] = v.x; xo[j + 1] = v.y; }
            /* the one-unit trial's forward output (linne_network.c:165-210) straight from the registers, and its search term */
            double *dst = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S + s;
            double ps0 = 0.0, ps1 = 0.0, ps2 = 0.0;
#pragma unroll
            for (int j = 0; j < FIR_SPL; j += 2) {
                lnn_d2 o; o.x = (s + j == 0) ? xo[j] : (xo[j] + a0[j]); o.y = xo[j + 1] + a0[j + 1];
                *(lnn_d2 *)(dst + j) = o;
                ps0 += (s + j == 0) ? 0.0 : fabs(o.x); ps0 += fabs(o.y);
                ps1 += (s + j == 0) ? 0.0 : fabs(a1[j]); ps1 += fabs(a1[j + 1]);
                if (NBIG == 3) { ps2 += (s + j == 0) ? 0.0 : fabs(a2[j]); ps2 += fabs(a2[j + 1]); }
            }
            ps0 = wave_sum_f64_lane63(ps0); ps1 = wave_sum_f64_lane63(ps1);
            if (NBIG == 3) ps2 = wave_sum_f64_lane63(ps2);
            if (last_lane) {
                p.tsum[((size_t)job * LNN_MAXT + 0) * p.npart + part] = ps0;
                p.tsum[((size_t)job * LNN_MAXT + 1) * p.npart + part] = ps1;
                if (NBIG == 3) p.tsum[((size_t)job * LNN_MAXT + 2) * p.npart + part] = ps2;
            }
        }
    }

    /* ---------------- the small trials (16, 8, 4, 2, 1 taps) from one register window ---------------- */
    {
        double wv[24], xo[FIR_SPL];                          /* x[s - 16 .. s + 7] */
#pragma unroll
        for (int j = 0; j < FIR_SPL; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc + j); xo[j] = v.x; xo[j + 1] = v.y; }
#pragma unroll
        for (int j = 0; j < 8; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc - 20 + j); wv[j] = v.x; wv[j + 1] = v.y; }
#pragma unroll
        for (int j = 0; j < 8; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc - 10 + j); wv[8 + j] = v.x; wv[9 + j] = v.y; }
#pragma unroll
        for (int j = 0; j < 8; j++) wv[16 + j] = xo[j];
        const uint32_t fine_unit = s / (na / LNN_MAXU);      /* unit of sample s under the finest split (128 units; na is a multiple of 2048) */
#define SL_SMALL(TT, NP) { \
            const uint32_t unit = fine_unit >> (7u - (uint32_t)(NBIG + TT)); \
            const double *hb = hsm[TT] + (size_t)unit * NP; \
            double acc[FIR_SPL]; \
            _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) acc[j] = xo[j]; \
            _Pragma("unroll") for (int k = 0; k < NP; k++) { const double h_ = hb[k]; \
                _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) acc[j] = __builtin_fma(h_, wv[16 + j - NP + k], acc[j]); } \
            double ps = (s == 0) ? 0.0 : fabs(acc[0]); \
            _Pragma("unroll") for (int j = 1; j < FIR_SPL; j++) ps += fabs(acc[j]); \
            ps = wave_sum_f64_lane63(ps); \
            if (last_lane) p.tsum[((size_t)job * LNN_MAXT + (NBIG + TT)) * p.npart + part] = ps; }
        SL_SMALL(0, 16) SL_SMALL(1, 8) SL_SMALL(2, 4) SL_SMALL(3, 2) SL_SMALL(4, 1)
#undef SL_SMALL
    }
}

#undef SL_XPAD
#endif

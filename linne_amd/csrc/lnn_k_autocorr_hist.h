/* lnn_k_autocorr_hist.h -- the long layer's lags with lanes = jobs: k_autocorr_hist (trials whose order exceeds 32: the
 * history lives in an LDS ring) and k_autocorr_sub (its shorter trials: autocorr_rows' register rings).
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_AUTOCORR_HIST_H_INCLUDED
#define LNN_K_AUTOCORR_HIST_H_INCLUDED

/* k_autocorr_hist<P, TT>: trial TT (order PT = P >> TT >= 64) of 64 jobs per block, lanes = jobs.
 *
 * The block turns the trial's input into the PADDED WINDOWED STREAM of k_autocorr2 -- each unit's n windowed samples followed
 * by `pad` zeros (pad = PT: the longest lag) -- 16 positions at a time, transposed into an LDS ring of D + 16 positions x 64
 * rows (coalesced loads, the window applied once for all waves; the first 16 positions are mirrored behind the ring so that a
 * 16-position read never wraps).  Wave w owns LPW lags from J0 = LPW * w on: per position it reads the stream value v[m] and the
 * DELAYED value v[m - J0] (one ds_read each), keeps the last 16 delayed values in a register ring, and adds ring[m-J0-j] * v[m]
 * to lag J0+j -- the reference's products in the reference's order; the pairs that reach across a unit's end meet the zeros
 * (+0.0 added).  No per-wave window generator, no stream bookkeeping in the inner loop: 2 LPW multiply/adds and 2 LDS reads per
 * position and wave. */
/* 16 bytes per lane from global memory straight into LDS: lane l's land at lds_wave_base + 16 l (global_load_lds_dwordx4, the LDS
 * base in M0).  Inline assembly on purpose: through the builtin the compiler waits for the load (vmcnt) in front of the next LDS read
 * of ANY array -- it cannot tell the landing zone from the ring the lags are read from -- and the prefetch is gone; here the one
 * s_waitcnt sits where the data is needed (commit). */
__device__ __forceinline__ void lds_dma16(const void *g, void *lds_wave_base)
{
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base;
    uint32_t keep;                                                  /* M0 is the compiler's: handed back as it was (the s_nop: a scalar write of M0 needs a wait state before an LDS-DMA load reads it, and nothing inserts it inside an asm) */
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(base), "v"(g) : "memory");
}
template <int N> __device__ __forceinline__ void hist_wait_loads(double (&r)[N])
{
    static_assert(N == 9 || N == 11, "one operand list per lag count");
    if constexpr (N == 9)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]) : : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]) : : "memory");
}
/* lags per wave (measured): order 128 -- 11 (12 waves, one block per CU); order 64 -- 9 (8 waves, two blocks per CU) */
#define HIST_LPW(PT_) ((PT_) >= 128 ? 11 : 9)
#define HIST_TILE(PT_) 16                                    /* positions per tile (a multiple of 16; 32 was measured: far slower for order 128) */
#define HIST_WAVES(PT_) (((PT_) + HIST_LPW(PT_)) / HIST_LPW(PT_))
template <int P, int TT>
__global__ __launch_bounds__(64 * HIST_WAVES(P >> TT), ((P >> TT) >= 128) ? 3 : 4) void k_autocorr_hist(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int PT = P >> TT, NLAG = PT + 1, LPW = HIST_LPW(PT), NW = HIST_WAVES(PT), T = HIST_TILE(PT), D = PT + 2 * T, NLD = T / 2, RPI = 128 / T;   /* RPI: rows per load instruction */
    constexpr int NSLOT = (NLD + NW - 1) / NW;                /* tile load instructions per wave */
    static_assert(LPW <= 16, "a wave's lags come from a 16-deep register ring");
    static_assert(PT % T == 0 && D % T == 0, "tiles must not straddle a unit's end or the ring's end");
    __shared__ __attribute__((aligned(16))) double ring_lds[D + T][65];
    __shared__ __attribute__((aligned(16))) lnn_d2 stage[NLD][64];      /* the tile on its way from memory: load instruction k's 64 x 16 bytes, as the lanes asked for them */
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const RowRuns &rr = p.runs[1];
    const uint32_t b = gridDim.x - 1u - blockIdx.x;
    uint32_t run = 0;
    while (run + 1 < rr.n && b >= rr.blk_begin[run + 1]) run++;
    const uint32_t row0 = rr.row_begin[run] + (b - rr.blk_begin[run]) * 64u, nrows = rr.row_begin[run + 1];
    uint32_t myrow = row0 + lane; if (myrow >= nrows) myrow = nrows - 1;
    const bool store = (row0 + lane) < nrows;
    const uint32_t ci = p.cls_of_frame[(myrow / p.R) / p.C];
    const uint32_t ci0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
    const DevClass &c0 = p.cls[ci0];
    if (!__all(ci == ci0) || !hist_takes(p, layer, c0)) return;      /* uniform for the block: k_autocorr2 has these rows */
    const uint32_t na = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.na);
    const uint32_t n = na >> TT, upl = n + (uint32_t)PT, units = 1u << TT, total = units * upl, ntiles = total / T;
    const double *wt = p.wtab + (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.wt_off[layer][TT]);
    const uint32_t J0 = wave * (uint32_t)LPW, JN = (J0 + (uint32_t)LPW <= (uint32_t)NLAG) ? (uint32_t)LPW : (uint32_t)NLAG - J0;
    double *out = p.acorr + ((size_t)myrow * LNN_MAXT + TT) * LNN_ACW + J0;
    /* zero the ring: the stream before position 0 */
    for (uint32_t i = threadIdx.x; i < (uint32_t)(D + T) * 65u; i += blockDim.x) (&ring_lds[0][0])[i] = 0.0;
    /* tile loads: instruction k covers rows RPI k + lane/(T/2), positions 2(lane%(T/2))..+1; wave w issues the instructions w, w + NW, ... */
    const uint32_t lrow = lane / (uint32_t)(T / 2), lsmp = 2u * (lane % (uint32_t)(T / 2));
    const double *src[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; i++) {
        const uint32_t k = wave + (uint32_t)i * NW;
        uint32_t lr = row0 + (uint32_t)RPI * (k < (uint32_t)NLD ? k : 0u) + lrow; if (lr >= nrows) lr = nrows - 1;
        src[i] = p.sig + ((size_t)lr * 2 + cur) * p.S + lsmp;
    }
    double r[LPW], q[LPW], hist[16];
#pragma unroll
    for (int j = 0; j < LPW; j++) { r[j] = 0.0; q[j] = 0.0; }
#pragma unroll
    for (int j = 0; j < 16; j++) hist[j] = 0.0;
    /* the tile being fetched: place in the padded stream */
    uint32_t f_unit = 0, f_loc = 0;                          /* unit and place inside the padded unit of the NEXT tile to fetch */
    uint32_t p_loc = 0;                                       /* place of the tile that is on its way */
    /* issue() only REQUESTS the tile's samples -- straight into LDS (global_load_lds_dwordx4: 16 bytes per lane to stage[k][lane],
     * no register holds them on the way); they are windowed when commit() moves them into the ring, three quarters of a tile's work
     * later.  (Round 4: the window multiply sat in issue() -- the wave then waited out every load on the spot, a trip to memory per
     * tile with the whole block at the barrier behind it; kept in registers instead, the samples in flight pushed the order-64 kernel
     * over its 128 registers and the compiler parked them in scratch, waiting for them just the same.)  The Welch weight of a sample is
     * computed where it is needed, with the host table's own two multiplies (build_classes: w[loc] = div * (double)h *
     * (double)(nu - 1 - h), h = min(loc, nu - 1 - loc): lpc.c:199-204) -- the same bits, and no second load to wait for. */
    const double wdiv = c0.trial_div[layer][TT];
    (void)wt;
    auto issue = [&]() {                                      /* request the tile at (f_unit, f_loc); zeros in the pad */
        p_loc = f_loc;
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            const uint32_t k = wave + (uint32_t)i * NW;
            if (k < (uint32_t)NLD && f_loc < n)
                lds_dma16(src[i] + (size_t)f_unit * n + f_loc, &stage[k][0]);
        }
        f_loc += T; if (f_loc == upl) { f_loc = 0; f_unit++; }
    };
    auto welch = [&](uint32_t loc) -> double {               /* (loc < n: inside the unit) */
        const uint32_t h = (loc < (n >> 1)) ? loc : (n - 1u - loc);
        return wdiv * (double)h * (double)(n - 1u - h);
    };
    auto commit = [&](uint32_t slot0) {                       /* slot0: ring slot of the tile's first position (multiple of T) */
        /* the tile has landed in `stage` (what this wave asked for: each lane reads its own 16 bytes back).  The wait takes the lags'
         * accumulators as operands: nothing of their arithmetic touches memory, and without a data dependence the compiler moves the
         * wait up to the top of the tile, in front of all of it */
        hist_wait_loads(r);
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            const uint32_t k = wave + (uint32_t)i * NW;
            if (k < (uint32_t)NLD) {
                const uint32_t r = (uint32_t)RPI * k + lrow;
                double vx = 0.0, vy = 0.0;
                if (p_loc < n) { const lnn_d2 x = stage[k][lane]; vx = x.x * welch(p_loc + lsmp); vy = x.y * welch(p_loc + lsmp + 1u); }
                ring_lds[slot0 + lsmp][r] = vx; ring_lds[slot0 + lsmp + 1][r] = vy;
                if (slot0 == 0) { ring_lds[D + lsmp][r] = vx; ring_lds[D + lsmp + 1][r] = vy; }
            }
        }
    };
    __syncthreads();                                          /* ring zeroed */
    issue(); commit(0);
    __syncthreads();
    /* ring slots of the current tile and of the delayed one (position - J0, modulo D; the first D positions before the stream
     * are the zeros written above) */
    uint32_t slot_c = 0, slot_d = (uint32_t)((D - (int)(J0 % (uint32_t)D)) % D);
    auto read_group = [&](uint32_t sc, uint32_t sd, double *cv, double *dv) {
#pragma unroll
        for (int i = 0; i < 4; i++) { cv[i] = ring_lds[sc + i][lane]; dv[i] = ring_lds[sd + i][lane]; }
    };
    double cv[4], dv[4];
    read_group(slot_c, slot_d, cv, dv);
    uint32_t a_unit = 0, a_loc = 0;                           /* the tile being accumulated */
#pragma unroll 1
    for (uint32_t ti = 0; ti < ntiles; ti++) {
        if (ti + 1 < ntiles) issue();
#pragma unroll
        for (int g = 0; g < T / 4; g++) {
            double ncv[4], ndv[4];
            if (g == T / 4 - 1) {                             /* the next group lies in the next tile: publish it */
                uint32_t nslot = slot_c + T; if (nslot == (uint32_t)D) nslot = 0;
                if (ti + 1 < ntiles) commit(nslot);
                __syncthreads();
                uint32_t nd = slot_d + T; if (nd >= (uint32_t)D) nd -= (uint32_t)D;
                read_group(nslot, nd, ncv, ndv);              /* past the last tile: stale data, never used */
                slot_c = nslot; slot_d = nd;
            } else read_group(slot_c + 4 * (g + 1), slot_d + 4 * (g + 1), ncv, ndv);
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int j = 0; j < LPW; j++) r[j] += q[j];                                /* adds of the previous position's products */
                hist[(4 * g + i) % 16] = dv[i];
                const double v = cv[i];
#pragma unroll
                for (int j = 0; j < LPW; j++) q[j] = hist[((4 * g + i - j) % 16 + 16) % 16] * v;   /* (the last wave's lags beyond the order: computed, not stored) */
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { cv[i] = ncv[i]; dv[i] = ndv[i]; }
        }
        a_loc += T;
        if (a_loc == upl) {                                   /* the unit and its zeros are through: store its lags */
#pragma unroll
            for (int j = 0; j < LPW; j++) { r[j] += q[j]; q[j] = 0.0; }
            if (store) {
                double *o = out + (size_t)a_unit * NLAG;
#pragma unroll
                for (int j = 0; j < LPW; j++) if ((uint32_t)j < JN) o[j] = r[j];
            }
#pragma unroll
            for (int j = 0; j < LPW; j++) r[j] = 0.0;
            a_loc = 0; a_unit++;
        }
    }
}

/* k_autocorr_sub<P>: the trials of orders 32 .. 1 of a layer of order P >= 64, for the frames hist_takes: autocorr_rows
 * with its register rings, 10 waves per 64 jobs (lags per wave: 9 8 8 8 | 9 8 | 9 | 5 | 3 | 2). */
template <int P>
__global__ __launch_bounds__(640, 3) void k_autocorr_sub(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int NT = AcCfg<P>::NT, NW = 10, T0 = NT - 6;     /* T0: the trial of order 32 */
    __shared__ double tile[2][32][65];
    __shared__ __attribute__((aligned(16))) double wts_mem[2 * NT * 32];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const RowRuns &rr = p.runs[1];
    const uint32_t b = gridDim.x - 1u - blockIdx.x;
    uint32_t run = 0;
    while (run + 1 < rr.n && b >= rr.blk_begin[run + 1]) run++;
    const uint32_t row0 = rr.row_begin[run] + (b - rr.blk_begin[run]) * 64u, nrows = rr.row_begin[run + 1];
    uint32_t row = row0 + lane; if (row >= nrows) row = nrows - 1;
    const uint32_t ci = p.cls_of_frame[(row / p.R) / p.C];
    const uint32_t ci0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
    const DevClass &c0 = p.cls[ci0];
    if (!__all(ci == ci0) || !hist_takes(p, layer, c0)) return;
    const uint32_t na = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.na);
    double (*wtile)[32] = (double (*)[32])wts_mem;
#define SUB_RUN(T_, J0_, JN_) autocorr_rows<P, false, T_, J0_, JN_, NW>(p, layer, cur, row0, nrows, 1u, na, c0, wave, lane, tile, wtile)
    switch (wave) {
        case 0: SUB_RUN(T0, 0, 9); break;      case 1: SUB_RUN(T0, 9, 8); break;      case 2: SUB_RUN(T0, 17, 8); break;   case 3: SUB_RUN(T0, 25, 8); break;
        case 4: SUB_RUN(T0 + 1, 0, 9); break;  case 5: SUB_RUN(T0 + 1, 9, 8); break;  case 6: SUB_RUN(T0 + 2, 0, 9); break;
        case 7: SUB_RUN(T0 + 3, 0, 5); break;  case 8: SUB_RUN(T0 + 4, 0, 3); break;  default: SUB_RUN(T0 + 5, 0, 2); break;
    }
#undef SUB_RUN
}

#endif

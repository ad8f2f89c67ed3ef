/* lnn_k_fwdloss.h -- k_fwd_loss: forward pass of the LAST layer fused with the ordered L1 loss of its output.
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_FWDLOSS_H_INCLUDED
#define LNN_K_FWDLOSS_H_INCLUDED

/* The last layer's forward output (linne_network.c:165-210) is needed for one thing only: the regulariser pass's loss, the
 * mean of |out[s]| added in sample order (linne_network.c:609-616).  Writing it to HBM with one kernel and reading it back
 * with another (k_fir2<1> + k_chain_sum<1>) moves 160 KB per job for 8 bytes of result, so for layers of <= 16 taps this
 * kernel does both with lanes = jobs: a wave streams the inputs of 64 jobs through a small transposed LDS tile (coalesced 16-byte
 * loads; lane = job reads), every lane runs its job's FIR on a register ring of the last P inputs and adds |x + predict| to
 * its own chain.  Nothing but the loss is written.
 *
 * Bit-exactness: a lane's coefficient vector always has P slots, the chosen unit's np = P / units coefficients in the LAST np
 * of them and zeros in front -- the reference's predict = 0.0; predict += h[k] * x[s - np + k] (k ascending) then starts with
 * products that are +-0.0, and 0.0 + (+-0.0) = +0.0: the chain is the reference's from its first real term on.  The same
 * holds for the taps before sample 0 (history 0.0; the reference skips them).  Sample 0 passes through as x[0] + 0.0.
 * The unit (and with it the coefficients) changes at multiples of the job's unit length, lane by lane.  The kernel takes the
 * jobs whose unit lengths are all multiples of 4 samples (analysis length a multiple of 4P: every full CLI block and most
 * tails) -- a unit then starts at a group of 4 samples; the other jobs are left to k_fir2<1> + k_chain_sum, which skip the
 * ones done here (fwd_loss_takes). */
template <int P>
__global__ __launch_bounds__(64, 2) void k_fwd_loss(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int T = 16, NLD = 8;                              /* a tile: 16 samples of 64 rows, eight 16-byte loads per lane */
    __shared__ __attribute__((aligned(16))) double xt[2][T][65];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64, nrows = p.J;
    uint32_t job = row0 + lane;
    bool mine = job < nrows;
    if (!mine) job = nrows - 1;
    uint32_t my_na = job_class(p, job).na;
    if (!fwd_loss_takes(p, layer, my_na)) mine = false;
    if (!mine) my_na = 0;
    uint32_t na_blk = my_na;                                    /* uniform loop bound: the longest row of the block */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)na_blk, o); na_blk = v > na_blk ? v : na_blk; }
    na_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)na_blk);
    if (na_blk == 0) return;
    const uint32_t units = p.lunits[(size_t)job * LNN_MAXL + layer], np = (uint32_t)P / units, n = mine ? my_na / units : 0xFFFFFFFFu;
    const double *hsrc = p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP;
    /* tile loads: instruction i covers rows 8i + lane/8, samples 2(lane%8)..+1 */
    const uint32_t lrow = lane >> 3, lsmp = 2u * (lane & 7u);
    const size_t rstride = (size_t)2 * p.S;                     /* doubles between consecutive jobs' inputs */
    const double *src0 = p.sig + (size_t)cur * p.S + lsmp;
    lnn_d2 pre[NLD];
    auto issue = [&](uint32_t tile_idx) {                       /* rows are S >= na_blk long; S is even: a pair never leaves its row */
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const uint32_t s = tile_idx * T + lsmp;
            uint32_t r = row0 + 8u * (uint32_t)i + lrow; if (r >= nrows) r = nrows - 1;
            lnn_d2 z; z.x = 0.0; z.y = 0.0;
            pre[i] = (s + 1 < p.S) ? *(const lnn_d2 *)(src0 + (size_t)r * rstride + (size_t)tile_idx * T) : z;
        }
    };
    auto commit = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < NLD; i++) { const uint32_t r = 8u * (uint32_t)i + lrow; xt[buf][lsmp][r] = pre[i].x; xt[buf][lsmp + 1][r] = pre[i].y; }
    };
    double hv[P], xw[P];
#pragma unroll
    for (int k = 0; k < P; k++) xw[k] = 0.0;
    uint32_t unit = 0, next_b = 0;                              /* first sample of my next unit (the first one starts at 0) */
    uint32_t nb_u = 0;                                          /* the earliest next_b of the wave: uniform */
    auto wave_min = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w < v ? w : v; }
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    if (!mine) next_b = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < P; k++) hv[k] = 0.0;
    double sum = 0.0;
    const uint32_t ntiles = (na_blk + T - 1) / T;
    issue(0); commit(0);
    __syncthreads();
#pragma unroll 1
    for (uint32_t ti = 0; ti < ntiles; ti++) {
        const uint32_t buf = ti & 1u;
        if (ti + 1 < ntiles) issue(ti + 1);
#pragma unroll
        for (int g = 0; g < T / 4; g++) {                       /* four samples at a time: four independent tap chains */
            const uint32_t s0 = ti * T + 4u * (uint32_t)g;
            if (s0 == nb_u) {                                   /* some lanes start a unit here (a few times per frame): their np coefficients
                                                                 * go into the last np slots, zeros in front */
                if (next_b == s0) {
#pragma unroll
                    for (int k = 0; k < P; k++) {
                        const int kk = k - (int)((uint32_t)P - np);
                        hv[k] = (kk >= 0) ? hsrc[unit * np + (uint32_t)(kk >= 0 ? kk : 0)] : 0.0;
                    }
                    unit++;
                    next_b = (unit < units) ? next_b + n : 0xFFFFFFFFu;
                }
                nb_u = wave_min(next_b);
            }
            double xs[4], pr[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int j = 0; j < 4; j++) xs[j] = xt[buf][4 * g + j][lane];
            /* sample j's taps x[s-P+k] sit in ring slots (s+k) % P, except those among this group's own samples */
#pragma unroll
            for (int k = 0; k < P; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int back = P - k;                     /* the tap lies `back` samples before sample j */
                    const double xv = (back <= j) ? xs[j - back] : xw[(4 * g + j + k) % P];
                    pr[j] += hv[k] * xv;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) xw[(4 * g + j) % P] = xs[j];
            if (s0 < my_na) {                                   /* na is a multiple of 8: a group is inside or outside */
#pragma unroll
                for (int j = 0; j < 4; j++) sum += fabs(xs[j] + pr[j]);
            }
        }
        if (ti + 1 < ntiles) commit(buf ^ 1u);
        __syncthreads();
    }
    if (mine) p.jloss[job] = sum / (double)my_na;
}

/* k_fwd_loss_mw: k_fwd_loss for chunks of FEW jobs (24 576 .. 65 535: a group of EncodeWhole).  With a wave per 64 jobs such a chunk is
 * fewer waves than the chip has SIMDs, each alone on its own, and a lone wave issues an instruction every ~9 cycles whatever its
 * dependences: 10 240 samples x 35 instructions = 1.4 ms however few the jobs (0.53 ms is the chunk's share of the large batch's 2.1).
 * Only the SUM over the frame is a chain; the filter outputs are independent.  So a block is 64 jobs x FIVE waves and walks the frames in
 * super-tiles of 64 samples: waves 0 .. 3 filter 16 samples each (lane = job, k_fwd_loss's arithmetic on a register ring primed with the
 * 16 samples in front of theirs) and leave |x + predict| in LDS; wave 4 adds the 64 magnitudes of the super-tile to its lanes' chains in
 * sample order -- the reference's sum, term by term -- while the others are already on the next one.  Two barriers per super-tile.
 * Same jobs as k_fwd_loss (fwd_loss_takes), same bits. */
#define FLM_FIR 4                                              /* filter waves */
template <int P>
__global__ __launch_bounds__(64 * (FLM_FIR + 1), 2) void k_fwd_loss_mw(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int T = 16, ST = T * FLM_FIR, NLD = 8;            /* a wave's tile: 16 samples of 64 rows, eight 16-byte loads per lane; a super-tile: 64 samples */
    static_assert(T % P == 0, "the ring index of a tile's sample is a constant");
    __shared__ __attribute__((aligned(16))) double xt[T + ST][65];      /* [16 samples of history | the super-tile][row] */
    __shared__ __attribute__((aligned(16))) double ob[ST][65];          /* |x + predict| of the super-tile */
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t row0 = blockIdx.x * 64, nrows = p.J;
    uint32_t job = row0 + lane;
    bool mine = job < nrows;
    if (!mine) job = nrows - 1;
    uint32_t my_na = job_class(p, job).na;
    if (!fwd_loss_takes(p, layer, my_na)) mine = false;
    if (!mine) my_na = 0;
    uint32_t na_blk = my_na;                                    /* uniform loop bound: the longest row of the block */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)na_blk, o); na_blk = v > na_blk ? v : na_blk; }
    na_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)na_blk);
    if (na_blk == 0) return;
    const uint32_t nst = (na_blk + ST - 1) / ST;                /* super-tiles */
    if (wave == (uint32_t)FLM_FIR) {
        /* ---- the chain: lane = job adds the super-tile's 64 magnitudes in sample order ---- */
        double sum = 0.0;
#pragma unroll 1
        for (uint32_t k = 0; k < nst; k++) {
            __syncthreads();                                    /* (B: the filter waves have their tiles) */
            __syncthreads();                                    /* (C: the magnitudes of super-tile k are in LDS) */
#pragma unroll 16
            for (int i = 0; i < ST; i++) sum += ob[i][lane];
        }
        __syncthreads();                                        /* (the filter waves' last B) */
        if (mine) p.jloss[job] = sum / (double)my_na;
        return;
    }
    /* ---- a filter wave: samples 64 k + 16 wave .. + 15 of every super-tile ---- */
    const uint32_t units = p.lunits[(size_t)job * LNN_MAXL + layer], np = (uint32_t)P / units, n = mine ? my_na / units : 0xFFFFFFFFu;
    const double *hsrc = p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP;
    const uint32_t lrow = lane >> 3, lsmp = 2u * (lane & 7u);   /* tile loads: instruction i covers rows 8i + lane/8, samples 2(lane%8)..+1 */
    const size_t rstride = (size_t)2 * p.S;
    const double *src0 = p.sig + (size_t)cur * p.S + lsmp;
    lnn_d2 pre[NLD];
    auto issue = [&](uint32_t k) {                              /* my tile of super-tile k (rows are S >= na_blk long; S is even) */
        const uint32_t s = k * ST + wave * T + lsmp;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            uint32_t r = row0 + 8u * (uint32_t)i + lrow; if (r >= nrows) r = nrows - 1;
            lnn_d2 z; z.x = 0.0; z.y = 0.0;
            pre[i] = (s + 1 < p.S) ? *(const lnn_d2 *)(src0 + (size_t)r * rstride + (size_t)(k * ST + wave * T)) : z;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; i++) { const uint32_t r = 8u * (uint32_t)i + lrow; xt[T + wave * T + lsmp][r] = pre[i].x; xt[T + wave * T + lsmp + 1][r] = pre[i].y; }
    };
    double hv[P];
#pragma unroll
    for (int k = 0; k < P; k++) hv[k] = 0.0;
    uint32_t uend = mine ? 0u : 0xFFFFFFFFu;                    /* first sample behind the unit whose coefficients hv holds (0: none yet) */
    if (wave == 0u) {                                           /* zeros in front of sample 0 */
#pragma unroll
        for (int i = 0; i < T; i++) xt[i][lane] = 0.0;
    }
    issue(0);
#pragma unroll 1
    for (uint32_t k = 0; k < nst; k++) {
        if (k > 0u && wave == (uint32_t)(FLM_FIR - 1)) {        /* the last 16 samples of the super-tile before become the history (mine: nobody else touches these rows) */
#pragma unroll
            for (int i = 0; i < T; i++) xt[i][lane] = xt[ST + i][lane];
        }
        commit();
        __syncthreads();                                        /* B: the super-tile is in LDS (and the chain has read the magnitudes of the one before) */
        if (k + 1 < nst) issue(k + 1);
        double xw[P];
#pragma unroll
        for (int i = 0; i < P; i++) xw[i] = xt[wave * T + (T - P) + i][lane];      /* the P samples in front of my tile: ring slot (sample index) % P, and my tile starts at a multiple of 16 */
#pragma unroll
        for (int g = 0; g < T / 4; g++) {                       /* four samples at a time: four independent tap chains */
            const uint32_t s0 = k * ST + wave * T + 4u * (uint32_t)g;
            if (__any(s0 >= uend)) {                            /* some lanes are in another unit than the one whose coefficients they hold (units start at multiples of 4 samples; the ones
                                                                 * that started inside the other waves' tiles are found here too): its np coefficients go into the last np slots, zeros in front */
                if (s0 >= uend) {
                    const uint32_t unit = s0 / n;
                    if (unit < units) {
#pragma unroll
                        for (int c = 0; c < P; c++) {
                            const int kk = c - (int)((uint32_t)P - np);
                            hv[c] = (kk >= 0) ? hsrc[unit * np + (uint32_t)(kk >= 0 ? kk : 0)] : 0.0;
                        }
                        uend = (unit + 1u) * n;
                    } else uend = 0xFFFFFFFFu;                  /* (behind the frame: nothing of it is added) */
                }
            }
            double xs[4], pr[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int j = 0; j < 4; j++) xs[j] = xt[T + wave * T + 4 * g + j][lane];
#pragma unroll
            for (int c = 0; c < P; c++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int back = P - c;                     /* the tap lies `back` samples before sample j */
                    const double xv = (back <= j) ? xs[j - back] : xw[(4 * g + j + c) % P];
                    pr[j] += hv[c] * xv;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) xw[(4 * g + j) % P] = xs[j];
            const bool inside = s0 < my_na;                     /* na is a multiple of 8: a group is inside or outside (+0.0 leaves the chain as it is) */
#pragma unroll
            for (int j = 0; j < 4; j++) ob[wave * T + 4 * g + j][lane] = inside ? fabs(xs[j] + pr[j]) : 0.0;
        }
        __syncthreads();                                        /* C: the magnitudes are in LDS; everybody is through with the super-tile's samples */
    }
    __syncthreads();                                            /* (the chain's last B) */
}
#undef FLM_FIR

#endif

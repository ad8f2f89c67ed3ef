/* lnn_k_lastlayer.h -- k_last_layer: the LAST layer's unit-count search, forward pass and loss in ONE pass over its input.
 * Part of the single translation unit lnn_device.hip (included there, behind lnn_k_fwdloss.h); not a stand-alone header. */
#ifndef LNN_K_LASTLAYER_H_INCLUDED
#define LNN_K_LASTLAYER_H_INCLUDED

/* What the last layer (<= 16 taps) needs is a handful of numbers per job: for every trial -- u = 1, 2, .., P units of P / u taps --
 * the mean of the search's |((x + p0) + p1) + ...| (linne_network.c:318-335; the frame's first sample counts 0.0), the strict-<
 * argmin of these (:338-341), and the winner's forward loss, the mean of |x + predict| with predict summed from 0.0
 * (:165-210, :609-616).  Three kernels made them: the certified search (k_fir_small: order-free sums on fused multiply-adds, 342 KB
 * per channel-frame read), the selection with its exact fallback, and k_fwd_loss over the same 332 KB again -- which runs at the
 * HBM's pace (4.9 TB/s).  The two sums of a trial share their PRODUCTS (h[k] * x[s - np + k], a separate multiply in both), and with
 * lanes = jobs a sum over the frame is a chain in the lane that owns the job: so one kernel in k_fwd_loss's form reads the input
 * once, runs for each of the NT trials the two chains on the same np products -- the reference's operations in the reference's
 * order: the EXACT search, no certificate, nothing left for a fallback -- and writes 2 NT means per job; k_select (exact = 2) takes
 * the argmin and hands the winner's forward loss on.  31 multiplies + 62 adds + 15 for the terms per sample (P = 16) against the 72
 * multiply-adds of the fused search + 40 of the forward pass, one pass over HBM instead of two.
 *
 * A lane keeps its trials' coefficients of its current unit in registers (2 P - 1 doubles over the block's two waves: trial t's np = P >> t
 * at 2 P - (2 P >> t));
 * units change at multiples of the finest unit (na / P samples, a multiple of 4: fwd_loss_takes), trial t's at every 2^(NT-1-t)-th of
 * them.  Zero history in front of sample 0 stands in for the taps the reference skips there (+-0.0 products: a chain's magnitude is
 * unchanged).  Takes the chunks k_fwd_loss takes whole (every frame's analysis length a multiple of 4 P) whose frames all have
 * every trial (the host checks: last_layer_all). */
/* the walk of one wave over its 64 jobs' frames for the trials TLO .. THI - 1; NW waves of a block share the tiles (NW = 2 -- the one-unit
 * trial in one wave, the others in a second: half the instructions and registers each -- was measured: 5.96 ms against 5.65 for the 60-minute
 * batch: the kernel is bound by the FP64 unit, 93 operations a sample, not by what a wave can issue) */
template <int P, int NT, int TLO, int THI, int NW>
__device__ __forceinline__ void last_layer_walk(const Plan &p, double (*xt)[16][65], const uint32_t lane, const uint32_t wave, const uint32_t row0, const uint32_t nrows,
        const uint32_t job, const bool mine, const uint32_t my_na, const uint32_t na_blk, const uint32_t cur)
{
    constexpr int T = 16, NLD = 8 / NW;                         /* a tile: 16 samples of 64 rows, eight 16-byte loads per lane, shared out over the block's waves */
    constexpr int HOFF = 2 * P - ((2 * P) >> TLO), NH = (2 * P - ((2 * P) >> THI)) - HOFF;      /* my trials' coefficients: trial t's np = P >> t at 2 P - (2 P >> t) */
    const double *tc = p.tcoef + (size_t)job * LNN_MAXT * LNN_MAXP;      /* trial t, unit un, tap k (the tap of x[s - np + k]): tc[t MAXP + un np + k] */
    /* tile loads: instruction i of wave w covers rows 8 (NLD w + i) + lane / 8, samples 2 (lane % 8) .. + 1 */
    const uint32_t lrow = lane >> 3, lsmp = 2u * (lane & 7u);
    const size_t rstride = (size_t)2 * p.S;                     /* doubles between consecutive jobs' inputs */
    const double *src0 = p.sig + (size_t)cur * p.S + lsmp;
    lnn_d2 pre[NLD];
    auto issue = [&](uint32_t tile_idx) {                       /* rows are S >= na_blk long; S is even: a pair never leaves its row */
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const uint32_t s = tile_idx * T + lsmp;
            uint32_t r = row0 + 8u * ((uint32_t)NLD * wave + (uint32_t)i) + lrow; if (r >= nrows) r = nrows - 1;
            lnn_d2 z; z.x = 0.0; z.y = 0.0;
            pre[i] = (s + 1 < p.S) ? *(const lnn_d2 *)(src0 + (size_t)r * rstride + (size_t)tile_idx * T) : z;
        }
    };
    auto commit = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < NLD; i++) { const uint32_t r = 8u * ((uint32_t)NLD * wave + (uint32_t)i) + lrow; xt[buf][lsmp][r] = pre[i].x; xt[buf][lsmp + 1][r] = pre[i].y; }
    };
    double hv[NH], xw[P], sa[THI - TLO], sb[THI - TLO];
#pragma unroll
    for (int k = 0; k < NH; k++) hv[k] = 0.0;
#pragma unroll
    for (int k = 0; k < P; k++) xw[k] = 0.0;
#pragma unroll
    for (int t = 0; t < THI - TLO; t++) { sa[t] = 0.0; sb[t] = 0.0; }
    const uint32_t nf = my_na >> (NT - 1);                      /* the finest unit */
    uint32_t ub = 0, next_b = mine ? 0u : 0xFFFFFFFFu;          /* index and first sample of my next finest unit */
    uint32_t nb_u = 0;                                          /* the earliest next_b of the wave: uniform */
    auto wave_min = [&](uint32_t v) -> uint32_t {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w < v ? w : v; }
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    const uint32_t ntiles = (na_blk + T - 1) / T;
    issue(0); commit(0);
    __syncthreads();
#pragma unroll 1
    for (uint32_t ti = 0; ti < ntiles; ti++) {
        const uint32_t buf = ti & 1u;
        if (ti + 1 < ntiles) issue(ti + 1);
#pragma unroll
        for (int g = 0; g < T / 4; g++) {                       /* four samples at a time */
            const uint32_t s0 = ti * T + 4u * (uint32_t)g;
            if (s0 == nb_u) {                                   /* some lanes start a finest unit here (P times per frame): the trials whose unit starts with it load its coefficients */
                if (next_b == s0) {
#pragma unroll
                    for (int t = TLO; t < THI; t++) {
                        const int np = P >> t, off = 2 * P - ((2 * P) >> t) - HOFF, sh = NT - 1 - t;
                        if ((ub & ((1u << sh) - 1u)) == 0u) {
                            const double *h = tc + (size_t)t * LNN_MAXP + (size_t)(ub >> sh) * (uint32_t)np;
#pragma unroll
                            for (int k = 0; k < np; k++) hv[off + k] = h[k];
                        }
                    }
                    ub++;
                    next_b = (ub < (uint32_t)P) ? next_b + nf : 0xFFFFFFFFu;
                }
                nb_u = wave_min(next_b);
            }
            double xs[4];
#pragma unroll
            for (int j = 0; j < 4; j++) xs[j] = xt[buf][4 * g + j][lane];
            const bool inside = s0 < my_na;                     /* na is a multiple of 4 P: a group is inside or outside */
#pragma unroll
            for (int t = TLO; t < THI; t++) {
                const int np = P >> t, off = 2 * P - ((2 * P) >> t) - HOFF;
                double a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { a[j] = xs[j]; b[j] = 0.0; }
                /* sample j's tap k is x[s - np + k]: among this group's own samples, or in ring slot (its index) % P */
#pragma unroll
                for (int k = 0; k < np; k++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int back = np - k;
                        const double xv = (back <= j) ? xs[j - back] : xw[(4 * g + j - back + 2 * P) % P];
                        const double prod = hv[off + k] * xv;
                        a[j] = a[j] + prod; b[j] = b[j] + prod;
                    }
                }
                if (inside) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        sa[t - TLO] += (s0 + (uint32_t)j == 0u) ? 0.0 : fabs(a[j]);      /* the search's term (the frame's first sample: 0.0, linne_network.c:318-335) */
                        sb[t - TLO] += fabs(xs[j] + b[j]);                              /* the forward output's magnitude */
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) xw[(4 * g + j) % P] = xs[j];
        }
        if (ti + 1 < ntiles) commit(buf ^ 1u);
        __syncthreads();
    }
    if (mine) {
#pragma unroll
        for (int t = TLO; t < THI; t++) {
            p.tloss[(size_t)job * LNN_MAXT + t] = sa[t - TLO] / (double)my_na;
            p.tsum[((size_t)job * LNN_MAXT + t) * p.npart] = sb[t - TLO] / (double)my_na;      /* (the search's partial sums are not used on this path: the slot carries trial t's forward loss to k_select) */
        }
    }
}

template <int P>
__global__ __launch_bounds__(64, 2) void k_last_layer(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int NT = (P == 16) ? 5 : (P == 8) ? 4 : (P == 4) ? 3 : 2;        /* trials: 1, 2, .., P units */
    static_assert(16 % P == 0 && (1 << (NT - 1)) == P, "the ring index of a tile's sample is a constant; one trial per power of two");
    __shared__ __attribute__((aligned(16))) double xt[2][16][65];
    const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64, nrows = p.J;
    uint32_t job = row0 + lane;
    const bool mine = job < nrows;
    if (!mine) job = nrows - 1;
    const uint32_t my_na = mine ? job_class(p, job).na : 0u;
    uint32_t na_blk = my_na;                                    /* uniform loop bound: the longest row of the block */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)na_blk, o); na_blk = v > na_blk ? v : na_blk; }
    na_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)na_blk);
    if (na_blk == 0) return;
    (void)layer;
    last_layer_walk<P, NT, 0, NT, 1>(p, xt, lane, 0u, row0, nrows, job, mine, my_na, na_blk, cur);
}

#endif

/* lnn_k_fir.h -- double-precision FIR kernels (k_fir2), ordered sums (k_chain_sum) and the unit-count decision (k_select).
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_FIR_H_INCLUDED
#define LNN_K_FIR_H_INCLUDED

/* ------------------------------------------------------------------------------------------------
 * K_C / K_D (v2): the two double-precision FIR evaluations of a layer, register-blocked.
 *   MODE 0  trial residual magnitude for every unit-count trial (linne_network.c:318-335):
 *           residual = x[s]; residual += h[k]*x[s-p+k], k = 0..p-1; |residual| -> wx[job][trial][s]
 *   MODE 1  forward with the chosen unit count (linne_network.c:165-210):
 *           predict = 0; predict += h[k]*x[s-p+k]; out[s] = x[s] + predict
 * A lane owns 4 consecutive samples and slides a 4-wide register window over the taps: per 4 taps it issues
 * 16 unfused mul+add pairs against 4 LDS reads (2 of samples, 2 of coefficients).  Each sample's sum stays one
 * chain in increasing tap order.  Lanes whose 4 samples touch the start of the frame (taps are skipped there),
 * a unit boundary of a ragged tail frame, or p < 4 take the sample-at-a-time path.
 * ---------------------------------------------------------------------------------------------- */
/* order-free sum of one double per lane over the wavefront, result in lane 63 (DPP row shifts / broadcasts on the two
 * halves of the value: no LDS round trip, unlike __shfl_xor).  Only for sums whose order is free (the certified search). */
__device__ __forceinline__ double wave_sum_f64_lane63(double v)
{
#define LNN_DPP_ADD(CTRL, ROWMASK) { \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true); \
        v += __hiloint2double(hi_, lo_); }
    LNN_DPP_ADD(0x111, 0xf)   /* row_shr:1 */
    LNN_DPP_ADD(0x112, 0xf)   /* row_shr:2 */
    LNN_DPP_ADD(0x114, 0xf)   /* row_shr:4 */
    LNN_DPP_ADD(0x118, 0xf)   /* row_shr:8 */
    LNN_DPP_ADD(0x142, 0xa)   /* row_bcast:15 -> rows 1, 3 */
    LNN_DPP_ADD(0x143, 0xc)   /* row_bcast:31 -> rows 2, 3 */
#undef LNN_DPP_ADD
    return v;
}

__device__ __forceinline__ double wave_max_f64_lane63(double v)     /* max of non-negative values over the wavefront, in lane 63 (lanes read 0.0 where a row has no source) */
{
#define LNN_DPP_MAX(CTRL, ROWMASK) { \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true); \
        v = fmax(v, __hiloint2double(hi_, lo_)); }
    LNN_DPP_MAX(0x111, 0xf) LNN_DPP_MAX(0x112, 0xf) LNN_DPP_MAX(0x114, 0xf) LNN_DPP_MAX(0x118, 0xf) LNN_DPP_MAX(0x142, 0xa) LNN_DPP_MAX(0x143, 0xc)
#undef LNN_DPP_MAX
    return v;
}

#define FIR_THREADS 256
#define FIR_SPL     8                       /* consecutive samples per lane */
#define FIR_TILE    (FIR_THREADS * FIR_SPL)
static_assert(FIR_TILE == LNN_FIR_TILE, "search_long_takes assumes the tile of the FIR kernels");
template <int MODE, bool L0, bool SPEC>
__global__ __launch_bounds__(FIR_THREADS, (MODE == 0) ? 3 : 4) void k_fir2(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr uint32_t spec = SPEC ? 1u : 0u;        /* MODE 2: also write the one-unit trial's forward output; MODE 1: skip the jobs it covered */
    __shared__ __attribute__((aligned(16))) double xs[LNN_MAXP + FIR_TILE + 8];
    __shared__ __attribute__((aligned(16))) double hs[(MODE == 1) ? 1 : LNN_MAXT][LNN_MAXP + 8];   /* every trial's coefficients; +8: the pipelined loop reads one step ahead */
    __shared__ __attribute__((aligned(16))) double ob[(MODE == 2) ? 1 : FIR_THREADS / 64][(MODE == 2) ? 2 : 64 * FIR_SPL];   /* per-wave store transpose (MODE 0/1) */
    __shared__ double chain[LNN_MAXT];                              /* MODE 0: the ordered sums, carried across tiles */
    const uint32_t job = blockIdx.x + p.job_off, tid = threadIdx.x;          /* grid = (jobs, tiles): the job count is not bound by 65535 */
    if (MODE == 0 && !p.uncertain[job]) return;                     /* exact search only where the certified one gave up */
    /* speculation (layers whose search usually picks one unit): the search pass already wrote the forward output of the
     * one-unit trial -- same coefficients, same products, its own accumulation chain -- so such a job has nothing to do here */
    if (MODE == 1 && spec && p.lunits[(size_t)job * LNN_MAXL + layer] == 1u) return;
    const DevClass &c = job_class(p, job);
    const uint32_t na = c.na;
    if (MODE == 1 && fwd_loss_takes(p, layer, na)) return;         /* the last layer of this job is k_fwd_loss's */
    if (MODE == 2 && !L0 && SPEC && search_long_takes(p, layer, c)) return;     /* this job's search is k_search_long's */
    const uint32_t P = p.P[layer];
    const double *x = p.sig + ((size_t)job * 2 + cur) * p.S;
    const int32_t *xi = p.xint + (size_t)(job / p.R) * p.S;          /* layer 0 reads the pre-emphasised int32 channel (linne_encoder.c:661-663) */
    uint32_t ntr = (MODE != 1) ? c.ntrials[layer] : 1u;
    if (MODE == 2 && LNN_DBG_MAXTR(p) && ntr > LNN_DBG_MAXTR(p)) ntr = LNN_DBG_MAXTR(p);
    if (MODE == 0 && tid < LNN_MAXT) chain[tid] = 0.0;
    {
        const double *hsrc = (MODE != 1) ? (p.tcoef + (size_t)job * LNN_MAXT * LNN_MAXP) : (p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP);
        for (uint32_t i = tid; i < ntr * LNN_MAXP; i += FIR_THREADS) { const uint32_t tt = i / LNN_MAXP, k = i % LNN_MAXP; if (k < P) hs[tt][k] = hsrc[i]; }
    }
    if (MODE == 2 && blockIdx.y == 0) {                             /* per trial: the largest L1 norm of a unit's coefficients (search_slack) */
        __syncthreads();
        if (tid < ntr) {
            const uint32_t u = c.trial_u[layer][tid], np = P / u;
            double mx = 0.0;
            for (uint32_t un = 0; un < u; un++) { double a = 0.0; for (uint32_t k = 0; k < np; k++) a += fabs(hs[tid][un * np + k]); mx = fmax(mx, a); }
            p.thsum[(size_t)job * LNN_MAXT + tid] = mx;
        }
    }
    /* MODE 0 walks every tile of the job in order inside one block; MODE 1/2 take one tile per block */
    /* (MODE 1 in batches: grid.y = 1 and the block walks its job's tiles -- most jobs chose one unit and leave at once, and 620 k
     * blocks that only look up their job and go cost 1.7 ms of dispatch) */
    for (uint32_t s0 = (MODE == 0) ? 0u : blockIdx.y * FIR_TILE; s0 < na; s0 += (MODE == 0) ? FIR_TILE : (MODE == 1 ? gridDim.y * FIR_TILE : 0x40000000u)) {
    __syncthreads();
    if (!L0 && s0 >= LNN_MAXP && s0 + FIR_TILE + 8 <= na) {          /* interior tile: 16-byte loads (S, s0, MAXP are even) */
        for (uint32_t i = 2 * tid; i < LNN_MAXP + FIR_TILE + 8; i += 2 * FIR_THREADS)
            *(lnn_d2 *)(xs + i) = *(const lnn_d2 *)(x + (s0 - LNN_MAXP + i));
    } else {
        for (uint32_t i = tid; i < LNN_MAXP + FIR_TILE + 8; i += FIR_THREADS) {
            const int64_t g = (int64_t)s0 - LNN_MAXP + i;
            xs[i] = (g >= 0 && g < (int64_t)na) ? (L0 ? ((double)xi[g] * p.scale) : x[g]) : 0.0;
        }
    }
    const uint32_t s = s0 + FIR_SPL * tid;
    /* unit index of sample s under the finest split (128 units), one division per tile: a trial with u units (a power of
     * two) then finds its unit with a shift.  Only when the analysis length is a multiple of 128; else divide per trial. */
    const bool fine_ok = (na % LNN_MAXU) == 0;
    const uint32_t fine_unit = fine_ok ? s / (na / LNN_MAXU) : 0u;
    const double *xc = xs + LNN_MAXP + FIR_SPL * tid;                /* -> x[s], 16-byte aligned */
    /* coalesced store of a tile of layer output: the wave's 64*FIR_SPL consecutive results go through LDS so that consecutive
     * lanes write consecutive 16-byte pieces (a lane's own 8 results are 64 bytes apart from its neighbour's) */
    auto store_rows = [&](const double *vals) {
        const uint32_t wv = tid >> 6, ln = tid & 63u, wbase = s0 + wv * 64 * FIR_SPL;
        double *dst = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S;
        if (wbase < na) {
            if (s < na) {
#pragma unroll
                for (int j = 0; j < FIR_SPL; j += 2) { lnn_d2 v; v.x = vals[j]; v.y = vals[j + 1]; *(lnn_d2 *)(&ob[wv][ln * FIR_SPL + j]) = v; }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);       /* lgkmcnt(0): the wave's own LDS writes have landed */
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < FIR_SPL / 2; i++) {
                const uint32_t e = 2 * ln + 128 * i, g = wbase + e;
                if (g + 1 < na) *(lnn_d2 *)(dst + g) = *(const lnn_d2 *)(&ob[wv][e]);
                else if (g < na) dst[g] = ob[wv][e];
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    /* one trial of the tile; DUAL (a compile-time tag): MODE 2's trial 0 with SPEC, whose chain is also the forward output */
    auto trial_body = [&](auto dual_tag, const uint32_t t) {
        constexpr bool dual = decltype(dual_tag)::value;
        const uint32_t u = (MODE != 1) ? c.trial_u[layer][t] : p.lunits[(size_t)job * LNN_MAXL + layer];
        const uint32_t lgu = 31u - (uint32_t)__builtin_clz(u | 1u);   /* unit counts are powers of two (linne_network.c:289): shifts, not divisions */
        const uint32_t n = na >> lgu, np = P >> lgu;
        const double *hbuf = hs[t];
        if (t == 0) __syncthreads();                                 /* tile and coefficients are staged */
        double acc[FIR_SPL];
        double acc2[FIR_SPL];                                        /* MODE 2, trial 0 with `spec`: predict-first sums of the forward pass */
#pragma unroll
        for (int j = 0; j < FIR_SPL; j++) acc2[j] = 0.0;
        if (s < na) {
            /* all FIR_SPL samples in one unit, every tap present */
            const bool whole = ((n & (FIR_SPL - 1)) == 0) && (s >= np) && (s + FIR_SPL - 1 < na);
            if (whole && (np & 3u) == 0) {
                const uint32_t my_unit = fine_ok ? (fine_unit >> (7u - lgu)) : (s / n);
                const double *hb = hbuf + (size_t)my_unit * np;
                const double *xw = xc - np;                              /* -> x[s - np] */
                /* Window x[s-np+k .. +11] in a register ring of 16 (element e lives in w[e % 16]): a step of 4 taps reads
                 * 11 of them, the LDS reads of the next step's 4 new samples and coefficients land in the free quarter
                 * while the 32 multiply-adds of this step issue, and nothing is ever moved. */
                double w[16];
                static_assert(FIR_SPL == 8, "the ring below is laid out for 8 samples per lane");
#pragma unroll
                for (int j = 0; j < 12; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xw + j); w[j] = v.x; w[j + 1] = v.y; }
#pragma unroll
                for (int j = 0; j < FIR_SPL; j++) acc[j] = (MODE != 1 && !dual) ? xc[j] : 0.0;     /* dual: the forward pass's chain (from 0.0) serves both */
                lnn_d2 ha0 = *(const lnn_d2 *)(hb), ha1 = *(const lnn_d2 *)(hb + 2), hb0, hb1;
                uint32_t k = 0;
#define FIR_STEP(G, HC0, HC1, HN0, HN1) { \
                    const lnn_d2 na_ = *(const lnn_d2 *)(xw + k + 12), nb_ = *(const lnn_d2 *)(xw + k + 14);   /* in bounds: xs/hs are padded */ \
                    HN0 = *(const lnn_d2 *)(hb + k + 4); HN1 = *(const lnn_d2 *)(hb + k + 6); \
                    w[(4 * G + 12) % 16] = na_.x; w[(4 * G + 13) % 16] = na_.y; w[(4 * G + 14) % 16] = nb_.x; w[(4 * G + 15) % 16] = nb_.y; \
                    const double hh_[4] = { HC0.x, HC0.y, HC1.x, HC1.y }; \
                    _Pragma("unroll") for (int kk = 0; kk < 4; kk++) { \
                        _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) acc[j] = FUSE_ ? __builtin_fma(hh_[kk], w[(4 * G + kk + j) % 16], acc[j]) : (acc[j] + hh_[kk] * w[(4 * G + kk + j) % 16]); } \
                    k += 4; }
                /* The search (MODE 2) only has to land inside the certificate's interval (k_select, search_slack), not on the
                 * reference's bits: its trials run on fused multiply-adds -- half the instructions.  The chain that doubles as the
                 * forward output (dual), the exact fallback and the forward pass keep the reference's separate multiply and add. */
                if (MODE == 2 && !dual) {
                    constexpr bool FUSE_ = true;
                    for (;;) {
                        FIR_STEP(0, ha0, ha1, hb0, hb1); if (k >= np) break;
                        FIR_STEP(1, hb0, hb1, ha0, ha1); if (k >= np) break;
                        FIR_STEP(2, ha0, ha1, hb0, hb1); if (k >= np) break;
                        FIR_STEP(3, hb0, hb1, ha0, ha1); if (k >= np) break;
                    }
                } else {
                    constexpr bool FUSE_ = false;
                    for (;;) {
                        FIR_STEP(0, ha0, ha1, hb0, hb1); if (k >= np) break;
                        FIR_STEP(1, hb0, hb1, ha0, ha1); if (k >= np) break;
                        FIR_STEP(2, ha0, ha1, hb0, hb1); if (k >= np) break;
                        FIR_STEP(3, hb0, hb1, ha0, ha1); if (k >= np) break;
                    }
                }
                if (dual) {
                    /* One chain for two results: acc is the forward pass's predict sum (linne_network.c:165-210), x + predict its
                     * output -- and |x + predict| stands in for the search's |((x + p0) + p1) + ...| (linne_network.c:318-335).
                     * The two differ by rounding only, within 2 gamma_(np+1) (|x| + sum |h_k x_k|) per sample; k_select widens
                     * this trial's interval by that bound (search_slack) before it certifies the argmin. */
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j++) { acc2[j] = acc[j]; acc[j] = xc[j] + acc[j]; }
                }
#undef FIR_STEP
            } else if (whole && np <= 2) {
                const uint32_t my_unit = fine_ok ? (fine_unit >> (7u - lgu)) : (s / n);
                const double *hb = hbuf + (size_t)my_unit * np;
                const double h0 = hb[0];
                if (np == 1) {
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j++) { acc[j] = (MODE != 1) ? xc[j] : 0.0; const double p0_ = h0 * xc[j - 1]; acc[j] += p0_; acc2[j] += p0_; }
                } else {
                    const double h1 = hb[1];
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j++) { acc[j] = (MODE != 1) ? xc[j] : 0.0; const double p0_ = h0 * xc[j - 2], p1_ = h1 * xc[j - 1]; acc[j] += p0_; acc[j] += p1_; acc2[j] += p0_; acc2[j] += p1_; }
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < FIR_SPL; j++) {
                    const uint32_t sj = s + j;
                    double v = (MODE != 1) ? xc[j] : 0.0, v2 = 0.0;
                    if (sj < na && sj != 0) {
                        const double *hb = hbuf + (size_t)(sj / n) * np;
                        const uint32_t kstart = (sj < np) ? (np - sj) : 0;  /* taps before sample 0 are skipped */
                        for (uint32_t k = kstart; k < np; k++) { const double pr_ = hb[k] * xc[(int)j - (int)np + (int)k]; v += pr_; v2 += pr_; }
                    }
                    acc[j] = v; acc2[j] = v2;
                }
            }
            if (dual) {                                              /* the forward output of the one-unit trial (linne_network.c:165-210): straight from
                                                                      * the registers -- a lane's 8 results are 64 contiguous bytes; routing them through
                                                                      * an LDS transpose like MODE 1 costs this kernel more in occupancy than it saves */
                double *dst = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S + s;
#pragma unroll
                for (int j = 0; j < FIR_SPL; j += 2) {
                    const double o0 = (s + j == 0) ? xc[j] : (xc[j] + acc2[j]), o1 = xc[j + 1] + acc2[j + 1];
                    if (s + j + 1 < na) { lnn_d2 v; v.x = o0; v.y = o1; *(lnn_d2 *)(dst + j) = v; }
                    else if (s + j < na) dst[j] = o0;
                }
            }
            /* results: |residual| (MODE 0) or x + predict (MODE 1) */
#pragma unroll
            for (int j = 0; j < FIR_SPL; j++) {
                if (MODE != 1) { double av = fabs(acc[j]); if (j == 0 && s == 0) av = 0.0; acc[j] = av; }     /* s is a multiple of 8: only j = 0 can be the frame's first sample */
                else { const double xv = xc[j]; acc[j] = (s + j == 0) ? xv : (xv + acc[j]); }
            }
        }
        if (MODE == 2) {
            /* order-free partial sum of this wave's |residual| values; the exact ordered chain is evaluated later only for
             * jobs whose argmin these sums cannot certify (k_select) */
            double ps = 0.0;
            if (s < na) {                                            /* na and s are multiples of 8: a lane's samples are all inside or all outside */
#pragma unroll
                for (int j = 0; j < FIR_SPL; j++) ps += acc[j];
            }
            ps = wave_sum_f64_lane63(ps);
            if ((tid & 63u) == 63u) p.tsum[((size_t)job * LNN_MAXT + t) * p.npart + blockIdx.y * (FIR_THREADS / 64) + (tid >> 6)] = ps;
            if (t == 0) {                                            /* max |x| of the wave's samples, for search_slack */
                double mx = 0.0;
                if (s < na) {
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j++) mx = fmax(mx, fabs(xc[j]));
                }
                mx = wave_max_f64_lane63(mx);
                if ((tid & 63u) == 63u) p.txmax[(size_t)job * p.npart + blockIdx.y * (FIR_THREADS / 64) + (tid >> 6)] = mx;
            }
        } else if (MODE == 0) {
            /* exact path: the tile's |residual| values go to LDS in sample order and ONE lane adds them to the trial's
             * running sum, continuing the single chain of linne_network.c:326-337 across tiles */
            if (s < na) {
#pragma unroll
                for (int j = 0; j < FIR_SPL; j++) ob[0][FIR_SPL * tid + j] = (s + j < na) ? acc[j] : 0.0;
            }
            __syncthreads();
            if (tid == 0) {
                const uint32_t cnt = (na - s0 < FIR_TILE) ? (na - s0) : FIR_TILE;
                double v = chain[t];
                for (uint32_t i = 0; i < cnt; i++) v += ob[0][i];
                chain[t] = v;
            }
            __syncthreads();
        } else {
            store_rows(acc);
        }
    };
    if (MODE == 2 && SPEC) {
        trial_body(std::true_type{}, 0u);
        for (uint32_t t = 1; t < ntr; t++) trial_body(std::false_type{}, t);
    } else {
        for (uint32_t t = 0; t < ntr; t++) trial_body(std::false_type{}, t);
    }
    }
    if (MODE == 0) {
        __syncthreads();
        if (tid < ntr) p.tloss[(size_t)job * LNN_MAXT + tid] = chain[tid] / (double)na;
    }
}

/* Unit-count search of a SHORT layer (P <= 16 taps in all): like k_fir2<2>, but a lane keeps its 8 samples and the P
 * before them in registers and evaluates every trial from there -- one window load per tile instead of one per trial, the
 * tap loops fully unrolled.  A tile whose lanes all see whole units and full history (every tile of a 10240-sample frame
 * but the first lanes of the first) takes this path; other lanes fall back to the sample-at-a-time form.  SPEC: also writes
 * the one-unit trial's forward output (see k_fir2). */
template <int P, bool L0, bool SPEC>
__global__ __launch_bounds__(FIR_THREADS, 4) void k_fir_small(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int NT = (P >= 16) ? 5 : (P >= 8) ? 4 : (P >= 4) ? 3 : 2;
    constexpr int HP = (P < 2) ? 2 : P;                           /* history kept in front of the tile (even, >= P) */
    __shared__ __attribute__((aligned(16))) double xs[HP + FIR_TILE];
    __shared__ __attribute__((aligned(16))) double hs_all[L0 ? LNN_MAXR : 1][NT][HP];    /* the coefficients of every job the block serves */
    __shared__ __attribute__((aligned(16))) double obuf[SPEC ? FIR_TILE : 2];      /* store transpose of the fused forward output */
    /* layer 0: the regulariser passes of a channel-frame read the same input, so one block stages the tile once and serves all
     * R jobs (grid.x = channel-frames); other layers: grid.x = jobs */
    const uint32_t nr = L0 ? p.R : 1u, job0 = blockIdx.x * nr, tid = threadIdx.x, s0 = blockIdx.y * FIR_TILE;
    const DevClass &c = job_class(p, job0);
    const uint32_t na = c.na;
    if (s0 >= na) return;
    const double *x = p.sig + ((size_t)job0 * 2 + cur) * p.S;
    const int32_t *xi = p.xint + (size_t)(job0 / p.R) * p.S;
    const uint32_t ntr = c.ntrials[layer];
    /* all jobs' coefficients are requested with the tile (fetched job by job behind a barrier they cost the block a trip to memory per
     * job: 76 % of the layer-0 kernel's wave cycles were waiting) */
    static_assert((L0 ? LNN_MAXR : 1) * NT * P <= 2 * FIR_THREADS, "two coefficient loads per thread");
    double hreg[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t i = tid + (uint32_t)q * FIR_THREADS;
        hreg[q] = (i < nr * NT * P) ? p.tcoef[((size_t)(job0 + i / (NT * P)) * LNN_MAXT + (i % (NT * P)) / P) * LNN_MAXP + i % P] : 0.0;
    }
    if (!L0 && s0 >= (uint32_t)HP && s0 + FIR_TILE <= na) {         /* an interior tile of doubles: 16-byte streaming loads (S, s0, HP are even) -- the last layer's search 4.31 -> 3.72 ms against 8-byte ones */
        for (uint32_t i = 2 * tid; i < HP + FIR_TILE; i += 2 * FIR_THREADS) *(lnn_d2 *)(xs + i) = __builtin_nontemporal_load((const lnn_d2 *)(x + (s0 - HP + i)));
    } else
    for (uint32_t i = tid; i < HP + FIR_TILE; i += FIR_THREADS) {
        const int64_t g = (int64_t)s0 - HP + i;
        xs[i] = (g >= 0 && g < (int64_t)na) ? (L0 ? ((double)xi[g] * p.scale) : x[g]) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t i = tid + (uint32_t)q * FIR_THREADS;
        if (i < nr * NT * P) hs_all[i / (NT * P)][(i % (NT * P)) / P][i % P] = hreg[q];
    }
    __syncthreads();
    const uint32_t s = s0 + FIR_SPL * tid;
    const double *xc = xs + HP + FIR_SPL * tid;                     /* -> x[s] */
    double wv[HP + FIR_SPL];                                        /* x[s - HP .. s + 7] */
#pragma unroll
    for (int i = 0; i < HP + FIR_SPL; i += 2) { const lnn_d2 v = *(const lnn_d2 *)(xc - HP + i); wv[i] = v.x; wv[i + 1] = v.y; }
    const bool live = s < na;
    /* A tile that lies whole inside a frame of whole 8-sample groups per finest unit (every tile of a 10 240-sample frame) takes the
     * register form in EVERY lane, as straight-line code without a per-lane test around each trial (round 4: those tests and their
     * exec-mask branches were a good part of the ~900 instructions a wave spent per tile on 248 multiply-adds).  The first lanes of
     * the first tile are no exception: the history in front of sample 0 is staged as zeros, and a tap on a zero adds +-0.0 to a chain
     * -- the reference's ramp (linne_network.c:320-327) term for term; only sample 0 itself, which the reference's loss leaves out,
     * is masked. */
    const bool tile_fast = ((na % (uint32_t)(FIR_SPL * P)) == 0) && (s0 + FIR_TILE <= na);          /* (uniform) */
    const bool fast = tile_fast || (live && (s >= (uint32_t)P) && (s + FIR_SPL - 1 < na) && ((na % (uint32_t)(FIR_SPL * P)) == 0));
    const uint32_t fine_unit = fast ? s / (na >> (NT - 1)) : 0u;   /* my samples' unit under the finest split (one division for all trials: the units nest) */
    for (uint32_t rr_ = 0; rr_ < nr; rr_++) {
    const uint32_t job = job0 + rr_;
    const double (*hs)[HP] = hs_all[rr_];
    double ps[NT], fwd[FIR_SPL];
#pragma unroll
    for (int j = 0; j < FIR_SPL; j++) fwd[j] = 0.0;
/* the register form of one trial (t and np are constants where it is expanded: the trial loops are unrolled) */
#define FS_REGISTER_FORM() do { \
                const double *hb = hs[t] + (size_t)(fine_unit >> (NT - 1 - t)) * np; \
                double h[(P >> 0)]; \
                _Pragma("unroll") for (int k = 0; k < np; k++) h[k] = hb[k]; \
                _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) { \
                    double term; \
                    if (SPEC && t == 0) {                           /* one chain for two results, as in k_fir2's dual form (search_slack) */ \
                        double acc2 = 0.0; \
                        _Pragma("unroll") for (int k = 0; k < np; k++) acc2 += h[k] * wv[HP - np + k + j]; \
                        fwd[j] = wv[HP + j] + acc2; \
                        term = fabs(fwd[j]); \
                    } else { \
                        double acc = wv[HP + j];                     /* fused multiply-adds: inside the certificate's interval (see k_fir2) */ \
                        _Pragma("unroll") for (int k = 0; k < np; k++) acc = __builtin_fma(h[k], wv[HP - np + k + j], acc); \
                        term = fabs(acc); \
                    } \
                    if (j == 0) term = (s == 0u) ? 0.0 : term;      /* sample 0 is not part of the loss (the reference's unit 0 starts at s = 1) */ \
                    sum += term; \
                } } while (0)
    if (tile_fast) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int np = P >> t;                                 /* a constant once the loop is unrolled */
            double sum = 0.0;
            if ((uint32_t)t < ntr) FS_REGISTER_FORM();
            ps[t] = sum;
        }
    } else
#pragma unroll
    for (int t = 0; t < NT; t++) {
        constexpr int dummy = 0; (void)dummy;
        const int np = P >> t;                                     /* a constant once the loop is unrolled */
        const uint32_t u = 1u << t, n = na / u;
        double sum = 0.0;
        if ((uint32_t)t < ntr && live) {
            if (fast) FS_REGISTER_FORM();
            else {
#pragma unroll 1
                for (int j = 0; j < FIR_SPL; j++) {
                    const uint32_t sj = s + j;
                    double v = xc[j], v2 = 0.0;
                    if (sj < na && sj != 0) {
                        const double *hb = hs[t] + (size_t)(sj / n) * np;
                        const uint32_t kstart = (sj < (uint32_t)np) ? ((uint32_t)np - sj) : 0;   /* taps before sample 0 are skipped */
                        for (uint32_t k = kstart; k < (uint32_t)np; k++) { const double pr = hb[k] * xc[(int)j - np + (int)k]; v += pr; v2 += pr; }
                    }
                    double av = (v > 0) ? v : -v;
                    if (sj == 0) av = 0.0;
                    if (sj < na) sum += av;
                    if (SPEC && t == 0) fwd[j] = (sj == 0) ? xc[j] : (xc[j] + v2);
                }
            }
        }
        ps[t] = sum;
    }
    if (SPEC) {
        /* forward output of the one-unit trial: a lane's 8 results are 64 bytes of the row, so the wave's 4 KB go through LDS and
         * leave as consecutive 16-byte pieces */
        double *ob = obuf + (size_t)(tid >> 6) * (64 * FIR_SPL);
        __builtin_amdgcn_wave_barrier();                            /* (the wave's reads of the previous job's pieces are done: in-order LDS) */
        const uint32_t ln = tid & 63u, wbase = s0 + (tid >> 6) * 64 * FIR_SPL;
#pragma unroll
        for (int j = 0; j < FIR_SPL; j += 2) { lnn_d2 v; v.x = fwd[j]; v.y = fwd[j + 1]; *(lnn_d2 *)(ob + ln * FIR_SPL + j) = v; }
        __builtin_amdgcn_s_waitcnt(0xC07F);                         /* lgkmcnt(0): the wave's own LDS writes have landed */
        __builtin_amdgcn_wave_barrier();
        double *dst = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S;
#pragma unroll
        for (int i = 0; i < FIR_SPL / 2; i++) {
            const uint32_t e = 2 * ln + 128 * i, g = wbase + e;
            if (g + 1 < na) __builtin_nontemporal_store(*(const lnn_d2 *)(ob + e), (lnn_d2 *)(dst + g));      /* (streaming: layer 0's kernel is its 330 KB of output per channel-frame -- 3.70 -> 3.13 ms; the same hint on k_search_long's output, on the decode kernels' stores and on the lanes = jobs kernels' loads changed nothing) */
            else if (g < na) dst[g] = ob[e];
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const double tot = wave_sum_f64_lane63(ps[t]);
        if ((uint32_t)t < ntr && (tid & 63u) == 63u) p.tsum[((size_t)job * LNN_MAXT + t) * p.npart + blockIdx.y * (FIR_THREADS / 64) + (tid >> 6)] = tot;
    }
    if (blockIdx.y == 0 && tid < ntr) {                             /* per trial: the largest L1 norm of a unit's coefficients (search_slack) */
        const uint32_t u = 1u << tid, np = (uint32_t)P >> tid;
        double mx = 0.0;
        for (uint32_t un = 0; un < u; un++) { double a = 0.0; for (uint32_t k = 0; k < np; k++) a += fabs(hs[tid][un * np + k]); mx = fmax(mx, a); }
        p.thsum[(size_t)job * LNN_MAXT + tid] = mx;
    }
    {                                                               /* max |x| of the wave's samples, for search_slack */
        double mx = 0.0;
        if (live) {
#pragma unroll
            for (int j = 0; j < FIR_SPL; j++) if (s + j < na) mx = fmax(mx, fabs(wv[HP + j]));
        }
        mx = wave_max_f64_lane63(mx);
        if ((tid & 63u) == 63u) p.txmax[(size_t)job * p.npart + blockIdx.y * (FIR_THREADS / 64) + (tid >> 6)] = mx;
    }
    }
}

/* ------------------------------------------------------------------------------------------------
 * ordered sums (v2): 64 chains per wavefront.  Rows are staged through LDS with coalesced loads and each lane then
 * adds its own row strictly in sample order, so every sum is the same single chain the reference evaluates.
 *   MODE 0  mean |residual| of each trial  (rows of wx)            (linne_network.c:326,334,337)
 *   MODE 1  L1 loss of the last layer's output (rows of sig, fabs) (linne_network.c:50-63)
 * ---------------------------------------------------------------------------------------------- */
#define SUM_THREADS 256
template <int MODE>
__global__ __launch_bounds__(SUM_THREADS) void k_chain_sum(Plan p, uint32_t layer, uint32_t cur)
{
    __shared__ double tile[2][64][65];
    __shared__ uint32_t row_na[64];
    __shared__ const double *row_ptr[64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, row0 = blockIdx.x * 64;
    const uint32_t nrows = (MODE == 0) ? p.J * LNN_MAXT : p.J;
    /* wave 0's lanes own the 64 chains; all four waves stage 16 rows each */
    uint32_t my_na = 0;
    if (wave == 0) {
        const uint32_t myrow = row0 + lane;
        const double *my_ptr = p.sig;               /* always dereferenceable */
        if (myrow < nrows) {
            const uint32_t job = (MODE == 0) ? myrow / LNN_MAXT : myrow;
            const DevClass &c = job_class(p, job);
            if ((MODE == 1 && !fwd_loss_takes(p, p.L - 1u, c.na)) || (MODE == 0 && (myrow % LNN_MAXT) < c.ntrials[layer] && p.uncertain[job])) {
                my_na = c.na;
                my_ptr = p.sig + ((size_t)job * 2 + cur) * p.S;
            }
        }
        row_na[lane] = my_na; row_ptr[lane] = my_ptr;
    }
    __syncthreads();
    uint32_t na_blk = 0;
    for (uint32_t i = 0; i < 64; i++) na_blk = row_na[i] > na_blk ? row_na[i] : na_blk;     /* uniform loop bound */
    const uint32_t ntiles = (na_blk + 63) / 64;
    double ld[16];
    auto fetch = [&](uint32_t tileidx) {            /* 16 unconditional loads in flight; zero beyond a row's end */
        const uint32_t sl = tileidx * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t r = wave * 16 + i, rn = row_na[r];
            const uint32_t idx = (sl < rn) ? sl : 0u;
            double v = row_ptr[r][idx];
            if (MODE == 1) v = fabs(v);
            ld[i] = (sl < rn) ? v : 0.0;            /* adding +0.0 leaves a non-negative chain unchanged */
        }
    };
    auto stash = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < 16; i++) tile[buf][wave * 16 + i][lane] = ld[i];
    };
    double sum = 0.0;
    if (ntiles) { fetch(0); stash(0); }
    __syncthreads();
    for (uint32_t k = 0; k < ntiles; k++) {
        if (k + 1 < ntiles) fetch(k + 1);
        if (wave == 0) {
#pragma unroll 16
            for (uint32_t j = 0; j < 64; j++) sum += tile[k & 1u][lane][j];
        }
        if (k + 1 < ntiles) stash((k + 1) & 1u);
        __syncthreads();
    }
    if (wave == 0 && my_na) {
        if (MODE == 0) p.tloss[row0 + lane] = sum / (double)my_na; else p.jloss[row0 + lane] = sum / (double)my_na;
    }
}

/* The same sums for a SMALL batch (a few frames: block-at-a-time calls): a wave per row.  It loads 256 samples of its row at a
 * time (coalesced; the next 256 are requested before these are added), leaves them in LDS and adds them in order -- every lane
 * runs the same chain.  A row takes ~40 us; the 64-rows-per-wave form above spends a trip to memory per 64-sample tile when
 * most of its rows are empty. */
template <int MODE>
__global__ __launch_bounds__(64) void k_chain_sum_wave(Plan p, uint32_t layer, uint32_t cur)
{
    __shared__ __attribute__((aligned(16))) double mag[256];
    const uint32_t lane = threadIdx.x, myrow = blockIdx.x;
    const uint32_t job = (MODE == 0) ? myrow / LNN_MAXT : myrow;
    const DevClass &c = job_class(p, job);
    if (!((MODE == 1 && !fwd_loss_takes(p, p.L - 1u, c.na)) || (MODE == 0 && (myrow % LNN_MAXT) < c.ntrials[layer] && p.uncertain[job]))) return;
    const uint32_t na = c.na;
    const double *src = p.sig + ((size_t)job * 2 + cur) * p.S;
    double nx[4], sum = 0.0;
#pragma unroll
    for (uint32_t r = 0; r < 4u; r++) { const uint32_t s = r * 64u + lane; const double v = (s < na) ? src[s] : 0.0; nx[r] = (MODE == 1) ? fabs(v) : v; }
    for (uint32_t s0 = 0; s0 < na; s0 += 256u) {
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++) mag[r * 64u + lane] = nx[r];
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++) { const uint32_t s = s0 + 256u + r * 64u + lane; const double v = (s < na) ? src[s] : 0.0; nx[r] = (MODE == 1) ? fabs(v) : v; }
        __builtin_amdgcn_wave_barrier();
        const uint32_t cnt = (na - s0 < 256u) ? (na - s0) : 256u;
        if (cnt == 256u) {
#pragma unroll 8
            for (uint32_t q = 0; q < 256u; q += 2u) { const lnn_d2 v = *(const lnn_d2 *)(mag + q); sum += v.x; sum += v.y; }
        } else for (uint32_t q = 0; q < cnt; q++) sum += mag[q];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { if (MODE == 0) p.tloss[myrow] = sum / (double)na; else p.jloss[myrow] = sum / (double)na; }
}

/* strict-< argmin over the trials (linne_network.c:338-341), keep its coefficients (== SetParameter,
 * :350-376, which recomputes the same values) and, for the last layer, the value the layer leaves in
 * parcor[P0] (Q2): the last call in reference order -- trials in order, then SetParameter's units -- that
 * wrote it. */
__global__ void k_select(Plan p, uint32_t layer, uint32_t exact)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= p.J) return;
    if (exact == 1u && !p.uncertain[job]) return;              /* (exact = 2: every job's ordered means are in tloss -- k_last_layer) */
    const DevClass &c = job_class(p, job);
    double min_loss = (double)FLT_MAX;
    uint32_t best = 0;
    const uint32_t nt = c.ntrials[layer];
    if (exact) {
        for (uint32_t t = 0; t < nt; t++) {
            const double l = p.tloss[(size_t)job * LNN_MAXT + t];
            if (l < min_loss) { min_loss = l; best = t; }
        }
    } else {
        /* Certified search.  m_t below is the mean of an order-free sum of the same non-negative terms the reference
         * adds sequentially; both sums are within gamma_n * S of the exact sum S, so they differ by at most
         * rel = (2 na + 8) * 2^-53 relatively.  If the smallest mean is separated from every other by more than that,
         * the reference's strict-< argmin (linne_network.c:338-341) is the same trial; otherwise the job is flagged and
         * the ordered chains are evaluated (k_fir2<0>, then k_select exact). */
        double m[LNN_MAXT];
        const double rel = (2.0 * (double)c.na + 8.0) * 1.1102230246251565e-16;
        int ok = 1;
        /* search_slack: the search kernels' terms are not the reference's |((x + p0) + p1) + ...| bit for bit -- the trial that
         * doubles as the forward output is |x + predict| with predict summed from 0.0, the others run on fused multiply-adds.
         * Each is a floating-point evaluation of the same np + 1 numbers' sum, like the reference's, within
         * gamma_(np+1) (|x| + sum |h_k x_k|) of the exact value: a term is off the reference's by at most
         * 2 gamma_(np+1) max|x| (1 + sum |h_k|) (taken with a factor 4 and np + 2 here), and so is the mean. */
        double slack[LNN_MAXT];
        {
            const uint32_t np_used = ((c.na + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64);
            const double *px = p.txmax + (size_t)job * p.npart;
            double xmax = 0.0;
            for (uint32_t i = 0; i < np_used; i++) xmax = fmax(xmax, px[i]);
            for (uint32_t t = 0; t < nt; t++) {
                const double npt = (double)(p.P[layer] / c.trial_u[layer][t]);
                slack[t] = 4.0 * (npt + 2.0) * 1.1102230246251565e-16 * xmax * (1.0 + p.thsum[(size_t)job * LNN_MAXT + t]);
                if (!(slack[t] >= 0.0) || !(slack[t] < (double)FLT_MAX)) ok = 0;
            }
        }
        for (uint32_t t = 0; t < nt; t++) {
            double sm = 0.0;
            const uint32_t np_used = ((c.na + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64);
            const double *ps = p.tsum + ((size_t)job * LNN_MAXT + t) * p.npart;
            for (uint32_t i = 0; i < np_used; i++) sm += ps[i];
            m[t] = sm / (double)c.na;
            if (!(m[t] >= 0.0) || !(m[t] < (double)FLT_MAX)) ok = 0;
            if (m[t] < min_loss) { min_loss = m[t]; best = t; }
        }
        {   /* the smallest mean's upper end must lie below every other mean's lower end */
            const double hi_best = min_loss * (1.0 + rel) + slack[best];
            double gap = (double)FLT_MAX;
            for (uint32_t t = 0; t < nt; t++) if (t != best) {
                const double lo = m[t] * (1.0 - rel) - slack[t];
                if (!(lo > hi_best)) ok = 0;
                gap = fmin(gap, lo - hi_best);
            }
            /* telemetry (LINNEAmd_GetLastMinMargin): the smallest certified gap of the call, relative to the winning mean --
             * how far real material stays from the certificate's bound.  Non-negative doubles order like their bit patterns. */
            if (ok && nt > 1 && min_loss > 0.0) atomicMin(p.min_margin, (unsigned long long)__double_as_longlong(gap / min_loss));
        }
        if (p.force_exact) ok = 0;                              /* LINNE_AMD_EXACT=1: every search takes the ordered chains */
        p.uncertain[job] = ok ? 0 : 1;
        if (!ok) atomicAdd(p.ucount, 1u);
    }
    const uint32_t P = p.P[layer];
    if (exact == 2u) p.jloss[job] = p.tsum[((size_t)job * LNN_MAXT + best) * p.npart];      /* the winner's forward loss, as k_last_layer left it */
    p.lunits[(size_t)job * LNN_MAXL + layer] = c.trial_u[layer][best];
    const double *h = p.tcoef + ((size_t)job * LNN_MAXT + best) * LNN_MAXP;
    double *dst = p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP;
    for (uint32_t k = 0; k < P; k++) dst[k] = h[k];
    if (layer + 1 == p.L) {
        double tail = 0.0; int set = 0;
        const uint32_t bu = c.trial_u[layer][best];
        for (int32_t unit = (int32_t)bu - 1; unit >= 0 && !set; unit--) {
            const size_t o = ((size_t)job * LNN_MAXT + best) * LNN_MAXU + unit;
            if (p.ptail_set[o]) { tail = p.ptail[o]; set = 1; }
        }
        for (int32_t t = (int32_t)c.ntrials[layer] - 1; t >= 0 && !set; t--)
            for (int32_t unit = (int32_t)c.trial_u[layer][t] - 1; unit >= 0 && !set; unit--) {
                const size_t o = ((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit;
                if (p.ptail_set[o]) { tail = p.ptail[o]; set = 1; }
            }
        p.jtail[job] = tail;
    }
}

/* k_select for a handful of jobs (block-at-a-time calls): a wave per job.  One thread per job walks ~180 partial sums and copies up
 * to 128 coefficients one dependent trip to memory after the other (41 us for the 128-tap layer of a stereo block); here the lanes
 * fetch everything at once into LDS, lane 0 takes the decision with k_select's own arithmetic in k_select's own order (the same
 * flags, the same telemetry), and all lanes copy the winner's coefficients. */
#define SELW_MAXPART 64u
__global__ __launch_bounds__(64) void k_select_wave(Plan p, uint32_t layer, uint32_t exact)
{
    __shared__ double s_sum[LNN_MAXT][SELW_MAXPART], s_xmax[SELW_MAXPART], s_loss[LNN_MAXT], s_hsum[LNN_MAXT];
    __shared__ uint32_t s_best;
    const uint32_t job = blockIdx.x, lane = threadIdx.x;
    if (exact && !p.uncertain[job]) return;
    const DevClass &c = job_class(p, job);
    const uint32_t nt = c.ntrials[layer];
    const uint32_t np_used = ((c.na + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64);       /* <= SELW_MAXPART: the host checks */
    if (exact) { if (lane < nt) s_loss[lane] = p.tloss[(size_t)job * LNN_MAXT + lane]; }
    else {
        for (uint32_t i = lane; i < nt * np_used; i += 64u) { const uint32_t t = i / np_used, k = i % np_used; s_sum[t][k] = p.tsum[((size_t)job * LNN_MAXT + t) * p.npart + k]; }
        if (lane < np_used) s_xmax[lane] = p.txmax[(size_t)job * p.npart + lane];
        if (lane < nt) s_hsum[lane] = p.thsum[(size_t)job * LNN_MAXT + lane];
    }
    __syncthreads();
    if (lane == 0) {
        double min_loss = (double)FLT_MAX;
        uint32_t best = 0;
        if (exact) {
            for (uint32_t t = 0; t < nt; t++) { const double l = s_loss[t]; if (l < min_loss) { min_loss = l; best = t; } }
        } else {        /* (k_select's certificate, statement for statement) */
            double m[LNN_MAXT], slack[LNN_MAXT];
            const double rel = (2.0 * (double)c.na + 8.0) * 1.1102230246251565e-16;
            int ok = 1;
            double xmax = 0.0;
            for (uint32_t i = 0; i < np_used; i++) xmax = fmax(xmax, s_xmax[i]);
            for (uint32_t t = 0; t < nt; t++) {
                const double npt = (double)(p.P[layer] / c.trial_u[layer][t]);
                slack[t] = 4.0 * (npt + 2.0) * 1.1102230246251565e-16 * xmax * (1.0 + s_hsum[t]);
                if (!(slack[t] >= 0.0) || !(slack[t] < (double)FLT_MAX)) ok = 0;
            }
            for (uint32_t t = 0; t < nt; t++) {
                double sm = 0.0;
                for (uint32_t i = 0; i < np_used; i++) sm += s_sum[t][i];
                m[t] = sm / (double)c.na;
                if (!(m[t] >= 0.0) || !(m[t] < (double)FLT_MAX)) ok = 0;
                if (m[t] < min_loss) { min_loss = m[t]; best = t; }
            }
            const double hi_best = min_loss * (1.0 + rel) + slack[best];
            double gap = (double)FLT_MAX;
            for (uint32_t t = 0; t < nt; t++) if (t != best) {
                const double lo = m[t] * (1.0 - rel) - slack[t];
                if (!(lo > hi_best)) ok = 0;
                gap = fmin(gap, lo - hi_best);
            }
            if (ok && nt > 1 && min_loss > 0.0) atomicMin(p.min_margin, (unsigned long long)__double_as_longlong(gap / min_loss));
            if (p.force_exact) ok = 0;
            p.uncertain[job] = ok ? 0 : 1;
            if (!ok) atomicAdd(p.ucount, 1u);
        }
        s_best = best;
        p.lunits[(size_t)job * LNN_MAXL + layer] = c.trial_u[layer][best];
        if (layer + 1 == p.L) {
            double tail = 0.0; int set = 0;
            const uint32_t bu = c.trial_u[layer][best];
            for (int32_t unit = (int32_t)bu - 1; unit >= 0 && !set; unit--) {
                const size_t o = ((size_t)job * LNN_MAXT + best) * LNN_MAXU + unit;
                if (p.ptail_set[o]) { tail = p.ptail[o]; set = 1; }
            }
            for (int32_t t = (int32_t)c.ntrials[layer] - 1; t >= 0 && !set; t--)
                for (int32_t unit = (int32_t)c.trial_u[layer][t] - 1; unit >= 0 && !set; unit--) {
                    const size_t o = ((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit;
                    if (p.ptail_set[o]) { tail = p.ptail[o]; set = 1; }
                }
            p.jtail[job] = tail;
        }
    }
    __syncthreads();
    const uint32_t P = p.P[layer], best = s_best;
    const double *h = p.tcoef + ((size_t)job * LNN_MAXT + best) * LNN_MAXP;
    double *dst = p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP;
    for (uint32_t k = lane; k < P; k += 64u) dst[k] = h[k];
}


#endif

/* lnn_dev_common.h -- shared device-side types, the launch plan and small device helpers.
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_DEV_COMMON_H_INCLUDED
#define LNN_DEV_COMMON_H_INCLUDED


#define LNN_MAXT        8       /* unit-count trials per layer: u = 1,2,...,128 */
#define LNN_MAXU        128
#define LNN_MAXP        128
#define LNN_MAXL        3
#define LNN_MAXR        4
#define LNN_MAXCLS      16
#define LNN_MAXCH       8
#define LNN_ACW         256     /* autocorrelation words per (job, trial): P + u <= 256 */
#define LNN_MAXSUB      8
#define LNN_META        8
typedef double lnn_d2 __attribute__((ext_vector_type(2)));
typedef int lnn_v4i __attribute__((ext_vector_type(4)));

/* one distinct frame length of a batch (full frames, the ragged tail, ...) */
struct DevClass {
    uint32_t n;                         /* valid samples                                             */
    uint32_t na;                        /* analysis length (linne_encoder.c:644-655)                 */
    uint32_t sin_off;                   /* offset of this class's SIN window table                   */
    uint32_t pad;
    uint32_t ntrials[LNN_MAXL];
    uint32_t trial_u[LNN_MAXL][LNN_MAXT];
    double   trial_div[LNN_MAXL][LNN_MAXT];   /* 4*pow(na/u - 1, -2) from the host libm (lpc.c:199)  */
    uint32_t wt_off[LNN_MAXL][LNN_MAXT];      /* offset of the trial's Welch weight table (padded unit: n + max(p,4) entries) */
};

/* Rows of a chunk (channel-frames, or jobs) cut into runs of one length class, so that the kernels that put 64 rows on
 * the lanes of a wave see blocks of one class and may take their wave-uniform fast paths; a ragged last frame gets a
 * block of its own.  Run i = rows [row_begin[i], row_begin[i+1]) = blocks [blk_begin[i], blk_begin[i+1]) of 64 rows.
 * The host hands the frames of a call to the kernels sorted by class (Plan.frame_map leads back to the caller's order), so a
 * chunk never has more runs than there are classes.  Only with the sort switched off (LINNE_AMD_SORT=0, a test knob) can a
 * chunk exceed LNN_MAXRUN runs: then one run covers everything (blocks may mix classes: slower, same results). */
#define LNN_MAXRUN LNN_MAXCLS
struct RowRuns { uint32_t n; uint32_t mixed; uint32_t row_begin[LNN_MAXRUN + 1]; uint32_t blk_begin[LNN_MAXRUN + 1]; };   /* mixed: the one-run fallback */

/* timing experiments (never in a release build): LNN_DBG_IS(p, v) is a compile-time false unless the library was built with
 * make EXPERIMENTS=1 */
#ifdef LNN_TIMING_EXPERIMENTS
#define LNN_DBG_IS(p, v) ((p).dbg_maxtr == (v))
#define LNN_DBG_MAXTR(p) ((p).dbg_maxtr)
#else
#define LNN_DBG_IS(p, v) false
#define LNN_DBG_MAXTR(p) 0u
#endif
struct Plan {
    uint32_t C, S, bits, L, R, ms, F, J;
    uint32_t hist;                      /* the long layer's lags come from k_autocorr_hist / k_autocorr_sub where they take the frame (LINNE_AMD_HIST, default 1) */
    uint32_t rows16;                    /* order-16 layers take the register-ring autocorrelation form (LINNE_AMD_ROWS16, default 1) */
    uint32_t search_long;               /* the long layer's search comes from k_search_long where it takes the job (LINNE_AMD_SEARCH_LONG, default 1) */
    uint32_t job_off;                   /* k_fir2: the launch covers the jobs job_off .. (its blockIdx.x is relative: the search of the frames k_search_long leaves) */
    uint32_t prep_general;              /* LINNE_AMD_PREP_GENERAL=1 (tests, A/B runs): k_prep streams the channel through global memory whatever its size */
    uint32_t prep_defer;                /* k_prep hands the channel-frames whose pre-emphasis correlations are not exact integers (loud 24-bit material) to k_prep_slow (lanes = channel-frames) instead of running their ordered chains on two lanes of its own block (LINNE_AMD_PREP_DEFER, default 1; needs S % 4 == 0) */
    uint32_t *prep_slow_n, *prep_slow_rows;   /* their count (one word, zeroed before k_prep) and rows f * C + ch of the chunk, in the order the blocks arrived */
    uint32_t fused_last;                /* the last layer's forward pass and loss come from k_fwd_loss where it takes the job (fwd_loss_takes) */
    RowRuns runs[2];                    /* [0] rows = channel-frames (layer 0), [1] rows = jobs */
    uint32_t P[LNN_MAXL], coef_off[LNN_MAXL];
    double regs[LNN_MAXR];
    double scale;                       /* 2^-(bits-1), exact */
    const int32_t *pcm; int32_t *resid; int32_t *prm; double *stats;
    uint32_t pcm16;                     /* how the caller staged the PCM: 0 int32, 1 int16 (<= 16-bit audio: half the H2D bytes), 2 packed little-endian 3-byte samples (<= 24 bits: three quarters) */
    const uint32_t *cls_of_frame; const uint32_t *frame_map; const DevClass *cls; const double *sintab; const double *wtab;
    int32_t *xint, *xtmp;               /* [F*C][S]                    */
    double *sig;                        /* [J][2][S]                   */
    double *acorr;                      /* [J][MAXT][ACW]              */
    double *tcoef;                      /* [J][MAXT][MAXP]  filter order (reversed LPC order) */
    double *ptail; uint8_t *ptail_set;  /* [J][MAXT][MAXU]             */
    double *tloss;                      /* [J][MAXT] exact mean |residual| (ordered chain)            */
    double *tsum;                       /* [J][MAXT][npart] per-wave partial sums of |residual| (certified search) */
    uint32_t npart;                     /* partial sums per (job, trial): tiles x waves per block      */
    double *txmax;                      /* [J][npart] per-wave max |input| of the layer (search_slack in k_select)               */
    double *thsum;                      /* [J][MAXT] per trial: the largest L1 norm of a unit's coefficients (search_slack)      */
    uint8_t *uncertain;                 /* [J] the order-free sums could not certify the argmin       */
    uint32_t *ucount;                   /* running count of such (job, layer) pairs of the call       */
    unsigned long long *min_margin;     /* bits of the smallest certified relative gap of the call (k_select)        */
    uint32_t dbg_maxtr;                 /* builds with LNN_TIMING_EXPERIMENTS only (make EXPERIMENTS=1; LINNE_AMD_DBG_MAXTR): switches parts of kernels off to time the rest -- results are wrong; a release build compiles every use out (LNN_DBG_IS) */
    uint32_t force_exact;               /* LINNE_AMD_EXACT: flag every search as uncertain              */
    double *lparams;                    /* [J][MAXL][MAXP]             */
    uint32_t *lunits;                   /* [J][MAXL]                   */
    double *jloss, *jtail;              /* [J]                         */
    /* -a N (auxiliary-function refinement, lpc.c:578-633): the FINAL pass of linne_network.c:605-630 runs as a plan of its own with
     * one job per channel-frame (R = 1) whose regulariser is the winner of the R search passes */
    const double *job_reg;              /* [J] regulariser of each job (NULL: regs[job % R])                                   */
    const uint32_t *af_best; const double *af_loss;   /* [J] winner of the search passes and its loss, for the stats record (NULL: from jloss) */
    double *af_a;                       /* [J][MAXP]   coefficients in LPC order, units back to back                           */
    double *af_inv;                     /* [J][S]      1 / max(|residual|, 1e-6) per sample                                    */
    double *af_R;                       /* [J][MAXP*MAXP] normal matrices, unit un of order np at un * np * np, row-major      */
    double *af_rv, *af_invd;            /* [J][MAXP]   right-hand sides / inverse diagonals, units back to back               */
    double *af_obj, *af_prev;           /* [J][MAXU]   objective of this / the previous iteration per unit                     */
    uint32_t *af_state;                 /* [J][MAXU]   0 iterating, 1 converged, 2 zero problem, 3 singular                    */
    uint32_t *af_prob, *af_nprob;       /* compact list of the (job, unit) problems of the layer (job * MAXU + unit), its length */
    double *af_pivot;                   /* [problems]  pivot sums out / pow(sum, -0.5) in (host libm), per Cholesky step       */
};

typedef const double __attribute__((address_space(4))) *lnn_cdp;    /* constant address space: wave-uniform loads become scalar loads */
#define LNN_FIR_TILE 2048u               /* samples per block of the search / forward kernels (lnn_k_fir.h FIR_TILE) */
/* does k_search_long (lnn_k_search.h) produce this job's unit-count search of `layer`?  The long layer (64 / 128 taps) of a preset
 * with a layer behind it, every trial present, the analysis length whole 2048-sample tiles (all full 10240-sample frames);
 * k_fir2<2> keeps the other frames */
__device__ __host__ __forceinline__ bool search_long_takes(const Plan &p, uint32_t layer, const DevClass &c)
{
    const uint32_t P = p.P[layer];
    uint32_t nt = 0; for (uint32_t u = 1; u <= P && u <= (uint32_t)LNN_MAXU; u <<= 1) nt++;
    return p.search_long && layer > 0 && layer + 1 < p.L && (P == 128u || P == 64u) && c.ntrials[layer] == nt && (c.na % LNN_FIR_TILE) == 0;
}

/* input sample i of the caller's PCM array */
__device__ __forceinline__ int32_t pcm24_at(const void *base, size_t i)
{
    const uint8_t *b = (const uint8_t *)base + 3u * i;
    return (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8)) | ((int32_t)(int8_t)b[2] << 16);
}
__device__ __forceinline__ int32_t pcm_at(const Plan &p, size_t i) { return p.pcm16 == 1u ? (int32_t)((const int16_t *)p.pcm)[i] : (p.pcm16 == 2u ? pcm24_at(p.pcm, i) : p.pcm[i]); }

/* does k_fwd_loss (lnn_k_fwdloss.h) produce this job's last-layer loss?  (na: the job's analysis length) */
__device__ __forceinline__ bool fwd_loss_takes(const Plan &p, uint32_t layer, uint32_t na) { return p.fused_last && layer + 1 == p.L && (na % (4u * p.P[layer])) == 0; }

/* Which frames k_autocorr_hist / k_autocorr_sub (lnn_k_autocorr_hist.h) take (the others stay with k_autocorr2, which skips
 * the ones taken there): every trial present,
 * every unit a whole number of 16-sample tiles, the finest unit at least one weight tile long, rows 16-byte aligned. */
__device__ __forceinline__ bool hist_takes(const Plan &p, uint32_t layer, const DevClass &c)
{
    const uint32_t P = p.P[layer];
    uint32_t nt = 0; for (uint32_t u = 1; u <= P && u <= (uint32_t)LNN_MAXU; u <<= 1) nt++;
    return p.hist && P >= 64u && c.ntrials[layer] == nt && (c.na % (16u << (nt - 1))) == 0 && (c.na >> (nt - 1)) >= 32u && (p.S & 3u) == 0;
}

/* ------------------------------------------------------------------------------------------------
 * small device helpers
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ double round_away(double d) { return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5); }   /* lpc.c:49-52 */
__device__ __forceinline__ int32_t mulshr5(int32_t x, int32_t c) { return (int32_t)((uint32_t)x * (uint32_t)c) >> 5; }

/* Levinson-Durbin, lpc.c:252-324, on a private array a[0..order+1]; r[1..order] are the lags, r0 the
 * ridge-scaled lag 0 (lpc.c:358).  The reference's u/v vectors are the old a and its mirror:
 * a_new[i] = u[i] + gamma*v[i] with u = (1,a1..ak,0), v = (0,ak..a1,1), so the update is done in place on
 * pairs (i, k+1-i).  On return a[1..order] are the LPC coefficients.  parcor_out (optional) gets
 * parcor[0..order-1] exactly as the reference writes them. */
__device__ void levinson(const double *r, double r0, uint32_t order, double *a, double *parcor_out)
{
    for (uint32_t i = 0; i < order + 2; i++) a[i] = 0.0;
    a[0] = 1.0;
    double ek = r0;
    a[1] = -r[1] / r0;
    if (parcor_out) parcor_out[0] = r[1] / ek;
    ek += r[1] * a[1];
    for (uint32_t k = 1; k < order; k++) {
        double gamma = 0.0;
        for (uint32_t i = 0; i < k + 1; i++) gamma += a[i] * r[k + 1 - i];
        gamma /= -ek;
        ek *= (1.0 - gamma * gamma);
        const double a0 = 1.0 + gamma * 0.0;              /* u[0]   + gamma*v[0]   */
        const double ak1 = 0.0 + gamma * 1.0;             /* u[k+1] + gamma*v[k+1] */
        uint32_t i = 1, j = k;
        while (i < j) {
            const double ai = a[i], aj = a[j];
            a[i] = ai + gamma * aj;
            a[j] = aj + gamma * ai;
            i++; j--;
        }
        if (i == j) { const double ai = a[i]; a[i] = ai + gamma * ai; }
        a[0] = a0; a[k + 1] = ak1;
        if (parcor_out) parcor_out[k] = -gamma;
    }
}


#endif

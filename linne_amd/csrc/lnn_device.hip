/*
 * lnn_device.hip -- gfx950 (MI355X) kernels of the LINNE per-frame prediction path and their C-ABI
 * launchers (declared in include/linne_amd.h).
 *
 * Bit-exactness rules (DESIGN.md "Arithmetic contract"):
 *   - this TU is compiled with -ffp-contract=off: every double multiply and add is a separate, correctly
 *     rounded IEEE-754 operation, issued in the reference's order.  A sum that the reference evaluates as
 *     one chain is owned by ONE thread here; parallelism comes only from independent chains (lags, units,
 *     unit-count trials, regulariser passes, samples, channels, frames).
 *   - libm values the reference takes from glibc (Welch divisor pow(n-1,-2), the SIN window) are computed
 *     on the host and passed in as tables.
 *   - int32 filters use 32-bit wrap-around arithmetic (uint32 multiply/add, arithmetic shift).
 *
 * Citations are file:line under /root/reference.
 */
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "linne_amd.h"
#include "lnn_common.h"

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

struct Plan {
    uint32_t C, S, bits, L, R, ms, F, J;
    uint32_t P[LNN_MAXL], coef_off[LNN_MAXL];
    double regs[LNN_MAXR];
    double scale;                       /* 2^-(bits-1), exact */
    const int32_t *pcm; int32_t *resid; int32_t *prm; double *stats;
    const uint32_t *cls_of_frame; const DevClass *cls; const double *sintab; const double *wtab;
    int32_t *xint, *xtmp;               /* [F*C][S]                    */
    double *sig;                        /* [J][2][S]                   */
    double *acorr;                      /* [J][MAXT][ACW]              */
    double *tcoef;                      /* [J][MAXT][MAXP]  filter order (reversed LPC order) */
    double *ptail; uint8_t *ptail_set;  /* [J][MAXT][MAXU]             */
    double *tloss;                      /* [J][MAXT] exact mean |residual| (ordered chain)            */
    double *tsum;                       /* [J][MAXT][npart] per-wave partial sums of |residual| (certified search) */
    uint32_t npart;                     /* partial sums per (job, trial): tiles x waves per block      */
    uint8_t *uncertain;                 /* [J] the order-free sums could not certify the argmin       */
    uint32_t *ucount;                   /* running count of such (job, layer) pairs of the call       */
    double *lparams;                    /* [J][MAXL][MAXP]             */
    uint32_t *lunits;                   /* [J][MAXL]                   */
    double *jloss, *jtail;              /* [J]                         */
};

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

/* ------------------------------------------------------------------------------------------------
 * K1: per frame -- copy, MS, two pre-emphasis stages, block-type statistics
 * ---------------------------------------------------------------------------------------------- */
#define PREP_THREADS 256
#define PREP_CHUNK   1024           /* products staged per round of the ordered pre-emphasis chains */
/* block-wide integer reductions: shuffle tree inside each wavefront, then one LDS hop (exact: integer add / max) */
__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t *sh)
{
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                        /* sh may still be read from a previous reduction */
    if ((t & 63u) == 0) sh[t >> 6] = v;
    __syncthreads();
    int64_t r = 0;
#pragma unroll
    for (uint32_t w = 0; w < PREP_THREADS / 64; w++) r += sh[w];
    return r;
}
__device__ __forceinline__ int64_t block_max_i64(int64_t v, int64_t *sh)
{
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int64_t x = __shfl_xor(v, o); v = x > v ? x : v; }
    __syncthreads();
    if ((t & 63u) == 0) sh[t >> 6] = v;
    __syncthreads();
    int64_t r = sh[0];
#pragma unroll
    for (uint32_t w = 1; w < PREP_THREADS / 64; w++) r = sh[w] > r ? sh[w] : r;
    return r;
}

__global__ __launch_bounds__(PREP_THREADS) void k_prep(Plan p)
{
    __shared__ int64_t sh[PREP_THREADS / 64];
    __shared__ int32_t sh_coef;
    __shared__ double sh_prod[2][PREP_CHUNK];
    const uint32_t f = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;     /* one block per (frame, channel) */
    const DevClass &c = p.cls[p.cls_of_frame[f]];
    const uint32_t n = c.n, S = p.S, C = p.C;
    const int32_t *in = p.pcm + (size_t)f * C * S;
    int32_t *src = p.xint + ((size_t)f * C + ch) * S, *dst = p.xtmp + ((size_t)f * C + ch) * S;
    int32_t *rec = p.prm + ((size_t)f * C + ch) * LINNE_AMD_PARAM_WORDS;

    /* copy with zero padding (linne_encoder.c:613-621); LR -> MS on channels 0/1 (linne_utility.c:120-132): each of
     * the two blocks derives its own channel from L and R */
    for (uint32_t s = tid; s < S; s += PREP_THREADS) {
        int32_t v = 0;
        if (s < n) {
            v = in[(size_t)ch * S + s];
            if (p.ms && ch < 2) {
                const uint32_t l = (uint32_t)in[s], r = (uint32_t)in[(size_t)S + s];
                const int32_t side = (int32_t)(r - l);
                v = (ch == 1) ? side : (int32_t)(l + (uint32_t)(side >> 1));
            }
        }
        src[s] = v;
    }
    __syncthreads();

    /* two pre-emphasis stages (linne_encoder.c:634-641) */
    for (uint32_t stage = 0; stage < 2; stage++) {
        /* coefficient: linne_utility.c:158-193.  corr0 = sum x[s]^2, corr1 = sum x[s]x[s+1], s < n-1, are double chains
         * in the reference; when max|x|^2 * n < 2^53 every partial sum is an exactly representable integer, so any
         * summation order gives the reference's bits (integer path); otherwise one thread runs the chains in order. */
        int64_t mx = 0; uint64_t s0 = 0, s1 = 0, sq = 0;
        for (uint32_t s = tid; s < n; s += PREP_THREADS) {
            const int64_t a = src[s]; const int64_t av = a < 0 ? -a : a;
            mx = av > mx ? av : mx;
            sq += (uint64_t)(a * a);
            if (s + 1 < n) { const int64_t b = src[s + 1]; s0 += (uint64_t)(a * a); s1 += (uint64_t)(a * b); }
        }
        mx = block_max_i64(mx, sh);
        /* every partial sum of either chain is bounded by sum x^2 (|ab| <= (a^2 + b^2) / 2): below 2^53 they are all
         * exactly representable integers.  (mx^2 * n < 2^62 first: then the 64-bit sums themselves cannot wrap.) */
        bool exact = ((double)mx * (double)mx * (double)n) < 4.0e18;
        if (exact) exact = (uint64_t)block_sum_i64((int64_t)sq, sh) < (1ull << 53);
        double c0 = 0.0, c1 = 0.0;
        if (exact) {
            c0 = (double)block_sum_i64((int64_t)s0, sh);
            c1 = (double)block_sum_i64((int64_t)s1, sh);
        } else {
            /* ordered chains: the products (exact: |x| < 2^31 squares may round, as in the reference's double multiply) are
             * formed by all threads, chunk by chunk, into LDS; lane 0 adds the squares and lane 1 the cross products in
             * sample order */
            double acc = 0.0;
            for (uint32_t base = 0; base + 1 < n; base += PREP_CHUNK) {
                const uint32_t cnt = (n - 1 - base < PREP_CHUNK) ? (n - 1 - base) : PREP_CHUNK;
                __syncthreads();
                for (uint32_t i = tid; i < cnt; i += PREP_THREADS) {
                    const double curr = (double)src[base + i], succ = (double)src[base + i + 1];
                    sh_prod[0][i] = curr * curr; sh_prod[1][i] = curr * succ;
                }
                __syncthreads();
                if (tid < 2) {
                    const double *q = sh_prod[tid];
                    uint32_t i = 0;
                    for (; i + 8 <= cnt; i += 8) {
                        const double q0 = q[i], q1 = q[i + 1], q2 = q[i + 2], q3 = q[i + 3], q4 = q[i + 4], q5 = q[i + 5], q6 = q[i + 6], q7 = q[i + 7];
                        acc += q0; acc += q1; acc += q2; acc += q3; acc += q4; acc += q5; acc += q6; acc += q7;
                    }
                    for (; i < cnt; i++) acc += q[i];
                }
            }
            __syncthreads();
            if (tid == 1) sh_prod[1][0] = acc;
            __syncthreads();
            if (tid == 0) { c0 = acc; c1 = sh_prod[1][0]; }
        }
        if (tid == 0) {
            int32_t coef;
            c1 /= c0;
            if ((c0 < 1e-6) || (c1 < 0.0)) coef = 0;
            else { coef = (int32_t)round_away(c1 * 32.0); if (coef >= 16) coef = 15; }
            sh_coef = coef;
            rec[LINNE_AMD_PRM_PREV + stage] = src[0];
            rec[LINNE_AMD_PRM_PCOEF + stage] = coef;
        }
        __syncthreads();
        const int32_t coef = sh_coef;
        /* linne_utility.c:196-212 with prev := first sample */
        for (uint32_t s = tid; s < S; s += PREP_THREADS) {
            int32_t v = src[s];
            if (s < n) { const int32_t prev = src[s ? s - 1 : 0]; v = (int32_t)((uint32_t)v - (uint32_t)mulshr5(prev, coef)); }
            dst[s] = v;
        }
        __syncthreads();
        int32_t *t = src; src = dst; dst = t;
    }
    /* two stages: xint -> xtmp -> xint, the pre-emphasised channel is back in xint */
}

/* block-type statistics (linne_encoder.c:494-503 -> lpc.c:810-848): SIN-window autocorrelation of the RAW channel at
 * order P0 = layer-0 size, one chain per (channel, lag), then Levinson-Durbin.  Independent of the analysis, so it runs
 * on a side stream concurrently with it. */
__global__ __launch_bounds__(64) void k_stats(Plan p)
{
    __shared__ double sh_r[LNN_MAXCH][8];
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const DevClass &c = p.cls[p.cls_of_frame[f]];
    const uint32_t n = c.n, S = p.S, C = p.C;
    const int32_t *in = p.pcm + (size_t)f * C * S;
    const uint32_t P0 = p.P[0];
    const double *sinw = p.sintab + c.sin_off;
    if (tid < C * (P0 + 1)) {
        const uint32_t ch = tid / (P0 + 1), lag = tid % (P0 + 1);
        const int32_t *x = in + (size_t)ch * S;
        double r = 0.0;
        if (lag < n) {
            const uint32_t cnt = n - lag;
            uint32_t i = 0;
            for (; i + 8 <= cnt; i += 8) {              /* loads and products of 8 steps are independent; the adds stay in order */
                double pr[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double a = ((double)x[i + k] * p.scale) * sinw[i + k];
                    const double b = ((double)x[i + k + lag] * p.scale) * sinw[i + k + lag];
                    pr[k] = a * b;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) r += pr[k];
            }
            for (; i < cnt; i++) {
                const double a = ((double)x[i] * p.scale) * sinw[i];
                const double b = ((double)x[i + lag] * p.scale) * sinw[i + lag];
                r += a * b;
            }
        }
        sh_r[ch][lag] = r;
    }
    __syncthreads();
    if (tid < C) {
        double a[8], pc[8], rl[8];
        double *st = p.stats + ((size_t)f * C + tid) * LINNE_AMD_STAT_WORDS;
        for (uint32_t i = 0; i <= P0; i++) rl[i] = sh_r[tid][i];
        const double r0 = rl[0] * (1.0 + 0.0);
        const int zero = (n < P0) || (fabs(r0) < (double)FLT_EPSILON);
        for (uint32_t i = 0; i < 8; i++) pc[i] = 0.0;
        if (!zero) levinson(rl, r0, P0, a, pc);
        st[LINNE_AMD_ST_R0] = rl[0];
        st[LINNE_AMD_ST_K1 + 0] = pc[1]; st[LINNE_AMD_ST_K1 + 1] = pc[2]; st[LINNE_AMD_ST_K1 + 2] = pc[3];
        st[LINNE_AMD_ST_ZERO] = zero ? 1.0 : 0.0;
    }
}

/* ------------------------------------------------------------------------------------------------
 * analysis, one layer at a time over every job = (channel-frame, regulariser pass)
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ const DevClass &job_class(const Plan &p, uint32_t job) { return p.cls[p.cls_of_frame[(job / p.R) / p.C]]; }

/* ------------------------------------------------------------------------------------------------
 * K_A (v2): Welch window + autocorrelation of every unit-count trial of one layer, fused.
 *
 * Work decomposition (DESIGN.md "autocorrelation kernel"): one wavefront per JPW jobs.  A LANE owns K = 5
 * consecutive lags of one trial of one job and walks ALL units of that trial in order, so every lane runs the
 * same number of steps (~ na + P) and each lag's sum stays one chain in increasing sample order.  The windowed
 * signal of a trial is produced on the fly into a small LDS ring ("padded stream": each unit is followed by
 * z = max(p,4) zeros, so a lane's 5-lag register window can slide across unit ends without masking and the
 * accumulators can be flushed at a group boundary inside the zero zone).  Per 5 steps a lane issues 25 unfused
 * mul+add pairs against 10 LDS reads.
 * ---------------------------------------------------------------------------------------------- */
template <int P> struct AcCfg {
    static constexpr int K = 5;
    static constexpr int NT = (P >= 128) ? 8 : ((P >= 64) ? 7 : (P >= 32) ? 6 : (P >= 16) ? 5 : (P >= 8) ? 4 : (P >= 4) ? 3 : 2);
    static constexpr int T = (P >= 32) ? 60 : 20;                       /* tile: padded positions per LDS refill */
    static constexpr int lanes(int t) { return ((P >> t) + 1 + K - 1) / K; }
    static constexpr int halo(int t) { return K * lanes(t) + K; }       /* furthest window read past a group start, +1 */
    static constexpr int rb(int t) { return ((T + halo(t) + 3 + 4 * K - 1) / (4 * K)) * (4 * K); }   /* ring length, multiple of 2K and of 4 */
    static constexpr int lpj() { int s = 0; for (int t = 0; t < NT; t++) s += lanes(t); return s; }
    static constexpr int ringsum() { int s = 0; for (int t = 0; t < NT; t++) s += rb(t); return s; }
    static constexpr int maxpad() { int m = 0; for (int t = 0; t < NT; t++) { const int p = P >> t, z = p > 4 ? p : 4, v = (1 << t) * z; if (v > m) m = v; } return m; }
    static constexpr int LPJ = lpj();
    static constexpr int JPW = 64 / LPJ;
    static constexpr int NS = JPW * NT;
    static constexpr int GL = (64 / NS) >= 1 ? (64 / NS) : 1;
    static constexpr int RINGSUM = ringsum();
    static constexpr int MAXPAD = maxpad();
};

template <int P, bool L0>
__global__ __launch_bounds__(64) void k_autocorr2(Plan p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    using Cfg = AcCfg<P>;
    constexpr int K = Cfg::K, NT = Cfg::NT, T = Cfg::T;
    __shared__ __attribute__((aligned(16))) double ring[Cfg::JPW * Cfg::RINGSUM];
    const uint32_t lane = threadIdx.x;

    /* ---- accumulate role: (job, trial, lag group) ---- */
    bool active = false;
    uint32_t a_job = 0, a_t = 0, a_lag0 = 0, a_u = 1, a_p = P, a_upl = 1;
    int32_t a_base = 0, a_rb = 5;          /* ring base (doubles) and ring length of my stream */
    {
        const uint32_t jl = lane / Cfg::LPJ;
        uint32_t rem = lane % Cfg::LPJ;
        if (jl < (uint32_t)Cfg::JPW) {
            uint32_t t = 0; int32_t off = 0;
            for (; t < (uint32_t)NT; t++) { if (rem < (uint32_t)Cfg::lanes(t)) break; rem -= Cfg::lanes(t); off += Cfg::rb(t); }
            a_job = blockIdx.x * Cfg::JPW + jl;
            if (a_job < p.J) {
                const DevClass &c = job_class(p, a_job);
                if (t < c.ntrials[layer]) {
                    active = true;
                    a_t = t; a_lag0 = rem * K; a_u = 1u << t; a_p = P >> t;
                    const uint32_t n = c.na / a_u;
                    a_upl = n + (a_p > 4 ? a_p : 4);
                    a_base = (int32_t)(jl * Cfg::RINGSUM) + off; a_rb = Cfg::rb(t);
                }
            }
        }
    }
    uint32_t a_n = 1;
    if (active) a_n = job_class(p, a_job).na / a_u;

    /* ---- generate role: (stream, sub-lane) ---- */
    bool gen = false;
    uint32_t g_n = 1, g_u = 1, g_upl = 5, g_halo = 0;
    int32_t g_base = 0, g_rb = 5, g_pos = 0;       /* ring slot of g_q */
    uint32_t g_q = 0, g_unit = 0, g_loc = 0, g_ubase = 0;
    double g_stale = 0.0;
    const double *g_wt = p.wtab;                    /* Welch weights of my trial, one padded unit */
    const double *g_xd = p.sig; const int32_t *g_xi = p.xint;      /* always dereferenceable */
    {
        const uint32_t gs = lane / Cfg::GL, sub = lane % Cfg::GL;
        if (gs < (uint32_t)Cfg::NS) {
            const uint32_t jl = gs / NT, t = gs % NT;
            const uint32_t job = blockIdx.x * Cfg::JPW + jl;
            if (job < p.J) {
                const DevClass &c = job_class(p, job);
                if (t < c.ntrials[layer]) {
                    gen = true;
                    g_u = 1u << t; g_n = c.na / g_u;
                    const uint32_t pp = P >> t;
                    g_upl = g_n + (pp > 4 ? pp : 4);
                    g_halo = Cfg::halo(t);
                    int32_t off = 0;
                    for (uint32_t i = 0; i < t; i++) off += Cfg::rb(i);
                    g_base = (int32_t)(jl * Cfg::RINGSUM) + off; g_rb = Cfg::rb(t);
                    g_wt = p.wtab + c.wt_off[layer][t];
                    if (L0) g_xi = p.xint + (size_t)(job / p.R) * p.S; else g_xd = p.sig + ((size_t)job * 2 + cur) * p.S;
                    g_q = sub; g_loc = sub; g_pos = (int32_t)sub;
                    while (g_loc >= g_upl) { g_loc -= g_upl; g_unit++; g_ubase += g_n; }
                    if (g_n & 1u) {     /* Q1: stale middle sample = previous trial's last unit at local index m */
                        const uint32_t m = g_n >> 1, n2 = 2 * g_n, si = (g_u / 2 - 1) * n2 + m;
                        const double xv = L0 ? ((double)g_xi[si] * p.scale) : g_xd[si];
                        const double wgt = c.trial_div[layer][t - 1] * (double)m * (double)(n2 - 1 - m);
                        g_stale = xv * wgt;
                    }
                }
            }
        }
    }

    /* Fast generator: when every stream of the wave has unit and padded-unit lengths that are multiples of 4 (always for
     * frame lengths that are multiples of 4 * 128), a generator lane produces 4 consecutive stream positions at a time --
     * they never straddle a unit end or the ring end -- with 16-byte loads and stores; the bookkeeping per element drops
     * to a quarter.  Same products in the same places; the choice is per wave and holds for the whole kernel. */
    const bool fastgen = __all(!gen || (((g_n | g_upl) & 3u) == 0));
    if (fastgen && gen) {
        const uint32_t sub = lane % Cfg::GL;
        g_halo = (g_halo + 3u) & ~3u;
        g_q = 4 * sub; g_loc = 4 * sub; g_pos = (int32_t)(4 * sub); g_unit = 0; g_ubase = 0;
        while (g_loc >= g_upl) { g_loc -= g_upl; g_unit++; g_ubase += g_n; }
    }
    constexpr uint32_t GSTEP4 = 4 * Cfg::GL;
    constexpr int E4 = (T + 4 * Cfg::GL - 1) / (4 * Cfg::GL);
    auto gen_advance4 = [&]() {
        g_q += GSTEP4; g_loc += GSTEP4;
        while (g_loc >= g_upl) { g_loc -= g_upl; g_unit++; g_ubase += g_n; }
    };
    struct Q4 { double v[4]; };
    auto gen_fetch4 = [&](uint32_t si) -> Q4 {          /* si is a multiple of 4: 16-byte aligned pieces */
        Q4 q;
        if (L0) { const int4 iv = *(const int4 *)(g_xi + si); q.v[0] = (double)iv.x * p.scale; q.v[1] = (double)iv.y * p.scale; q.v[2] = (double)iv.z * p.scale; q.v[3] = (double)iv.w * p.scale; }
        else { const lnn_d2 a = *(const lnn_d2 *)(g_xd + si), b = *(const lnn_d2 *)(g_xd + si + 2); q.v[0] = a.x; q.v[1] = a.y; q.v[2] = b.x; q.v[3] = b.y; }
        return q;
    };
    auto gen_weight4 = [&](uint32_t loc) -> Q4 {
        Q4 q; const lnn_d2 a = *(const lnn_d2 *)(g_wt + loc), b = *(const lnn_d2 *)(g_wt + loc + 2);
        q.v[0] = a.x; q.v[1] = a.y; q.v[2] = b.x; q.v[3] = b.y; return q;
    };
    auto gen_store4 = [&](const Q4 &x, const Q4 &wq, bool in_unit) {
        lnn_d2 a, b;
        a.x = in_unit ? x.v[0] * wq.v[0] : 0.0; a.y = in_unit ? x.v[1] * wq.v[1] : 0.0;
        b.x = in_unit ? x.v[2] * wq.v[2] : 0.0; b.y = in_unit ? x.v[3] * wq.v[3] : 0.0;
        *(lnn_d2 *)(ring + g_base + g_pos) = a; *(lnn_d2 *)(ring + g_base + g_pos + 2) = b;
        g_pos += (int32_t)GSTEP4; if (g_pos >= g_rb) g_pos -= g_rb;
    };

    /* per-lane accumulate state */
    double r[K], w[K];
#pragma unroll
    for (int j = 0; j < K; j++) { r[j] = 0.0; w[j] = 0.0; }
    uint32_t a_unit = 0, flush_pos = a_n;           /* first padded position after unit 0's samples */
    int32_t pa = 0, pw = (int32_t)a_lag0;           /* ring slots of q0 and of q0 + lag0 */
    double *out = nullptr;
    if (active) out = p.acorr + ((size_t)a_job * LNN_MAXT + a_t) * LNN_ACW;
    /* volatile LDS pointer: keeps the 8-byte reads unmerged (ds_read2_b64 runs at half the rate of two ds_read_b64) */
    typedef const volatile __attribute__((address_space(3))) double *lds_ro_ptr;
    lds_ro_ptr myring = (lds_ro_ptr)(ring + a_base);
    double *gring = ring + g_base;

    /* generator step: classify padded position g_q -> sample index (or zero / stale), then advance */
    constexpr int E = (T + Cfg::GL - 1) / Cfg::GL;  /* elements one generator lane adds per tile (at most) */
    auto gen_advance = [&]() {
        g_q += Cfg::GL; g_loc += Cfg::GL;
        while (g_loc >= g_upl) { g_loc -= g_upl; g_unit++; g_ubase += g_n; }
    };
    /* element of the padded stream: x[unit*n + loc] * w(loc); w is 0 in the zero zone; Q1 replaces an odd unit's middle */
    auto gen_value = [&](double xv, double wv, uint32_t loc) -> double {
        const double v = xv * wv;
        return ((g_n & 1u) && loc == (g_n >> 1)) ? g_stale : v;
    };
    auto gen_fetch = [&](uint32_t si) -> double { return L0 ? ((double)g_xi[si] * p.scale) : g_xd[si]; };

    /* initial fill: padded positions [0, T + halo) */
    if (gen && fastgen) {
        const uint32_t lim = T + g_halo;
        while (g_q < lim) {
            const bool in_unit = (g_unit < g_u) && (g_loc < g_n);
            const Q4 x = gen_fetch4(in_unit ? (g_ubase + g_loc) : 0u), wq = gen_weight4(in_unit ? g_loc : 0u);
            gen_store4(x, wq, in_unit);
            gen_advance4();
        }
    } else if (gen) {
        const uint32_t lim = T + g_halo;
        while (g_q < lim) {
            double v = 0.0;
            if (g_unit < g_u && g_loc < g_n) v = gen_value(gen_fetch(g_ubase + g_loc), g_wt[g_loc], g_loc);
            gring[g_pos] = v;
            g_pos += Cfg::GL; if (g_pos >= g_rb) g_pos -= g_rb;
            gen_advance();
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int j = 0; j < K; j++) w[j] = myring[pw + j];
        pw += K; if (pw >= a_rb) pw -= a_rb;
    }

    const uint32_t q_end = na_max + Cfg::MAXPAD + K;       /* uniform bound: past every lane's last flush */
    for (uint32_t tile0 = 0; tile0 < q_end; tile0 += T) {
        /* prefetch the samples of the NEXT refill (positions [tile0 + T + halo, tile0 + 2T + halo)) into registers;
         * their latency hides behind this tile's accumulation */
        double fx[E], fw[E]; uint32_t floc[E];
        Q4 qx[E4], qw[E4]; uint32_t qin[E4];                /* fast generator: 0 = not mine, 1 = zero zone, 2 = samples */
        if (fastgen) {
            const uint32_t lim = tile0 + 2 * T + g_halo;
#pragma unroll
            for (int e = 0; e < E4; e++) {
                const bool in_range = gen && (g_q < lim);
                const bool in_unit = in_range && (g_unit < g_u) && (g_loc < g_n);
                qin[e] = in_unit ? 2u : (in_range ? 1u : 0u);
                qx[e] = gen_fetch4(in_unit ? (g_ubase + g_loc) : 0u);
                qw[e] = gen_weight4(in_unit ? g_loc : 0u);
                if (in_range) gen_advance4();
            }
        } else {
            const uint32_t lim = tile0 + 2 * T + g_halo;
#pragma unroll
            for (int e = 0; e < E; e++) {           /* straight-line: E independent loads in flight */
                const bool in_range = gen && (g_q < lim);
                const bool in_unit = in_range && (g_unit < g_u) && (g_loc < g_n);
                const uint32_t si = in_unit ? (g_ubase + g_loc) : 0u;
                floc[e] = in_unit ? g_loc : (in_range ? 0xFFFFFFFEu : 0xFFFFFFFFu);
                fx[e] = gen_fetch(si);
                fw[e] = g_wt[in_unit ? g_loc : 0u];
                if (in_range) gen_advance();
            }
        }
        if (active) {
            /* two 5-step groups per trip: the window registers swap roles (w -> nw -> w), so nothing is moved; the ring
             * length is a multiple of 10, so q0's slot wraps at most once per trip */
            auto flush = [&](uint32_t q) {
                if (q >= flush_pos) {                   /* inside the zero zone after a unit: store and restart */
                    double *o = out + (size_t)a_unit * (a_p + 1) + a_lag0;
#pragma unroll
                    for (int j = 0; j < K; j++) { if (a_lag0 + j <= a_p) o[j] = r[j]; r[j] = 0.0; }
                    a_unit++;
                    flush_pos = (a_unit < a_u) ? (flush_pos + a_upl) : 0xFFFFFFFFu;
                }
            };
#pragma unroll 1
            for (uint32_t q0 = tile0; q0 < tile0 + T; q0 += 2 * K) {
                double a[K], nw[K];
                flush(q0);
#pragma unroll
                for (int j = 0; j < K; j++) { a[j] = myring[pa + j]; nw[j] = myring[pw + j]; }
                int32_t pw2 = pw + K; if (pw2 >= a_rb) pw2 -= a_rb;
#pragma unroll
                for (int t = 0; t < K; t++) {
#pragma unroll
                    for (int j = 0; j < K; j++) r[j] += a[t] * ((t + j < K) ? w[t + j] : nw[t + j - K]);
                }
                flush(q0 + K);
#pragma unroll
                for (int j = 0; j < K; j++) { a[j] = myring[pa + K + j]; w[j] = myring[pw2 + j]; }
                pa += 2 * K; if (pa >= a_rb) pa -= a_rb;
                pw = pw2 + K; if (pw >= a_rb) pw -= a_rb;
#pragma unroll
                for (int t = 0; t < K; t++) {
#pragma unroll
                    for (int j = 0; j < K; j++) r[j] += a[t] * ((t + j < K) ? nw[t + j] : w[t + j - K]);
                }
            }
        }
        __syncthreads();
        if (fastgen) {
#pragma unroll
            for (int e = 0; e < E4; e++) if (qin[e]) gen_store4(qx[e], qw[e], qin[e] == 2u);
        } else if (gen) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (floc[e] != 0xFFFFFFFFu) {
                    const double gv = gen_value(fx[e], fw[e], floc[e]);
                    gring[g_pos] = (floc[e] == 0xFFFFFFFEu) ? 0.0 : gv;
                    g_pos += Cfg::GL; if (g_pos >= g_rb) g_pos -= g_rb;
                }
            }
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------------------------------
 * K_A for the short layers (P <= 16): one lane per (job, trial) owns ALL p+1 lags of the trial; a wavefront holds 64
 * jobs of the SAME trial (grid.y = trial), so every lane issues the same number of multiply-adds.  A lane produces its
 * own padded windowed stream on the fly -- one new element per step, loads prefetched one 4-step group ahead -- into a
 * register window w[0..K+3]; step q adds w[q]*w[q+j] to lag j.  No LDS, no barrier.  Same chains, same order as
 * k_autocorr2.
 * ---------------------------------------------------------------------------------------------- */
template <int K, bool L0>
__device__ __forceinline__ void autocorr_lane(const Plan &p, uint32_t layer, uint32_t cur, uint32_t q_end, uint32_t job, uint32_t t, bool active)
{
    constexpr int U = 4;
    constexpr uint32_t np = K - 1;
    const DevClass &c = job_class(p, job);
    const uint32_t u = 1u << t;
    const uint32_t n = active ? (c.na / u) : 1u;
    const uint32_t upl = n + (np > 4 ? np : 4);
    const double *wt = p.wtab + (active ? c.wt_off[layer][t] : 0u);
    const int32_t *xi = p.xint + (size_t)(job / p.R) * p.S;
    const double *xd = p.sig + ((size_t)job * 2 + cur) * p.S;
    double stale = 0.0;
    if (active && (n & 1u)) {                       /* Q1, as in k_autocorr2 */
        const uint32_t m = n >> 1, n2 = 2 * n, si = (u / 2 - 1) * n2 + m;
        const double xv = L0 ? ((double)xi[si] * p.scale) : xd[si];
        stale = xv * (c.trial_div[layer][t - 1] * (double)m * (double)(n2 - 1 - m));
    }
    const bool odd = (n & 1u) != 0;
    const uint32_t mid = n >> 1;
    uint32_t g_loc = 0, g_ubase = 0, g_left = active ? u : 0u;     /* units still to come (incl. the current one) */
    auto gen_issue = [&](double &rx, double &rw, uint32_t &rloc) {
        const bool in_unit = (g_loc < n) && (g_left != 0);
        rx = L0 ? ((double)xi[in_unit ? (g_ubase + g_loc) : 0u] * p.scale) : xd[in_unit ? (g_ubase + g_loc) : 0u];
        rw = wt[g_loc];                              /* zero inside the zero zone (table covers the padded unit) */
        rloc = in_unit ? g_loc : 0xFFFFFFFFu;
        const bool wrap = (g_loc + 1 >= upl);
        g_loc = wrap ? 0u : g_loc + 1;
        g_ubase = (wrap && g_left > 1) ? g_ubase + n : g_ubase;
        g_left = (wrap && g_left) ? g_left - 1 : g_left;
    };
    auto gen_finish = [&](double rx, double rw, uint32_t rloc) -> double {
        const double v = rx * rw;
        const double vv = (odd && rloc == mid) ? stale : v;
        return (rloc == 0xFFFFFFFFu) ? 0.0 : vv;
    };
    double r[K], w[K + U], fx[U], fw[U]; uint32_t fl[U];
#pragma unroll
    for (int j = 0; j < K; j++) {
        r[j] = 0.0;
        double rx, rw; uint32_t rl;
        gen_issue(rx, rw, rl);
        w[j] = gen_finish(rx, rw, rl);
    }
#pragma unroll
    for (int j = 0; j < U; j++) gen_issue(fx[j], fw[j], fl[j]);
    uint32_t a_unit = 0, flush_pos = n;
    double *out = p.acorr + ((size_t)job * LNN_MAXT + t) * LNN_ACW;
#pragma unroll 1
    for (uint32_t q0 = 0; q0 < q_end; q0 += U) {
        if (active && q0 >= flush_pos) {           /* in the zero zone after a unit: store its lags, restart */
            double *o = out + (size_t)a_unit * K;
#pragma unroll
            for (int j = 0; j < K; j++) { o[j] = r[j]; r[j] = 0.0; }
            a_unit++;
            flush_pos = (a_unit < u) ? (flush_pos + upl) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < U; j++) w[K + j] = gen_finish(fx[j], fw[j], fl[j]);
#pragma unroll
        for (int j = 0; j < U; j++) gen_issue(fx[j], fw[j], fl[j]);      /* next group's loads fly during the MACs */
#pragma unroll
        for (int tt = 0; tt < U; tt++) {
#pragma unroll
            for (int j = 0; j < K; j++) r[j] += w[tt] * w[tt + j];
        }
#pragma unroll
        for (int j = 0; j < K; j++) w[j] = w[j + U];
    }
}

/* Fast form for a block whose 64 rows share one length class with every unit length a multiple of 4 (any frame length
 * that is a multiple of 64: the CLI's 10240-sample blocks and their usual tails): wave t of the block owns trial t of the
 * same 64 rows.  The samples are read from HBM ONCE for all trials, coalesced (one load instruction covers 16 consecutive
 * samples of 4 rows), and handed to the lanes through a transposed LDS tile; the stream bookkeeping (unit position, pad
 * zones, flushes) is wave-uniform.  Same products, same chains, same order as autocorr_lane. */
#define ACS_T 32
template <int K, int J0, int JN, bool L0, int NT>
__device__ __forceinline__ void autocorr_shared(const Plan &p, uint32_t layer, uint32_t cur, uint32_t row0, uint32_t nrows, uint32_t rstride,
        uint32_t na, uint32_t wt_off, uint32_t t, uint32_t wave, uint32_t lane, double (*tile)[ACS_T][65])
{
    constexpr uint32_t np = K - 1, pad = (np > 4 ? np : 4);
    constexpr int L = ((int)np + 3) / 4 * 4;               /* window lead: elements held ahead of the current step */
    constexpr int NSLOT = (ACS_T + NT - 1) / NT;           /* load slots (64/ACS_T rows x ACS_T samples) this wave may own; ACS_T slots per tile */
    const uint32_t u = 1u << t, n = na / u, upl = n + pad, ntiles = na / ACS_T;
    const double *wt = p.wtab + wt_off;
    uint32_t myrow = row0 + lane; if (myrow >= nrows) myrow = nrows - 1;
    const bool store = (row0 + lane) < nrows;
    double *out = p.acorr + ((size_t)myrow * rstride * LNN_MAXT + t) * LNN_ACW;
    /* loads run two tiles ahead of the tile being consumed, in two register sets picked by the tile's parity */
    double preA[NSLOT], preB[NSLOT], wA = 0.0, wB = 0.0, wcur = 0.0;   /* w*: Welch weights of a tile, lane j holds sample j's */
    const uint32_t ls = lane & (ACS_T - 1u), lr = lane / ACS_T;
    constexpr uint32_t RPS = 64 / ACS_T;                    /* rows per load slot; a tile has 64 / RPS = ACS_T slots */
    auto issue = [&](uint32_t tile_idx, double *pre, double &wv) {
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            const uint32_t k = wave + (uint32_t)i * NT;
            if (k < 64 / RPS) {
                uint32_t r = row0 + RPS * k + lr; if (r >= nrows) r = nrows - 1;
                const uint32_t sidx = tile_idx * ACS_T + ls;
                if (L0) pre[i] = (double)p.xint[(size_t)r * p.S + sidx] * p.scale;   /* rows are channel-frames */
                else pre[i] = p.sig[((size_t)r * 2 + cur) * p.S + sidx];
            }
        }
        wv = wt[(tile_idx * ACS_T + ls) % n];               /* the weight depends on the place inside the unit only */
    };
    auto commit = [&](uint32_t buf, const double *pre, double wv) {
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            const uint32_t k = wave + (uint32_t)i * NT;
            if (k < 64 / RPS) tile[buf][ls][RPS * k + lr] = pre[i];
        }
        wcur = wv;
    };
    auto lane_bcast = [&](double v, uint32_t src_lane) -> double {     /* wave-uniform src_lane: two v_readlane */
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)src_lane);
        return __hiloint2double(hi, lo);
    };
    issue(0, preA, wA); commit(0, preA, wA);
    __syncthreads();
    if (ntiles > 1) issue(1, preB, wB);
    if (ntiles > 2) issue(2, preA, wA);
    uint32_t g_loc = 0, g_tile = 0, g_off = 0;              /* generator: place in the padded unit, tile, offset in it */
    struct D4 { double v0, v1, v2, v3; };
    auto next4 = [&]() -> D4 {
        D4 d; d.v0 = 0.0; d.v1 = 0.0; d.v2 = 0.0; d.v3 = 0.0;      /* zero zone after a unit, or past the last unit */
        if (g_loc < n && g_tile < ntiles) {                 /* four samples of the current unit */
            const double *src = &tile[g_tile & 1u][g_off][lane];
            d.v0 = src[0] * lane_bcast(wcur, g_off); d.v1 = src[65] * lane_bcast(wcur, g_off + 1);
            d.v2 = src[130] * lane_bcast(wcur, g_off + 2); d.v3 = src[195] * lane_bcast(wcur, g_off + 3);
            g_off += 4;
            if (g_off == ACS_T) {                           /* tile used up: publish the prefetched one */
                g_off = 0; g_tile++;
                if (g_tile < ntiles) {                      /* odd tiles travel in set B, even ones in set A */
                    if (g_tile & 1u) { commit(1, preB, wB); __syncthreads(); if (g_tile + 2 < ntiles) issue(g_tile + 2, preB, wB); }
                    else             { commit(0, preA, wA); __syncthreads(); if (g_tile + 2 < ntiles) issue(g_tile + 2, preA, wA); }
                }
            }
        }
        g_loc += 4; if (g_loc >= upl) g_loc = 0;
        return d;
    };
    /* This wave accumulates lags J0 .. J0+JN-1 of the trial.  The stream window is a register ring of W = L + 4 elements
     * (w[i % W] = element i): the loop body is unrolled over one turn of the ring, so the window never moves. */
    constexpr int W = L + 4, NG = W / 4;
    double r[JN], w[W];
#pragma unroll
    for (int j = 0; j < JN; j++) r[j] = 0.0;
#pragma unroll
    for (int j = 0; j < L / 4; j++) { const D4 d = next4(); w[4 * j] = d.v0; w[4 * j + 1] = d.v1; w[4 * j + 2] = d.v2; w[4 * j + 3] = d.v3; }
    uint32_t a_unit = 0, flush_pos = n, q0 = 0;
    bool done = false;
#pragma unroll 1
    while (!done) {
#pragma unroll
        for (int g = 0; g < NG; g++) {                     /* steps q0 .. q0+3 with element q0 + i in w[(4g + i) % W] */
            if (!done) {
                if (q0 >= flush_pos) {                     /* in the zero zone after a unit: store its lags, restart */
                    if (store) {
                        double *o = out + (size_t)a_unit * K + J0;
#pragma unroll
                        for (int j = 0; j < JN; j++) o[j] = r[j];
                    }
#pragma unroll
                    for (int j = 0; j < JN; j++) r[j] = 0.0;
                    a_unit++;
                    flush_pos += upl;
                    done = (a_unit == u);
                }
                if (!done) {
                    const D4 d = next4();
                    w[(4 * g + L) % W] = d.v0; w[(4 * g + L + 1) % W] = d.v1; w[(4 * g + L + 2) % W] = d.v2; w[(4 * g + L + 3) % W] = d.v3;
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) {
#pragma unroll
                        for (int j = 0; j < JN; j++) r[j] += w[(4 * g + tt) % W] * w[(4 * g + tt + J0 + j) % W];
                    }
                    q0 += 4;
                }
            }
        }
    }
}

/* Short layers (P <= 16).  grid.x = groups of 64 rows: a row is a job, or for layer 0 a channel-frame (its input, the
 * pre-emphasised channel, is the same for every regulariser pass, so the lags are computed once and the Levinson kernels
 * read pass 0's copy).  A wave of the block owns 2 to 7 lags of one trial of the 64 rows (AcsWaves), ordered so that the
 * SIMDs of the CU carry about the same number of lags. */
template <int P> struct AcsWaves;
template <> struct AcsWaves<16> { static constexpr int NW = 8; };
template <> struct AcsWaves<8>  { static constexpr int NW = 5; };
template <> struct AcsWaves<4>  { static constexpr int NW = 3; };
template <> struct AcsWaves<2>  { static constexpr int NW = 2; };

template <int P, bool L0>
__global__ __launch_bounds__(64 * AcsWaves<P>::NW, 4) void k_autocorr_lane(Plan p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    using Cfg = AcCfg<P>;
    constexpr int NT = Cfg::NT, NW = AcsWaves<P>::NW;
    __shared__ double tile[2][ACS_T][65];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t rstride = L0 ? p.R : 1u, nrows = p.J / rstride, row0 = blockIdx.x * 64;
    uint32_t row = row0 + lane;
    const bool inrange = row < nrows;
    if (!inrange) row = nrows - 1;
    const uint32_t job = row * rstride;
    const uint32_t ci = p.cls_of_frame[(job / p.R) / p.C];
    const uint32_t ci0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
    const DevClass &c0 = p.cls[ci0];
    const uint32_t na = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.na);
    const bool fast = __all(ci == ci0) && c0.ntrials[layer] == (uint32_t)NT && (na % (4u << (NT - 1))) == 0 && (na % ACS_T) == 0;
#define ACS_RUN(T_, K_, J0_, JN_) autocorr_shared<K_, J0_, JN_, L0, NW>(p, layer, cur, row0, nrows, rstride, na, \
        (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.wt_off[layer][T_]), T_, wave, lane, tile)
    if (fast) {                                             /* every wave of the block sees the same rows: same decision */
        if (P == 16) switch (wave) {                        /* waves w and w+4 share a SIMD: 9 / 9 / 10 / 8 lags per SIMD */
            case 0: ACS_RUN(0, 17, 10, 7); break; case 1: ACS_RUN(0, 17, 0, 5); break;  case 2: ACS_RUN(0, 17, 5, 5); break;
            case 3: ACS_RUN(1, 9, 0, 5); break;   case 4: ACS_RUN(4, 2, 0, 2); break;   case 5: ACS_RUN(1, 9, 5, 4); break;
            case 6: ACS_RUN(2, 5, 0, 5); break;   default: ACS_RUN(3, 3, 0, 3); break;
        } else if (P == 8) switch (wave) {
            case 0: ACS_RUN(0, 9, 0, 5); break;   case 1: ACS_RUN(0, 9, 5, 4); break;   case 2: ACS_RUN(1, 5, 0, 5); break;
            case 3: ACS_RUN(2, 3, 0, 3); break;   default: ACS_RUN(3, 2, 0, 2); break;
        } else if (P == 4) switch (wave) {
            case 0: ACS_RUN(0, 5, 0, 5); break;   case 1: ACS_RUN(1, 3, 0, 3); break;   default: ACS_RUN(2, 2, 0, 2); break;
        } else switch (wave) {
            case 0: ACS_RUN(0, 3, 0, 3); break;   default: ACS_RUN(1, 2, 0, 2); break;
        }
        return;
    }
#undef ACS_RUN
    /* general form: the first NT waves take one whole trial each */
    if (wave >= (uint32_t)NT) return;
    const uint32_t t = wave;
    const bool active = inrange && (t < job_class(p, job).ntrials[layer]);
    const uint32_t q_end = na_max + Cfg::MAXPAD + 8;
    switch (P >> t) {       /* wave-uniform: the trial fixes the number of lags */
    case 16: if (P >= 16) autocorr_lane<17, L0>(p, layer, cur, q_end, job, t, active); break;
    case 8:  if (P >= 8)  autocorr_lane<9, L0>(p, layer, cur, q_end, job, t, active); break;
    case 4:  if (P >= 4)  autocorr_lane<5, L0>(p, layer, cur, q_end, job, t, active); break;
    case 2:  autocorr_lane<3, L0>(p, layer, cur, q_end, job, t, active); break;
    default: autocorr_lane<2, L0>(p, layer, cur, q_end, job, t, active); break;
    }
}

template <int P> static void launch_autocorr_small(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    const uint32_t nrows = (layer == 0) ? p.J / p.R : p.J;
    const dim3 grid((nrows + 63) / 64);
    if (layer == 0) hipLaunchKernelGGL((k_autocorr_lane<P, true>), grid, dim3(64 * AcsWaves<P>::NW), 0, st, p, layer, cur, na_max);
    else hipLaunchKernelGGL((k_autocorr_lane<P, false>), grid, dim3(64 * AcsWaves<P>::NW), 0, st, p, layer, cur, na_max);
}

template <int P> static void launch_autocorr2(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    using Cfg = AcCfg<P>;
    const uint32_t blocks = (p.J + Cfg::JPW - 1) / Cfg::JPW;
    if (layer == 0) hipLaunchKernelGGL((k_autocorr2<P, true>), dim3(blocks), dim3(64), 0, st, p, layer, cur, na_max);
    else hipLaunchKernelGGL((k_autocorr2<P, false>), dim3(blocks), dim3(64), 0, st, p, layer, cur, na_max);
}
static void dispatch_autocorr2(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    switch (p.P[layer]) {
    case 2: launch_autocorr_small<2>(st, p, layer, cur, na_max); break;
    case 4: launch_autocorr_small<4>(st, p, layer, cur, na_max); break;
    case 8: launch_autocorr_small<8>(st, p, layer, cur, na_max); break;
    case 16: launch_autocorr_small<16>(st, p, layer, cur, na_max); break;
    case 32: launch_autocorr2<32>(st, p, layer, cur, na_max); break;
    case 64: launch_autocorr2<64>(st, p, layer, cur, na_max); break;
    default: launch_autocorr2<128>(st, p, layer, cur, na_max); break;
    }
}

/* ridge + Levinson-Durbin per (trial, unit) (lpc.c:327-366, 578-633 with zero AF iterations), lanes = jobs.
 * Writes the coefficients in filter order (reversed, linne_network.c:310-316). */
__global__ void k_levinson(Plan p, uint32_t layer)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= p.J) return;
    const DevClass &c = job_class(p, job);
    uint32_t pr = blockIdx.y, t = 0;
    for (; t < c.ntrials[layer]; t++) {
        if (pr < c.trial_u[layer][t]) break;
        pr -= c.trial_u[layer][t];
    }
    if (t >= c.ntrials[layer]) return;
    const uint32_t P = p.P[layer], u = c.trial_u[layer][t], n = c.na / u, np = P / u, unit = pr;
    if (np >= 16u) return;                                  /* orders >= 16 are solved by k_levinson_wave */
    const uint32_t P0 = p.P[0];
    const double reg = p.regs[job % p.R];
    const uint32_t ajob = (layer == 0) ? job - job % p.R : job;          /* layer 0: lags are computed once per channel-frame */
    const double *r = p.acorr + ((size_t)ajob * LNN_MAXT + t) * LNN_ACW + (size_t)unit * (np + 1);
    double *h = p.tcoef + ((size_t)job * LNN_MAXT + t) * LNN_MAXP + (size_t)unit * np;
    double a[LNN_MAXP + 2];
    double tail = 0.0; int tail_set = 0;
    int zero = 0;
    if (n < np) {                                           /* lpc.c:349-355 */
        zero = 1;
    } else {
        const double r0 = r[0] * (1.0 + reg);               /* lpc.c:358 */
        if (fabs(r0) < (double)FLT_EPSILON) zero = 1;       /* lpc.c:271-276, 597-602 */
        else {
            double pc[LNN_MAXP + 1];
            levinson(r, r0, np, a, (layer + 1 == p.L) ? pc : nullptr);
            if (layer + 1 == p.L && np > P0) { tail = pc[P0]; tail_set = 1; }
        }
    }
    if (zero) {
        for (uint32_t k = 0; k < np; k++) h[k] = 0.0;
        if (np >= P0) { tail = 0.0; tail_set = 1; }         /* zero branches write parcor[0..order] */
    } else {
        for (uint32_t k = 0; k < np; k++) h[k] = a[np - k];
    }
    if (layer + 1 == p.L) {
        p.ptail[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = tail;
        p.ptail_set[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = (uint8_t)tail_set;
    }
}

/* Levinson-Durbin for the large problems (order >= 16), one wavefront per (job, trial, unit).  The coefficient vector
 * lives in LDS; the products a[i]*r[k+1-i] of a step are formed in parallel, their sum -- one chain in increasing i,
 * lpc.c:295-297 -- is added in order by every lane redundantly (so gamma needs no broadcast), and the pairwise update
 * a[i] <- a[i] + gamma*a[k+1-i] is element-parallel.  Same operations, same order as the scalar `levinson` above. */
#define LEV_WAVE_MIN_ORDER 16u
__global__ __launch_bounds__(64) void k_levinson_wave(Plan p, uint32_t layer)
{
    __shared__ __attribute__((aligned(16))) double sa[LNN_MAXP + 2], sr[LNN_MAXP + 2], sp[LNN_MAXP + 4];
    const uint32_t job = blockIdx.y, lane = threadIdx.x;
    const DevClass &c = job_class(p, job);
    uint32_t pr = blockIdx.x, t = 0;
    for (; t < c.ntrials[layer]; t++) {
        if (pr < c.trial_u[layer][t]) break;
        pr -= c.trial_u[layer][t];
    }
    if (t >= c.ntrials[layer]) return;
    const uint32_t P = p.P[layer], u = c.trial_u[layer][t], n = c.na / u, np = P / u, unit = pr, P0 = p.P[0];
    if (np < LEV_WAVE_MIN_ORDER) return;
    const double reg = p.regs[job % p.R];
    const uint32_t ajob = (layer == 0) ? job - job % p.R : job;          /* layer 0: lags are computed once per channel-frame */
    const double *r = p.acorr + ((size_t)ajob * LNN_MAXT + t) * LNN_ACW + (size_t)unit * (np + 1);
    double *h = p.tcoef + ((size_t)job * LNN_MAXT + t) * LNN_MAXP + (size_t)unit * np;
    const bool last = (layer + 1 == p.L);
    for (uint32_t i = lane; i <= np; i += 64) sr[i] = r[i];
    for (uint32_t i = lane; i < np + 2; i += 64) sa[i] = 0.0;
    __syncthreads();
    double tail = 0.0; int tail_set = 0;
    const double r0 = sr[0] * (1.0 + reg);
    const bool zero = (n < np) || (fabs(r0) < (double)FLT_EPSILON);
    if (zero) {
        for (uint32_t k = lane; k < np; k += 64) h[k] = 0.0;
        if (np >= P0) { tail = 0.0; tail_set = 1; }
    } else {
        const double r1 = sr[1];
        double ek = r0;
        const double a1 = -r1 / r0;
        ek += r1 * a1;
        if (lane == 0) { sa[0] = 1.0; sa[1] = a1; }
        __syncthreads();
        for (uint32_t k = 1; k < np; k++) {
            for (uint32_t i = lane; i <= k; i += 64) sp[i] = sa[i] * sr[k + 1 - i];
            __syncthreads();
            double gamma = 0.0;
            {   /* ordered sum, 16-byte LDS reads */
                uint32_t i = 0;
                for (; i + 4 <= k + 1; i += 4) {
                    const lnn_d2 v0 = *(const lnn_d2 *)(sp + i), v1 = *(const lnn_d2 *)(sp + i + 2);
                    gamma += v0.x; gamma += v0.y; gamma += v1.x; gamma += v1.y;
                }
                for (; i <= k; i++) gamma += sp[i];
            }
            gamma /= -ek;
            ek *= (1.0 - gamma * gamma);
            /* pairwise in-place update from the old vector: read, barrier, write */
            double na0 = 0.0, na1 = 0.0;
            const uint32_t i0 = lane + 1, i1 = lane + 65;
            if (i0 <= k) na0 = sa[i0] + gamma * sa[k + 1 - i0];
            if (i1 <= k) na1 = sa[i1] + gamma * sa[k + 1 - i1];
            __syncthreads();
            if (i0 <= k) sa[i0] = na0;
            if (i1 <= k) sa[i1] = na1;
            if (lane == 0) { sa[0] = 1.0 + gamma * 0.0; sa[k + 1] = 0.0 + gamma * 1.0; }
            __syncthreads();
            if (last && k == P0) { tail = -gamma; tail_set = 1; }
        }
        for (uint32_t k = lane; k < np; k += 64) h[k] = sa[np - k];
    }
    if (last && lane == 0) {
        p.ptail[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = tail;
        p.ptail_set[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = (uint8_t)tail_set;
    }
}

/* Levinson-Durbin for the large problems (order >= 16), lanes = jobs: a wavefront solves the SAME (trial, unit) problem of 64
 * consecutive jobs, each lane running the scalar recursion of `levinson` above on its own column of two LDS arrays
 * (a[i][lane], r[i][lane]: conflict-free 8-byte accesses).  Every wave instruction therefore advances 64 problems; the
 * ordered sum a[0]r[k+1] + ... + a[k]r[1] (lpc.c:295-297) is one chain per lane.  grid = (job groups, units of the trial). */
__global__ __launch_bounds__(64) void k_levinson_lds(Plan p, uint32_t layer, uint32_t t)
{
    extern __shared__ __attribute__((aligned(16))) double lev_lds[];
    const uint32_t lane = threadIdx.x, unit = blockIdx.y;
    uint32_t job = blockIdx.x * 64 + lane;
    const bool inrange = job < p.J;
    if (!inrange) job = p.J - 1;
    const DevClass &c = job_class(p, job);
    const uint32_t P = p.P[layer], u = 1u << t, np = P >> t, P0 = p.P[0];
    const bool have = inrange && t < c.ntrials[layer];
    const uint32_t n = c.na / u;
    double *sa = lev_lds + lane, *sr = lev_lds + (size_t)(np + 2) * 64 + lane;      /* element i at [i * 64] */
    const double reg = p.regs[job % p.R];
    const uint32_t ajob = (layer == 0) ? job - job % p.R : job;          /* layer 0: lags are computed once per channel-frame */
    const double *r = p.acorr + ((size_t)ajob * LNN_MAXT + t) * LNN_ACW + (size_t)unit * (np + 1);
    double *h = p.tcoef + ((size_t)job * LNN_MAXT + t) * LNN_MAXP + (size_t)unit * np;
    const bool last = (layer + 1 == p.L);
    for (uint32_t i = 0; i <= np; i++) sr[(size_t)i * 64] = have ? r[i] : 0.0;
    for (uint32_t i = 0; i < np + 2; i++) sa[(size_t)i * 64] = 0.0;
    double tail = 0.0; int tail_set = 0;
    const double r0 = sr[0] * (1.0 + reg);                    /* lpc.c:358 */
    const bool zero = (n < np) || (fabs(r0) < (double)FLT_EPSILON);      /* lpc.c:349-355, 271-276 */
    {   /* lanes with a zero problem (or none) run along on values nobody reads */
        const double r1 = sr[64];
        double ek = r0;
        const double a1 = -r1 / r0;
        ek += r1 * a1;
        sa[0] = 1.0; sa[64] = a1;
        for (uint32_t k = 1; k < np; k++) {
            double gamma = 0.0;
            {
                const double *pa = sa, *pr = sr + (size_t)(k + 1) * 64;
                uint32_t i = 0;
                for (; i + 4 <= k + 1; i += 4) {                /* four terms per trip, reads ahead of the adds */
                    const double a0 = pa[0], a1_ = pa[64], a2 = pa[128], a3 = pa[192];
                    const double q0 = pr[0], q1 = *(pr - 64), q2 = *(pr - 128), q3 = *(pr - 192);
                    gamma += a0 * q0; gamma += a1_ * q1; gamma += a2 * q2; gamma += a3 * q3;
                    pa += 256; pr -= 256;
                }
                for (; i <= k; i++) { gamma += pa[0] * pr[0]; pa += 64; pr -= 64; }
            }
            gamma /= -ek;
            ek *= (1.0 - gamma * gamma);
            const double a0n = 1.0 + gamma * 0.0;              /* u[0]   + gamma*v[0]   */
            const double ak1 = 0.0 + gamma * 1.0;              /* u[k+1] + gamma*v[k+1] */
            uint32_t i = 1, j = k;
            while (i < j) {
                const double ai = sa[(size_t)i * 64], aj = sa[(size_t)j * 64];
                sa[(size_t)i * 64] = ai + gamma * aj;
                sa[(size_t)j * 64] = aj + gamma * ai;
                i++; j--;
            }
            if (i == j) { const double ai = sa[(size_t)i * 64]; sa[(size_t)i * 64] = ai + gamma * ai; }
            sa[0] = a0n; sa[(size_t)(k + 1) * 64] = ak1;
            if (last && k == P0) { tail = -gamma; tail_set = 1; }
        }
    }
    if (!have) return;
    if (zero) {
        for (uint32_t k = 0; k < np; k++) h[k] = 0.0;
        tail = 0.0; tail_set = (np >= P0) ? 1 : 0;             /* zero branches write parcor[0..order] */
    } else {
        for (uint32_t k = 0; k < np; k++) h[k] = sa[(size_t)(np - k) * 64];
        if (!(last && np > P0)) { tail = 0.0; tail_set = 0; }
    }
    if (last) {
        p.ptail[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = tail;
        p.ptail_set[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = (uint8_t)tail_set;
    }
}

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

#define FIR_THREADS 256
#define FIR_SPL     8                       /* consecutive samples per lane */
#define FIR_TILE    (FIR_THREADS * FIR_SPL)
template <int MODE, bool L0>
__global__ __launch_bounds__(FIR_THREADS, 4) void k_fir2(Plan p, uint32_t layer, uint32_t cur)
{
    __shared__ __attribute__((aligned(16))) double xs[LNN_MAXP + FIR_TILE + 8];
    __shared__ __attribute__((aligned(16))) double hs[(MODE == 1) ? 1 : LNN_MAXT][LNN_MAXP + 8];   /* every trial's coefficients; +8: the pipelined loop reads one step ahead */
    __shared__ __attribute__((aligned(16))) double ob[(MODE == 2) ? 1 : FIR_THREADS / 64][(MODE == 2) ? 2 : 64 * FIR_SPL];   /* per-wave store transpose (MODE 0/1) */
    __shared__ double chain[LNN_MAXT];                              /* MODE 0: the ordered sums, carried across tiles */
    const uint32_t job = blockIdx.x, tid = threadIdx.x;          /* grid = (jobs, tiles): the job count is not bound by 65535 */
    if (MODE == 0 && !p.uncertain[job]) return;                     /* exact search only where the certified one gave up */
    const DevClass &c = job_class(p, job);
    const uint32_t na = c.na;
    const uint32_t P = p.P[layer];
    const double *x = p.sig + ((size_t)job * 2 + cur) * p.S;
    const int32_t *xi = p.xint + (size_t)(job / p.R) * p.S;          /* layer 0 reads the pre-emphasised int32 channel (linne_encoder.c:661-663) */
    const uint32_t ntr = (MODE != 1) ? c.ntrials[layer] : 1u;
    if (MODE == 0 && tid < LNN_MAXT) chain[tid] = 0.0;
    {
        const double *hsrc = (MODE != 1) ? (p.tcoef + (size_t)job * LNN_MAXT * LNN_MAXP) : (p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP);
        for (uint32_t i = tid; i < ntr * LNN_MAXP; i += FIR_THREADS) { const uint32_t tt = i / LNN_MAXP, k = i % LNN_MAXP; if (k < P) hs[tt][k] = hsrc[i]; }
    }
    /* MODE 0 walks every tile of the job in order inside one block; MODE 1/2 take one tile per block */
    for (uint32_t s0 = (MODE == 0) ? 0u : blockIdx.y * FIR_TILE; s0 < na; s0 += (MODE == 0) ? FIR_TILE : 0x40000000u) {
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
    const double *xc = xs + LNN_MAXP + FIR_SPL * tid;                /* -> x[s], 16-byte aligned */
    for (uint32_t t = 0; t < ntr; t++) {
        const uint32_t u = (MODE != 1) ? c.trial_u[layer][t] : p.lunits[(size_t)job * LNN_MAXL + layer];
        const uint32_t n = na / u, np = P / u;
        const double *hbuf = hs[t];
        if (t == 0) __syncthreads();                                 /* tile and coefficients are staged */
        double acc[FIR_SPL];
        if (s < na) {
            /* all FIR_SPL samples in one unit, every tap present */
            const bool whole = ((n & (FIR_SPL - 1)) == 0) && (s >= np) && (s + FIR_SPL - 1 < na);
            if (whole && (np & 3u) == 0) {
                const double *hb = hbuf + (size_t)(s / n) * np;
                const double *xw = xc - np;                              /* -> x[s - np] */
                /* Window x[s-np+k .. +11] in a register ring of 16 (element e lives in w[e % 16]): a step of 4 taps reads
                 * 11 of them, the LDS reads of the next step's 4 new samples and coefficients land in the free quarter
                 * while the 32 multiply-adds of this step issue, and nothing is ever moved. */
                double w[16];
                static_assert(FIR_SPL == 8, "the ring below is laid out for 8 samples per lane");
#pragma unroll
                for (int j = 0; j < 12; j += 2) { const lnn_d2 v = *(const lnn_d2 *)(xw + j); w[j] = v.x; w[j + 1] = v.y; }
#pragma unroll
                for (int j = 0; j < FIR_SPL; j++) acc[j] = (MODE != 1) ? xc[j] : 0.0;
                lnn_d2 ha0 = *(const lnn_d2 *)(hb), ha1 = *(const lnn_d2 *)(hb + 2), hb0, hb1;
                uint32_t k = 0;
#define FIR_STEP(G, HC0, HC1, HN0, HN1) { \
                    const lnn_d2 na_ = *(const lnn_d2 *)(xw + k + 12), nb_ = *(const lnn_d2 *)(xw + k + 14);   /* in bounds: xs/hs are padded */ \
                    HN0 = *(const lnn_d2 *)(hb + k + 4); HN1 = *(const lnn_d2 *)(hb + k + 6); \
                    w[(4 * G + 12) % 16] = na_.x; w[(4 * G + 13) % 16] = na_.y; w[(4 * G + 14) % 16] = nb_.x; w[(4 * G + 15) % 16] = nb_.y; \
                    const double hh_[4] = { HC0.x, HC0.y, HC1.x, HC1.y }; \
                    _Pragma("unroll") for (int kk = 0; kk < 4; kk++) { \
                        _Pragma("unroll") for (int j = 0; j < FIR_SPL; j++) acc[j] += hh_[kk] * w[(4 * G + kk + j) % 16]; } \
                    k += 4; }
                for (;;) {
                    FIR_STEP(0, ha0, ha1, hb0, hb1); if (k >= np) break;
                    FIR_STEP(1, hb0, hb1, ha0, ha1); if (k >= np) break;
                    FIR_STEP(2, ha0, ha1, hb0, hb1); if (k >= np) break;
                    FIR_STEP(3, hb0, hb1, ha0, ha1); if (k >= np) break;
                }
#undef FIR_STEP
            } else if (whole && np <= 2) {
                const double *hb = hbuf + (size_t)(s / n) * np;
                const double h0 = hb[0];
                if (np == 1) {
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j++) { acc[j] = (MODE != 1) ? xc[j] : 0.0; acc[j] += h0 * xc[j - 1]; }
                } else {
                    const double h1 = hb[1];
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j++) { acc[j] = (MODE != 1) ? xc[j] : 0.0; acc[j] += h0 * xc[j - 2]; acc[j] += h1 * xc[j - 1]; }
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < FIR_SPL; j++) {
                    const uint32_t sj = s + j;
                    double v = (MODE != 1) ? xc[j] : 0.0;
                    if (sj < na && sj != 0) {
                        const double *hb = hbuf + (size_t)(sj / n) * np;
                        const uint32_t kstart = (sj < np) ? (np - sj) : 0;  /* taps before sample 0 are skipped */
                        for (uint32_t k = kstart; k < np; k++) v += hb[k] * xc[(int)j - (int)np + (int)k];
                    }
                    acc[j] = v;
                }
            }
            /* results: |residual| (MODE 0) or x + predict (MODE 1) */
#pragma unroll
            for (int j = 0; j < FIR_SPL; j++) {
                if (MODE != 1) { double av = (acc[j] > 0) ? acc[j] : -acc[j]; if (s + j == 0) av = 0.0; acc[j] = av; }
                else { const double xv = xc[j]; acc[j] = (s + j == 0) ? xv : (xv + acc[j]); }
            }
        }
        if (MODE == 2) {
            /* order-free partial sum of this wave's |residual| values; the exact ordered chain is evaluated later only for
             * jobs whose argmin these sums cannot certify (k_select) */
            double ps = 0.0;
            if (s < na) {
#pragma unroll
                for (int j = 0; j < FIR_SPL; j++) if (s + j < na) ps += acc[j];
            }
            ps = wave_sum_f64_lane63(ps);
            if ((tid & 63u) == 63u) p.tsum[((size_t)job * LNN_MAXT + t) * p.npart + blockIdx.y * (FIR_THREADS / 64) + (tid >> 6)] = ps;
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
        } else {   /* coalesced store: the wave's 64*FIR_SPL consecutive results go through LDS so that consecutive lanes write
             * consecutive 16-byte pieces (a lane's own 8 results are 64 bytes apart from its neighbour's) */
            const uint32_t wv = tid >> 6, ln = tid & 63u, wbase = s0 + wv * 64 * FIR_SPL;
            double *dst = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S;
            if (wbase < na) {
                if (s < na) {
#pragma unroll
                    for (int j = 0; j < FIR_SPL; j += 2) { lnn_d2 v; v.x = acc[j]; v.y = acc[j + 1]; *(lnn_d2 *)(&ob[wv][ln * FIR_SPL + j]) = v; }
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
        }
    }
    }
    if (MODE == 0) {
        __syncthreads();
        if (tid < ntr) p.tloss[(size_t)job * LNN_MAXT + tid] = chain[tid] / (double)na;
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
            if (MODE == 1 || ((myrow % LNN_MAXT) < c.ntrials[layer] && p.uncertain[job])) {
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

/* strict-< argmin over the trials (linne_network.c:338-341), keep its coefficients (== SetParameter,
 * :350-376, which recomputes the same values) and, for the last layer, the value the layer leaves in
 * parcor[P0] (Q2): the last call in reference order -- trials in order, then SetParameter's units -- that
 * wrote it. */
__global__ void k_select(Plan p, uint32_t layer, uint32_t exact)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= p.J) return;
    if (exact && !p.uncertain[job]) return;
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
        for (uint32_t t = 0; t < nt; t++) {
            double sm = 0.0;
            const uint32_t np_used = ((c.na + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64);
            const double *ps = p.tsum + ((size_t)job * LNN_MAXT + t) * p.npart;
            for (uint32_t i = 0; i < np_used; i++) sm += ps[i];
            m[t] = sm / (double)c.na;
            if (!(m[t] >= 0.0) || !(m[t] < (double)FLT_MAX)) ok = 0;
            if (m[t] < min_loss) { min_loss = m[t]; best = t; }
        }
        for (uint32_t t = 0; t < nt; t++) if (t != best && !(m[t] * (1.0 - rel) > min_loss * (1.0 + rel))) ok = 0;
        p.uncertain[job] = ok ? 0 : 1;
        if (!ok) atomicAdd(p.ucount, 1u);
    }
    const uint32_t P = p.P[layer];
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

/* ------------------------------------------------------------------------------------------------
 * finalize per channel-frame: best regulariser, quantisation, int32 FIR cascade
 * ---------------------------------------------------------------------------------------------- */
#define FIN_THREADS 256
__global__ __launch_bounds__(FIN_THREADS) void k_finalize(Plan p)
{
    __shared__ int32_t s_coef[LNN_MAXL][LNN_MAXP];
    __shared__ uint32_t s_rshift[LNN_MAXL], s_units[LNN_MAXL], s_best;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t n = c.n, S = p.S;
    int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    double *st = p.stats + (size_t)cf * LINNE_AMD_STAT_WORDS;

    if (tid == 0) {     /* linne_network.c:618-626 */
        double min_loss = (double)FLT_MAX; uint32_t best = 0;
        for (uint32_t r = 0; r < p.R; r++) { const double l = p.jloss[(size_t)cf * p.R + r]; if (l < min_loss) { min_loss = l; best = r; } }
        s_best = best;
        st[LINNE_AMD_ST_TAIL] = p.jtail[(size_t)cf * p.R + best];
        st[LINNE_AMD_ST_BEST] = (double)best;
        st[LINNE_AMD_ST_LOSS] = p.jloss[(size_t)cf * p.R + best];
    }
    __syncthreads();
    const uint32_t job = cf * p.R + s_best;
    if (tid < p.L) {    /* lpc.c:981-1040 over all units of the layer together */
        const uint32_t l = tid, P = p.P[l];
        const double *d = p.lparams + ((size_t)job * LNN_MAXL + l) * LNN_MAXP;
        double mx = 0.0;
        for (uint32_t k = 0; k < P; k++) if (mx < fabs(d[k])) mx = fabs(d[k]);
        uint32_t rshift;
        if (mx <= 0.0078125) {                              /* 2^-(8-1) */
            rshift = 8;
            for (uint32_t k = 0; k < P; k++) s_coef[l][k] = 0;
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
                s_coef[l][k] = q;
            }
        }
        s_rshift[l] = rshift;
        s_units[l] = p.lunits[(size_t)job * LNN_MAXL + l];
        rec[LINNE_AMD_PRM_UNITS + l] = (int32_t)s_units[l];
        rec[LINNE_AMD_PRM_RSHIFT + l] = (int32_t)rshift;
        for (uint32_t k = 0; k < P; k++) rec[LINNE_AMD_PRM_COEF + p.coef_off[l] + k] = s_coef[l][k];
    }
    __syncthreads();
    /* FIR cascade (linne_encoder.c:687-696, linne_lpc_predict.c:7-38) on the n valid samples; each layer streams the
     * channel through an LDS tile (1024 samples + 128 of history) so the tap loop reads LDS, not global memory */
    __shared__ int32_t xt[LNN_MAXP + 4 * FIN_THREADS];
    int32_t *src = p.xint + (size_t)cf * S, *dst = p.xtmp + (size_t)cf * S;
    for (uint32_t l = 0; l < p.L; l++) {
        const uint32_t units = s_units[l], np = p.P[l] / units, ns = n / units, rs = s_rshift[l];
        const uint32_t half = 1u << ((rs - 1u) & 31u);
        int32_t *out = (l + 1 == p.L) ? (p.resid + (size_t)cf * S) : dst;
        for (uint32_t s0 = 0; s0 < n; s0 += 4 * FIN_THREADS) {
            __syncthreads();
            for (uint32_t i = tid; i < LNN_MAXP + 4 * FIN_THREADS; i += FIN_THREADS) {
                const int64_t g = (int64_t)s0 - LNN_MAXP + i;
                xt[i] = (g >= 0 && g < (int64_t)n) ? src[g] : 0;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t e = tid + j * FIN_THREADS, s = s0 + e;
                if (s >= n) continue;
                int32_t v = xt[LNN_MAXP + e];
                const uint32_t unit = s / (ns ? ns : 1u);
                if (ns >= np && unit < units) {
                    const uint32_t loc = s - unit * ns;
                    if (loc >= np) {
                        uint32_t pred = half;
                        const int32_t *cc = s_coef[l] + unit * np;
                        const int32_t *xx = xt + LNN_MAXP + e - np;
                        for (uint32_t k = 0; k < np; k++) pred += (uint32_t)cc[k] * (uint32_t)xx[k];
                        v = (int32_t)((uint32_t)v + (uint32_t)((int32_t)pred >> (rs & 31u)));
                    }
                }
                out[s] = v;
            }
        }
        if (l + 1 == p.L) for (uint32_t s = n + tid; s < S; s += FIN_THREADS) out[s] = 0;
        __syncthreads();
        if (l + 1 < p.L) { int32_t *t = src; src = dst; dst = t; }
    }
}

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
 * channel-frames; lane g of a channel-frame owns the taps k = g (mod 4) of the zero-extended coefficient vector, in
 * registers as doubles.  The last PL outputs live in LDS as doubles in a ring stored twice (slot i and i + PL), so a
 * lane's taps are a strided run without wrap-around, read with immediate offsets; the row stride and the tap
 * interleaving put the 32 lanes of each LDS access group on 32 different banks.  The int32 dot product is evaluated in
 * FP64 (exact, see k_synth_small), the four partial sums of a channel-frame meet through two DPP quad permutes, and
 * every lane of the quad finishes the step redundantly.  The newest output is forwarded in a register (its tap belongs
 * to lane 3), so the LDS write -> read round trip is off the critical path. */
#define SYB_T 64
template <int PL>
__global__ __launch_bounds__(64) void k_synth_big(DecPlan p, uint32_t layer)
{
    constexpr int TP = PL / 4;                                   /* taps per lane */
    constexpr int RSTR = 2 * PL + 4;                             /* ring row stride in doubles: = 4 (mod 32) */
    __shared__ __attribute__((aligned(16))) double ring[16 * RSTR];
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
    double c[TP];
#pragma unroll
    for (int j = 0; j < TP; j++) c[j] = 0.0;
    double *myring = ring + cfl * RSTR;
    for (uint32_t i = g; i < 2 * PL; i += 4) myring[i] = 0.0;
    uint32_t tl = 0, unit = 0;
    bool fresh = true;
    double ynew = 0.0, part = 0.0;                               /* the previous step's output, forwarded; the partial sum made ahead */
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
#pragma unroll 1
        for (uint32_t s = 0; s < SYB_T; s++) {
            const uint32_t sidx = t * SYB_T + s, tm = sidx & (PL - 1);
            if (fresh && !skip && unit < units) {                /* first sample of a unit: my taps of its zero-extended coefficients */
#pragma unroll
                for (int j = 0; j < TP; j++) {
                    const uint32_t k = 4u * (uint32_t)j + g;
                    c[j] = (k >= PL - np) ? (double)crec[unit * np + (k - (PL - np))] : 0.0;
                }
            }
            fresh = false;
            const int32_t res = tile[cfl][s];
            /* tap k = 4j + g multiplies y[sidx - PL + k] (ring position tm + k).  Everything but the newest tap (k = PL - 1,
             * lane 3) was summed during the previous step (`part`); only that one product is on this step's critical path */
            double acc = (g == 3u) ? __builtin_fma(c[TP - 1], ynew, part) : part;
            {   /* quad sum: lanes 4q .. 4q+3 */
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc), 0xB1, 0xf, 0xf, true);     /* quad_perm: 1,0,3,2 */
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc), 0xB1, 0xf, 0xf, true);
                acc += __hiloint2double(hi, lo);
                lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc), 0x4E, 0xf, 0xf, true);         /* quad_perm: 2,3,0,1 */
                hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc), 0x4E, 0xf, 0xf, true);
                acc += __hiloint2double(hi, lo);
            }
            {   /* next step's partial sum: positions tm + 1 + k, k <= PL - 2 -- none of them is written by this step */
                const double *hp = myring + ((tm + 1u) & (PL - 1)) + g;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int j = 0; j < TP - 1; j++) {
                    const double hv = hp[4 * j];
                    if ((j & 3) == 0) a0 = __builtin_fma(c[j], hv, a0);
                    else if ((j & 3) == 1) a1 = __builtin_fma(c[j], hv, a1);
                    else if ((j & 3) == 2) a2 = __builtin_fma(c[j], hv, a2);
                    else a3 = __builtin_fma(c[j], hv, a3);
                }
                const double hl = (g == 3u) ? 0.0 : hp[4 * (TP - 1)];
                a3 = __builtin_fma(c[TP - 1], hl, a3);
                part = (a0 + a1) + (a2 + a3);
            }
            const uint32_t sum32 = (uint32_t)__double2loint(acc + 6755399441055744.0);      /* acc mod 2^32 (see k_synth_small) */
            const uint32_t pred = half + sum32;
            int32_t y = res;
            if (!skip && tl >= np && unit < units) y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)pred >> (rs & 31u)));
            ynew = (double)y;
            if (g == 0) { myring[tm] = ynew; myring[tm + PL] = ynew; tile[cfl][s] = y; }
            tl++;
            if (tl == ns) { tl = 0; unit++; fresh = true; }
        }
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


/* ================================================================================================
 * Rice planning (linne_coder.c:217-279): one block per channel-frame.
 * Integer sums are exact, the means repeat the host's divisions ((double)sum / ns, then pairwise (a + b) / 2.0), the
 * parameter is a search in the table of steps the host located with its libm (a mean inside a guard band raises the
 * record's flag and the host searches that channel-frame itself), code lengths are uint32 with wrap-around.
 * ============================================================================================== */
#define RICE_THREADS 256
struct RicePlanArgs {
    const int32_t *resid; const uint32_t *nsmp; uint8_t *plan; uint32_t C, S, nsteps;
    double steps[32];
};
__device__ __forceinline__ uint32_t rp_zz(int32_t v) { const uint32_t d = (uint32_t)v << 1; return (v < 0) ? ((0u - d) - 1u) : d; }
__device__ __forceinline__ uint32_t rp_gamma_len(uint32_t u) { return u ? (2u * (32u - (uint32_t)__clz((int)(u + 1u))) - 1u) : 1u; }   /* 2*ceil_log2(u+2)-1 */

__global__ __launch_bounds__(RICE_THREADS) void k_rice_plan(RicePlanArgs a)
{
    __shared__ double mean[2048];            /* level o (2^o partitions) at [2^o - 1, 2^(o+1) - 1) */
    __shared__ uint8_t kk[2048];
    __shared__ uint32_t tot[12];
    __shared__ uint32_t flag, best_s;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = a.nsmp[cf / a.C];
    const int32_t *x = a.resid + (size_t)cf * a.S;
    uint8_t *rec = a.plan + (size_t)cf * LINNE_AMD_RICE_PLAN_BYTES;
    uint32_t max_order = 1;
    while (max_order <= 11 && (n % (1u << max_order)) == 0) max_order++;
    max_order = (max_order - 1 < 10u) ? max_order - 1 : 10u;
    const uint32_t parts = 1u << max_order, nsf = n / parts;
    if (tid < 12) tot[tid] = 0;
    if (tid == 0) flag = 0;
    for (uint32_t p = tid; p < parts; p += RICE_THREADS) {
        const int32_t *q = x + (size_t)p * nsf;
        uint64_t sum = 0;
        for (uint32_t j = 0; j < nsf; j++) sum += rp_zz(q[j]);
        mean[parts - 1 + p] = (double)sum / (double)nsf;
    }
    __syncthreads();
    for (int o = (int)max_order - 1; o >= 0; o--) {
        const uint32_t base = (1u << o) - 1u, cbase = (2u << o) - 1u;
        for (uint32_t p = tid; p < (1u << o); p += RICE_THREADS) mean[base + p] = (mean[cbase + 2 * p] + mean[cbase + 2 * p + 1]) / 2.0;
        __syncthreads();
    }
    const uint32_t nent = 2u * parts - 1u;
    for (uint32_t e = tid; e < nent; e += RICE_THREADS) {
        const double m = mean[e];
        uint32_t k = 0;
        for (uint32_t i = 0; i < a.nsteps; i++) k += (m >= a.steps[i]) ? 1u : 0u;
        bool guard = !(m >= 0.0);
        if (k < a.nsteps && m >= a.steps[k] * (1.0 - LNN_RICE_GUARD)) guard = true;
        if (k > 0 && m <= a.steps[k - 1] * (1.0 + LNN_RICE_GUARD)) guard = true;
        if (guard) atomicOr(&flag, 1u);
        kk[e] = (uint8_t)(k & 31u);
    }
    __syncthreads();
    /* per entry: the samples' fixed part and the parameter's own code */
    for (uint32_t e = tid; e < nent; e += RICE_THREADS) {
        const uint32_t o = 31u - (uint32_t)__clz((int)(e + 1u)), p = e - ((1u << o) - 1u);
        const uint32_t k = kk[e];
        uint32_t bits = (n >> o) * (k + 2u);
        bits += p ? rp_gamma_len(rp_zz((int32_t)k - (int32_t)kk[e - 1])) : 5u;
        atomicAdd(&tot[o], bits);
    }
    /* per finest partition: the excess of its samples under the parameter of each order's enclosing partition */
    for (uint32_t p = tid; p < parts; p += RICE_THREADS) {
        const int32_t *q = x + (size_t)p * nsf;
        uint32_t kc[11], acc[11];
#pragma unroll
        for (uint32_t o = 0; o < 11; o++) { acc[o] = 0; kc[o] = (o <= max_order) ? kk[((1u << o) - 1u) + (p >> (max_order - o))] : 0u; }
        for (uint32_t j = 0; j < nsf; j++) {
            const uint32_t v = rp_zz(q[j]);
#pragma unroll
            for (uint32_t o = 0; o < 11; o++) { const uint32_t k1pow = 1u << ((kc[o] + 1u) & 31u); acc[o] += ((v > k1pow) ? (v - k1pow) : 0u) >> kc[o]; }
        }
#pragma unroll
        for (uint32_t o = 0; o < 11; o++) if (o <= max_order) atomicAdd(&tot[o], acc[o]);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0, min_bits = 0xFFFFFFFFu;
        for (uint32_t o = 0; o <= max_order; o++) if (min_bits > tot[o]) { min_bits = tot[o]; best = o; }
        best_s = best;
        rec[0] = (uint8_t)best; rec[1] = (uint8_t)flag;
    }
    __syncthreads();
    const uint32_t best = best_s;
    for (uint32_t p = tid; p < (1u << best); p += RICE_THREADS) rec[LINNE_AMD_RICE_PLAN_K2 + p] = kk[((1u << best) - 1u) + p];
}

/* ================================================================================================
 * host side of this TU: context, scratch arena, launch sequences, C-ABI
 * ============================================================================================== */
struct LINNEAmdContext {
    int device;
    hipStream_t stream; int own_stream;
    void *arena; uint64_t arena_bytes;
    char err[256];
    int timing;
    uint32_t na_max;                    /* largest analysis length of the current batch */
    hipEvent_t ev[2]; int ev_valid;
    /* per-kernel spans of the last call (timing enabled): HIP events on the launch stream */
    hipEvent_t *span_ev; int *span_kind; int nspans, span_cap;
    /* cached class tables */
    uint32_t *d_ucount;
    /* frame groups of one call rotate over these streams so that the latency-bound phases of one group (short
     * layers, Levinson, ordered sums) overlap the throughput-bound phases of another */
    hipStream_t sub[LNN_MAXSUB]; hipEvent_t sub_done[LNN_MAXSUB]; hipEvent_t ev_start; int nsub;
    hipStream_t side; hipEvent_t side_done; int has_side;     /* block-type statistics run beside the analysis */
    DevClass *d_cls; double *d_sin; uint64_t sin_cap; double *d_wt; uint64_t wt_cap; uint32_t *d_clsidx; uint64_t clsidx_cap; uint32_t *d_nsmp; uint64_t nsmp_cap;
    /* what the resident class tables were built for: a call with the same shape and frame lengths re-uses them */
    DevClass sig_cls[LNN_MAXCLS]; struct LINNEAmdShape sig_shape; int sig_for_encode, sig_valid;
    /* pinned ring for the per-call frame metadata (class index, length), so that a call enqueues without a host sync */
    uint32_t *meta_h[LNN_META]; uint64_t meta_cap[LNN_META]; hipEvent_t meta_ev[LNN_META]; int meta_used[LNN_META]; int meta_next;
    /* copy streams of the staging slots (H2D of the next group and D2H of the previous one overlap the kernels) */
    hipStream_t copy_in, copy_out; int has_copy;
    uint32_t *d_plan_nsmp; uint64_t plan_nsmp_cap; double rice_steps[32]; uint32_t rice_nsteps;
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return LNN_NG; } } while (0)

static const uint32_t k_layers_a[] = { 2, 32 }, k_layers_b[] = { 4, 64, 8 }, k_layers_c[] = { 4, 128, 16 };
static const double k_regs_1[] = { 0.0 }, k_regs_2[] = { 0.0, 1.0 / 512.0 }, k_regs_4[] = { 0.0, 1.0 / 2048.0, 1.0 / 512.0, 1.0 / 128.0 };

extern "C" int lnn_preset_info(uint32_t preset, uint32_t *num_layers, uint32_t *layers, uint32_t *num_regs, double *regs)
{   /* libs/linne_internal/src/linne_internal.c:16-41 */
    const uint32_t *L; const double *R; uint32_t nl, nr;
    if (preset >= 8) return -1;
    if (preset < 2) { L = k_layers_a; nl = 2; } else if (preset < 5) { L = k_layers_b; nl = 3; } else { L = k_layers_c; nl = 3; }
    switch (preset) { case 0: case 2: case 5: R = k_regs_1; nr = 1; break; case 1: case 3: case 6: R = k_regs_2; nr = 2; break; default: R = k_regs_4; nr = 4; }
    if (num_layers) *num_layers = nl;
    if (layers) for (uint32_t i = 0; i < nl; i++) layers[i] = L[i];
    if (num_regs) *num_regs = nr;
    if (regs) for (uint32_t i = 0; i < nr; i++) regs[i] = R[i];
    return 0;
}

extern "C" int LINNEAmd_GetDeviceCount(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" struct LINNEAmdContext *LINNEAmd_ContextCreate(int device, uint64_t scratch_bytes)
{
    int n = 0;
    hipError_t e;
#define CC_FAIL(what) do { fprintf(stderr, "liblinne_amd: ContextCreate(device=%d): %s: %s\n", device, what, hipGetErrorString(e)); } while (0)
    if ((e = hipGetDeviceCount(&n)) != hipSuccess) { CC_FAIL("hipGetDeviceCount"); return NULL; }
    if (n <= 0 || device < 0 || device >= n) { fprintf(stderr, "liblinne_amd: ContextCreate(device=%d): %d HIP device(s) visible\n", device, n); return NULL; }
    if ((e = hipSetDevice(device)) != hipSuccess) { CC_FAIL("hipSetDevice"); return NULL; }
    LINNEAmdContext *ctx = (LINNEAmdContext *)calloc(1, sizeof(*ctx));
    if (!ctx) return NULL;
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) { CC_FAIL("hipStreamCreate"); free(ctx); return NULL; }
    ctx->own_stream = 1;
    if (scratch_bytes == 0) scratch_bytes = 6ull << 30;
    if ((e = hipMalloc(&ctx->arena, scratch_bytes)) != hipSuccess) { CC_FAIL("hipMalloc(arena)"); hipStreamDestroy(ctx->stream); free(ctx); return NULL; }
    ctx->arena_bytes = scratch_bytes;
    if ((e = hipMalloc((void **)&ctx->d_cls, sizeof(DevClass) * LNN_MAXCLS)) != hipSuccess) { CC_FAIL("hipMalloc(classes)"); hipFree(ctx->arena); hipStreamDestroy(ctx->stream); free(ctx); return NULL; }
    if ((e = hipMalloc((void **)&ctx->d_ucount, sizeof(uint32_t))) != hipSuccess) { CC_FAIL("hipMalloc(counter)"); }
    {
        const char *env = getenv("LINNE_AMD_STREAMS");
        int ns = env ? atoi(env) : 1;
        if (ns < 1) ns = 1;
        if (ns > LNN_MAXSUB) ns = LNN_MAXSUB;
        ctx->nsub = 0;
        for (int i = 0; i < ns; i++) {
            if (hipStreamCreateWithFlags(&ctx->sub[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->sub_done[i], hipEventDisableTiming) != hipSuccess) break;
            ctx->nsub++;
        }
        if (hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming) != hipSuccess) ctx->nsub = 0;
        ctx->has_side = (ctx->nsub > 0) && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) == hipSuccess
                && hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming) == hipSuccess;
    }
    (void)hipFuncSetAttribute((const void *)k_levinson_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 64 * (2 * LNN_MAXP + 3)));
    if ((e = hipEventCreate(&ctx->ev[0])) != hipSuccess || (e = hipEventCreate(&ctx->ev[1])) != hipSuccess) { CC_FAIL("hipEventCreate"); }
#undef CC_FAIL
    return ctx;
}

extern "C" void LINNEAmd_ContextDestroy(struct LINNEAmdContext *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->d_cls) hipFree(ctx->d_cls);
    if (ctx->d_ucount) hipFree(ctx->d_ucount);
    for (int i = 0; i < ctx->nsub; i++) { hipStreamSynchronize(ctx->sub[i]); hipStreamDestroy(ctx->sub[i]); hipEventDestroy(ctx->sub_done[i]); }
    if (ctx->nsub) hipEventDestroy(ctx->ev_start);
    if (ctx->has_side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); hipEventDestroy(ctx->side_done); }
    if (ctx->d_sin) hipFree(ctx->d_sin);
    if (ctx->d_wt) hipFree(ctx->d_wt);
    if (ctx->d_clsidx) hipFree(ctx->d_clsidx);
    if (ctx->d_nsmp) hipFree(ctx->d_nsmp);
    if (ctx->d_plan_nsmp) hipFree(ctx->d_plan_nsmp);
    for (int i = 0; i < LNN_META; i++) { if (ctx->meta_h[i]) hipHostFree(ctx->meta_h[i]); if (ctx->meta_ev[i]) hipEventDestroy(ctx->meta_ev[i]); }
    if (ctx->has_copy) { hipStreamSynchronize(ctx->copy_in); hipStreamSynchronize(ctx->copy_out); hipStreamDestroy(ctx->copy_in); hipStreamDestroy(ctx->copy_out); }
    hipEventDestroy(ctx->ev[0]); hipEventDestroy(ctx->ev[1]);
    for (int i = 0; i < 2 * ctx->span_cap; i++) hipEventDestroy(ctx->span_ev[i]);
    free(ctx->span_ev); free(ctx->span_kind);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    free(ctx);
}

extern "C" const char *LINNEAmd_GetLastError(const struct LINNEAmdContext *ctx) { return ctx ? ctx->err : "no context"; }

extern "C" int LINNEAmd_SetStream(struct LINNEAmdContext *ctx, void *hip_stream)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { HIPCHK(ctx, hipStreamDestroy(ctx->stream)); ctx->own_stream = 0; }
    ctx->stream = (hipStream_t)hip_stream;              /* NULL is the device's default (null) stream */
    return LNN_OK;
}

extern "C" int LINNEAmd_ReserveScratch(struct LINNEAmdContext *ctx, uint64_t bytes)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    if (ctx->arena_bytes >= bytes) return LNN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    void *fresh = NULL;
    if (hipMalloc(&fresh, bytes) != hipSuccess) { (void)hipGetLastError(); snprintf(ctx->err, sizeof(ctx->err), "cannot reserve %llu bytes of scratch", (unsigned long long)bytes); return LNN_NG; }
    if (ctx->arena) HIPCHK(ctx, hipFree(ctx->arena));
    ctx->arena = fresh; ctx->arena_bytes = bytes;
    return LNN_OK;
}

extern "C" int64_t LINNEAmd_GetLastFallbackCount(struct LINNEAmdContext *ctx)
{
    if (!ctx) return -1;
    uint32_t v = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess
            || hipMemcpy(&v, ctx->d_ucount, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

extern "C" int LINNEAmd_Synchronize(struct LINNEAmdContext *ctx)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LNN_OK;
}

extern "C" int LINNEAmd_EnableTiming(struct LINNEAmdContext *ctx, int enable) { if (!ctx) return LNN_INVALID_ARGUMENT; ctx->timing = enable; return LNN_OK; }
/* span bookkeeping: span_begin/span_end bracket one kernel launch with events when timing is on */
static int span_begin(LINNEAmdContext *ctx, int kind, hipStream_t st)
{
    if (!ctx->timing) return -1;
    if (ctx->nspans == ctx->span_cap) {
        const int ncap = ctx->span_cap ? ctx->span_cap * 2 : 256;
        hipEvent_t *ne = (hipEvent_t *)realloc(ctx->span_ev, sizeof(hipEvent_t) * 2 * ncap);
        if (!ne) return -1;
        ctx->span_ev = ne;
        int *nk = (int *)realloc(ctx->span_kind, sizeof(int) * ncap);
        if (!nk) return -1;
        ctx->span_kind = nk;
        for (int i = 2 * ctx->span_cap; i < 2 * ncap; i++) if (hipEventCreate(&ctx->span_ev[i]) != hipSuccess) return -1;
        ctx->span_cap = ncap;
    }
    const int id = ctx->nspans++;
    ctx->span_kind[id] = kind;
    (void)hipEventRecord(ctx->span_ev[2 * id], st);
    return id;
}
static void span_end(LINNEAmdContext *ctx, int id, hipStream_t st) { if (id >= 0) (void)hipEventRecord(ctx->span_ev[2 * id + 1], st); }

extern "C" double LINNEAmd_GetLastTimingMs(struct LINNEAmdContext *ctx, int which)
{
    if (!ctx || which < 0) return -1.0;
    if (which == 0) {
        float ms = -1.0f;
        if (!ctx->ev_valid) return -1.0;
        if (hipEventSynchronize(ctx->ev[1]) != hipSuccess || hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) != hipSuccess) return -1.0;
        return ms;
    }
    double sum = 0.0; int cnt = 0;
    for (int i = 0; i < ctx->nspans; i++) if (ctx->span_kind[i] == which) {
        float ms = 0.0f;
        if (hipEventSynchronize(ctx->span_ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms, ctx->span_ev[2 * i], ctx->span_ev[2 * i + 1]) != hipSuccess) return -1.0;
        sum += ms; cnt++;
    }
    return cnt ? sum : -1.0;
}
extern "C" int LINNEAmd_GetLastTimingLaunches(struct LINNEAmdContext *ctx, int which)
{
    if (!ctx) return 0;
    int cnt = 0;
    for (int i = 0; i < ctx->nspans; i++) if (ctx->span_kind[i] == which) cnt++;
    return cnt;
}

static int ensure_buf(LINNEAmdContext *ctx, void **ptr, uint64_t *cap, uint64_t need)
{
    if (*cap >= need) return LNN_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (*ptr) HIPCHK(ctx, hipFree(*ptr));
    *ptr = NULL; *cap = 0;
    HIPCHK(ctx, hipMalloc(ptr, need));
    *cap = need;
    return LNN_OK;
}

struct HostShape { uint32_t L, R, P[LNN_MAXL], coef_off[LNN_MAXL], maxP; double regs[LNN_MAXR]; };
static int shape_info(const struct LINNEAmdShape *s, HostShape *h)
{
    if (!s || s->preset >= 8 || s->num_channels == 0 || s->num_channels > LNN_MAXCH || s->bits_per_sample == 0 || s->bits_per_sample > 32
            || s->num_samples_per_block == 0 || s->ch_process_method > 1 || (s->ch_process_method == 1 && s->num_channels < 2)) return LNN_INVALID_FORMAT;
    lnn_preset_info(s->preset, &h->L, h->P, &h->R, h->regs);
    uint32_t off = 0; h->maxP = 0;
    for (uint32_t l = 0; l < h->L; l++) { h->coef_off[l] = off; off += h->P[l]; if (h->P[l] > h->maxP) h->maxP = h->P[l]; }
    for (uint32_t l = 0; l < h->L; l++) if (s->num_samples_per_block <= h->P[l]) return LNN_INVALID_FORMAT;   /* linne_encoder.c:176-181 */
    return LNN_OK;
}

/* next buffer of the pinned metadata ring, holding at least 2 * F words; waits for the copy that last read it */
static int meta_acquire(LINNEAmdContext *ctx, uint32_t F, int *m_out)
{
    const int m = ctx->meta_next;
    ctx->meta_next = (m + 1) % LNN_META;
    if (ctx->meta_used[m]) { HIPCHK(ctx, hipEventSynchronize(ctx->meta_ev[m])); ctx->meta_used[m] = 0; }
    if (!ctx->meta_ev[m]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->meta_ev[m], hipEventDisableTiming));
    if (ctx->meta_cap[m] < 2ull * F) {
        const uint64_t cap = 2ull * (F < 4096u ? 4096u : F);
        if (ctx->meta_h[m]) { HIPCHK(ctx, hipHostFree(ctx->meta_h[m])); ctx->meta_h[m] = NULL; ctx->meta_cap[m] = 0; }
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->meta_h[m], sizeof(uint32_t) * cap, hipHostMallocDefault));
        ctx->meta_cap[m] = cap;
    }
    *m_out = m;
    return LNN_OK;
}

/* Builds the per-length classes of a batch (tables are host libm values).  The tables stay resident and
 * are uploaded again only when the shape or the set of frame lengths changes; the per-frame class index and length go
 * through a pinned ring, so a call with resident tables enqueues without synchronising the host. */
static int build_classes(LINNEAmdContext *ctx, const struct LINNEAmdShape *shape, const HostShape *hs,
        const uint32_t *h_num_samples, uint32_t F, int for_encode)
{
    DevClass cls[LNN_MAXCLS];
    uint32_t ncls = 0;
    const uint32_t S = shape->num_samples_per_block;
    int m;
    { const int r_ = meta_acquire(ctx, F, &m); if (r_ != LNN_OK) return r_; }
    uint32_t *idx = ctx->meta_h[m], *nsm = ctx->meta_h[m] + F;
    memset(cls, 0, sizeof(cls));
    ctx->na_max = 0;
    uint64_t sin_total = 0, wt_total = 0;
    for (uint32_t f = 0; f < F; f++) {
        const uint32_t n = h_num_samples ? h_num_samples[f] : S;
        if (n == 0 || n > S) { snprintf(ctx->err, sizeof(ctx->err), "frame %u: num_samples %u out of range", f, n); return LNN_INVALID_ARGUMENT; }
        nsm[f] = n;
        uint32_t k = 0;
        for (; k < ncls; k++) if (cls[k].n == n) break;
        if (k == ncls) {
            if (ncls == LNN_MAXCLS) { snprintf(ctx->err, sizeof(ctx->err), "more than %d distinct frame lengths in one batch", LNN_MAXCLS); return LNN_INVALID_ARGUMENT; }
            DevClass &c = cls[ncls++];
            c.n = n;
            uint32_t na = ((n + 7u) / 8u) * 8u;             /* linne_encoder.c:652-654 */
            if (na < hs->maxP) na = hs->maxP;
            if (na > S) na = S;
            c.na = na;
            c.sin_off = (uint32_t)sin_total; sin_total += n;
            if (for_encode) {
                if (na & 1u) { snprintf(ctx->err, sizeof(ctx->err), "odd analysis length %u (odd num_samples_per_block) is not supported by the device path", na); return LNN_INVALID_FORMAT; }
                for (uint32_t l = 0; l < hs->L; l++) {
                    const uint32_t maxu = hs->P[l] < 128u ? hs->P[l] : 128u;    /* linne_network.c:586,594 */
                    uint32_t nt = 0;
                    for (uint32_t u = 1; u <= maxu; u <<= 1) {
                        if ((hs->P[l] % u) != 0 || (na % u) != 0) continue;      /* linne_network.c:291-294 */
                        c.trial_u[l][nt] = u;
                        c.trial_div[l][nt] = 4.0 * pow((double)(na / u - 1u), -2.0);   /* lpc.c:199 */
                        c.wt_off[l][nt] = (uint32_t)wt_total;
                        { const uint32_t pu = hs->P[l] / u; wt_total += na / u + (pu > 4 ? pu : 4); wt_total = (wt_total + 3u) & ~(uint64_t)3u; }   /* tables start 32-byte aligned */
                        nt++;
                    }
                    c.ntrials[l] = nt;
                }
            }
        }
        if (cls[k].na > ctx->na_max) ctx->na_max = cls[k].na;
        idx[f] = k;
    }
    int ret;
    if ((ret = ensure_buf(ctx, (void **)&ctx->d_clsidx, &ctx->clsidx_cap, sizeof(uint32_t) * (uint64_t)(F ? F : 1))) != LNN_OK) return ret;
    if ((ret = ensure_buf(ctx, (void **)&ctx->d_nsmp, &ctx->nsmp_cap, sizeof(uint32_t) * (uint64_t)(F ? F : 1))) != LNN_OK) return ret;
    const bool resident = ctx->sig_valid && ctx->sig_for_encode == for_encode && memcmp(&ctx->sig_shape, shape, sizeof(*shape)) == 0
            && memcmp(ctx->sig_cls, cls, sizeof(cls)) == 0;
    if (!resident) {
        hipError_t e = hipSuccess;
        double *tab = NULL, *wt = NULL;
        ctx->sig_valid = 0;
        if (for_encode) {
            tab = (double *)malloc(sizeof(double) * (sin_total ? sin_total : 1));
            wt = (double *)calloc(wt_total ? wt_total : 1, sizeof(double));
            if (!tab || !wt) { free(tab); free(wt); snprintf(ctx->err, sizeof(ctx->err), "out of host memory"); return LNN_NG; }
            for (uint32_t k = 0; k < ncls; k++) {
                const uint32_t n = cls[k].n;
                for (uint32_t s = 0; s < n; s++) tab[cls[k].sin_off + s] = sin((3.1415926535897932384626433832795029 * s) / (n - 1));   /* lpc.c:192 */
                /* Welch weights per trial over one padded unit (lpc.c:199-204): w[loc] = (div * h) * (n-1-h), h = min(loc, n-1-loc);
                 * zero in the zero zone; the (never written) middle of an odd unit is handled on the device (Q1) */
                for (uint32_t l = 0; l < hs->L; l++)
                    for (uint32_t t = 0; t < cls[k].ntrials[l]; t++) {
                        const uint32_t u = cls[k].trial_u[l][t], nu = cls[k].na / u;
                        const double div = cls[k].trial_div[l][t];
                        double *w = wt + cls[k].wt_off[l][t];
                        for (uint32_t loc = 0; loc < nu; loc++) {
                            const uint32_t h = (loc < (nu >> 1)) ? loc : (nu - 1 - loc);
                            w[loc] = div * (double)h * (double)(nu - 1 - h);
                        }
                    }
            }
            ret = ensure_buf(ctx, (void **)&ctx->d_sin, &ctx->sin_cap, sizeof(double) * (sin_total ? sin_total : 1));
            if (ret == LNN_OK) ret = ensure_buf(ctx, (void **)&ctx->d_wt, &ctx->wt_cap, sizeof(double) * (wt_total ? wt_total : 1));
            if (ret != LNN_OK) { free(tab); free(wt); return ret; }
            e = hipMemcpyAsync(ctx->d_sin, tab, sizeof(double) * sin_total, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_wt, wt, sizeof(double) * wt_total, hipMemcpyHostToDevice, ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_cls, cls, sizeof(DevClass) * LNN_MAXCLS, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          /* cls is on the stack, tab/wt are freed here */
        free(tab); free(wt);
        if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "class table upload: %s", hipGetErrorString(e)); return LNN_NG; }
        memcpy(ctx->sig_cls, cls, sizeof(cls)); ctx->sig_shape = *shape; ctx->sig_for_encode = for_encode; ctx->sig_valid = 1;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_clsidx, idx, sizeof(uint32_t) * F, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_nsmp, nsm, sizeof(uint32_t) * F, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->meta_ev[m], ctx->stream));
    ctx->meta_used[m] = 1;
    return LNN_OK;
}

static uint64_t align_up(uint64_t v) { return (v + 255u) & ~(uint64_t)255u; }

/* bytes of scratch one frame needs (C channel-frames, R passes each) */
static uint64_t frame_scratch_bytes(const struct LINNEAmdShape *shape, const HostShape *hs)
{
    const uint64_t C = shape->num_channels, S = shape->num_samples_per_block, J = C * hs->R;
    uint64_t b = 0;
    b += 2 * C * S * sizeof(int32_t);
    b += J * 2 * S * sizeof(double);
    b += J * LNN_MAXT * LNN_ACW * sizeof(double);
    b += J * LNN_MAXT * LNN_MAXP * sizeof(double);
    b += J * LNN_MAXT * LNN_MAXU * (sizeof(double) + 1);
    b += J * LNN_MAXT * sizeof(double) * (1 + (uint64_t)((S + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64)) + J;
    b += J * LNN_MAXL * LNN_MAXP * sizeof(double);
    b += J * LNN_MAXL * sizeof(uint32_t);
    b += J * 2 * sizeof(double);
    return b + 4096;
}

extern "C" int LINNEAmd_EncodeFramesDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_pcm, const uint32_t *h_num_samples, uint32_t num_frames,
        int32_t *d_residual, int32_t *d_params, double *d_stats)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_pcm || !d_residual || !d_params || !d_stats) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HostShape hs;
    int ret = shape_info(shape, &hs);
    if (ret != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((ret = build_classes(ctx, shape, &hs, h_num_samples, num_frames, 1)) != LNN_OK) return ret;

    const uint32_t C = shape->num_channels, S = shape->num_samples_per_block;
    const uint64_t per_frame = frame_scratch_bytes(shape, &hs);
    if (ctx->arena_bytes < per_frame * 4 + 65536) {       /* grow the arena to hold at least a few frames */
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(ctx->arena)); ctx->arena = NULL; ctx->arena_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->arena, per_frame * 4 + 65536));
        ctx->arena_bytes = per_frame * 4 + 65536;
    }
    /* frame groups ("chunks") rotate over nsub streams, each with its own slice of the arena */
    uint32_t nsub = ctx->nsub > 0 ? (uint32_t)ctx->nsub : 1u;
    while (nsub > 1 && ((ctx->arena_bytes - 65536) / nsub < per_frame * 2 || num_frames < nsub * 512u)) nsub--;
    const uint64_t part_bytes = ((ctx->arena_bytes - 65536) / nsub) & ~(uint64_t)255;
    uint64_t chunk = part_bytes / per_frame;
    if (chunk == 0) chunk = 1;
    {   /* even split over the streams; every kernel carries the job index in grid.x: J = chunk * C * R is kept below 2^22 */
        const uint64_t even = (num_frames + nsub - 1) / nsub;
        if (chunk > even) chunk = even;
        const uint64_t lim = 4194304u / ((uint64_t)C * hs.R);
        if (chunk > lim) chunk = lim;
    }
    ctx->nspans = 0;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ucount, 0, sizeof(uint32_t), ctx->stream));
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream)); }
    const bool use_sub = ctx->nsub > 0;
    if (use_sub) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        for (uint32_t i = 0; i < nsub; i++) HIPCHK(ctx, hipStreamWaitEvent(ctx->sub[i], ctx->ev_start, 0));
    }
    {   /* statistics of every frame of the call: one launch beside the analysis */
        Plan ps; memset(&ps, 0, sizeof(ps));
        ps.C = C; ps.S = S; ps.bits = shape->bits_per_sample; ps.L = hs.L; ps.R = hs.R; ps.F = num_frames;
        for (uint32_t l = 0; l < hs.L; l++) ps.P[l] = hs.P[l];
        ps.scale = ldexp(1.0, -(int)(shape->bits_per_sample - 1));
        ps.pcm = d_pcm; ps.stats = d_stats; ps.cls_of_frame = ctx->d_clsidx; ps.cls = ctx->d_cls; ps.sintab = ctx->d_sin;
        hipStream_t ss = ctx->stream;
        if (ctx->has_side && use_sub) { ss = ctx->side; HIPCHK(ctx, hipStreamWaitEvent(ss, ctx->ev_start, 0)); }
        const int sp_ = span_begin(ctx, 13, ss); hipLaunchKernelGGL(k_stats, dim3(num_frames), dim3(64), 0, ss, ps); span_end(ctx, sp_, ss);
        if (ss != ctx->stream) HIPCHK(ctx, hipEventRecord(ctx->side_done, ss));
    }
    uint32_t chunk_index = 0;
    for (uint32_t f0 = 0; f0 < num_frames; f0 += (uint32_t)chunk, chunk_index++) {
        const uint32_t slot = chunk_index % nsub;
        hipStream_t st = use_sub ? ctx->sub[slot] : ctx->stream;
        const uint32_t Fc = (num_frames - f0 < chunk) ? (num_frames - f0) : (uint32_t)chunk;
        const uint64_t CF = (uint64_t)Fc * C, J = CF * hs.R;
        Plan p; memset(&p, 0, sizeof(p));
        p.C = C; p.S = S; p.bits = shape->bits_per_sample; p.L = hs.L; p.R = hs.R; p.ms = shape->ch_process_method; p.F = Fc; p.J = (uint32_t)J;
        for (uint32_t l = 0; l < hs.L; l++) { p.P[l] = hs.P[l]; p.coef_off[l] = hs.coef_off[l]; }
        for (uint32_t r = 0; r < hs.R; r++) p.regs[r] = hs.regs[r];
        p.scale = ldexp(1.0, -(int)(shape->bits_per_sample - 1));
        p.pcm = d_pcm + (size_t)f0 * C * S; p.resid = d_residual + (size_t)f0 * C * S;
        p.prm = d_params + (size_t)f0 * C * LINNE_AMD_PARAM_WORDS; p.stats = d_stats + (size_t)f0 * C * LINNE_AMD_STAT_WORDS;
        p.cls_of_frame = ctx->d_clsidx + f0; p.cls = ctx->d_cls; p.sintab = ctx->d_sin; p.wtab = ctx->d_wt; p.ucount = ctx->d_ucount;
        uint8_t *const abase = (uint8_t *)ctx->arena + (size_t)slot * part_bytes;
        uint8_t *a = abase;
#define TAKE(ptr, type, count) do { ptr = (type *)a; a += align_up(sizeof(type) * (uint64_t)(count)); } while (0)
        TAKE(p.xint, int32_t, CF * S); TAKE(p.xtmp, int32_t, CF * S);
        TAKE(p.sig, double, J * 2 * S);
        TAKE(p.acorr, double, J * LNN_MAXT * LNN_ACW); TAKE(p.tcoef, double, J * LNN_MAXT * LNN_MAXP);
        TAKE(p.ptail, double, J * LNN_MAXT * LNN_MAXU); TAKE(p.ptail_set, uint8_t, J * LNN_MAXT * LNN_MAXU);
        TAKE(p.tloss, double, J * LNN_MAXT); p.npart = ((S + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64); TAKE(p.tsum, double, J * LNN_MAXT * p.npart); TAKE(p.uncertain, uint8_t, J); TAKE(p.lparams, double, J * LNN_MAXL * LNN_MAXP);
        TAKE(p.lunits, uint32_t, J * LNN_MAXL); TAKE(p.jloss, double, J); TAKE(p.jtail, double, J);
#undef TAKE
        if ((uint64_t)(a - abase) > part_bytes) { snprintf(ctx->err, sizeof(ctx->err), "internal: arena overflow"); return LNN_NG; }
        const uint32_t sblocks = (S + 255) / 256;
        { const int sp_ = span_begin(ctx, 1, st); hipLaunchKernelGGL(k_prep, dim3(Fc, C), dim3(PREP_THREADS), 0, st, p); span_end(ctx, sp_, st); }
        uint32_t cur = 0;
        for (uint32_t l = 0; l < hs.L; l++) {
            const uint32_t maxu = hs.P[l] < 128u ? hs.P[l] : 128u;
            uint32_t nt = 0, nprob = 0, nchain = 0;
            for (uint32_t u = 1; u <= maxu; u <<= 1) { nt++; nprob += u; nchain += hs.P[l] + u; }
            {
                const int sp_ = span_begin(ctx, (hs.P[l] >= 32u) ? 3 : 14, st); dispatch_autocorr2(st, p, l, cur, ctx->na_max); span_end(ctx, sp_, st);
            }
            { const int sp_ = span_begin(ctx, 4, st);
              uint32_t nbig = 0; for (uint32_t u = 1; u <= maxu && hs.P[l] / u >= 16u; u <<= 1) nbig += u;
              (void)nbig;
              for (uint32_t t = 0, u = 1; u <= maxu && hs.P[l] / u >= LEV_WAVE_MIN_ORDER; u <<= 1, t++) {
                  const uint32_t np = hs.P[l] / u;
                  const size_t lds = sizeof(double) * 64 * (size_t)(2 * np + 3);
                  hipLaunchKernelGGL(k_levinson_lds, dim3(((uint32_t)J + 63) / 64, u), dim3(64), lds, st, p, l, t);
              }
              hipLaunchKernelGGL(k_levinson, dim3(((uint32_t)J + 63) / 64, nprob), dim3(64), 0, st, p, l); span_end(ctx, sp_, st); }
            { const int sp_ = span_begin(ctx, (l == 0) ? 15 : 5, st); if (l == 0) hipLaunchKernelGGL((k_fir2<2, true>), dim3((uint32_t)J, (S + FIR_TILE - 1) / FIR_TILE), dim3(FIR_THREADS), 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir2<2, false>), dim3((uint32_t)J, (S + FIR_TILE - 1) / FIR_TILE), dim3(FIR_THREADS), 0, st, p, l, cur); span_end(ctx, sp_, st); }
            { const int sp_ = span_begin(ctx, 7, st); hipLaunchKernelGGL(k_select, dim3(((uint32_t)J + 63) / 64), dim3(64), 0, st, p, l, 0u); span_end(ctx, sp_, st); }
            /* exact ordered chains for the (rare) jobs the certified search flagged; everything else exits at once */
            { const int sp_ = span_begin(ctx, 6, st); if (l == 0) hipLaunchKernelGGL((k_fir2<0, true>), dim3((uint32_t)J, 1), dim3(FIR_THREADS), 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir2<0, false>), dim3((uint32_t)J, 1), dim3(FIR_THREADS), 0, st, p, l, cur);
              hipLaunchKernelGGL(k_select, dim3(((uint32_t)J + 63) / 64), dim3(64), 0, st, p, l, 1u); span_end(ctx, sp_, st); }
            { const int sp_ = span_begin(ctx, (l == 0) ? 16 : 8, st); if (l == 0) hipLaunchKernelGGL((k_fir2<1, true>), dim3((uint32_t)J, (S + FIR_TILE - 1) / FIR_TILE), dim3(FIR_THREADS), 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir2<1, false>), dim3((uint32_t)J, (S + FIR_TILE - 1) / FIR_TILE), dim3(FIR_THREADS), 0, st, p, l, cur); span_end(ctx, sp_, st); }
            cur ^= 1u;
        }
        { const int sp_ = span_begin(ctx, 9, st); hipLaunchKernelGGL(k_chain_sum<1>, dim3(((uint32_t)J + 63) / 64), dim3(SUM_THREADS), 0, st, p, 0u, cur); span_end(ctx, sp_, st); }
        { const int sp_ = span_begin(ctx, 10, st); hipLaunchKernelGGL(k_finalize, dim3((uint32_t)CF), dim3(FIN_THREADS), 0, st, p); span_end(ctx, sp_, st); }
        HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->has_side && use_sub) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
    if (use_sub) {
        for (uint32_t i = 0; i < nsub; i++) { HIPCHK(ctx, hipEventRecord(ctx->sub_done[i], ctx->sub[i])); HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->sub_done[i], 0)); }
    }
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream)); ctx->ev_valid = 1; }
    return LNN_OK;
}

extern "C" int LINNEAmd_DecodeFramesDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        int32_t *d_data, const uint32_t *h_num_samples, uint32_t num_frames, const int32_t *d_params)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_data || !d_params) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HostShape hs;
    int ret = shape_info(shape, &hs);
    if (ret != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((ret = build_classes(ctx, shape, &hs, h_num_samples, num_frames, 0)) != LNN_OK) return ret;
    DecPlan p; memset(&p, 0, sizeof(p));
    p.C = shape->num_channels; p.S = shape->num_samples_per_block; p.L = hs.L; p.ms = shape->ch_process_method; p.F = num_frames;
    for (uint32_t l = 0; l < hs.L; l++) { p.P[l] = hs.P[l]; p.coef_off[l] = hs.coef_off[l]; }
    p.data = d_data; p.prm = d_params; p.nsmp = ctx->d_nsmp;
    ctx->nspans = 0;
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream)); }
    {   /* layers in reverse order (linne_decoder.c:503-509): long layers one wave per channel-frame, short ones (order <= 16)
         * with lanes = channel-frames; the de-emphasis rides on layer 0's pass */
        const int sp_ = span_begin(ctx, 11, ctx->stream);
        const uint32_t CF = num_frames * p.C, gsmall = (CF + 63) / 64;
        if (hs.P[0] > 16) { snprintf(ctx->err, sizeof(ctx->err), "internal: layer 0 of order %u", hs.P[0]); return LNN_NG; }
        for (int32_t l = (int32_t)hs.L - 1; l >= 0; l--) {
            const bool de = (l == 0);
            switch (hs.P[l]) {
            case 2:  if (de) hipLaunchKernelGGL((k_synth_small<2, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<2, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 4:  if (de) hipLaunchKernelGGL((k_synth_small<4, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<4, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 8:  if (de) hipLaunchKernelGGL((k_synth_small<8, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<8, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 16: if (de) hipLaunchKernelGGL((k_synth_small<16, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<16, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 32:  hipLaunchKernelGGL((k_synth_big<32>), dim3((CF + 15) / 16), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 64:  hipLaunchKernelGGL((k_synth_big<64>), dim3((CF + 15) / 16), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 128: hipLaunchKernelGGL((k_synth_big<128>), dim3((CF + 15) / 16), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            default: hipLaunchKernelGGL(k_synthesize, dim3(CF), dim3(64), 0, ctx->stream, p, (uint32_t)l, 0u); break;      /* not a preset size */
            }
        }
        span_end(ctx, sp_, ctx->stream);
    }
    if (p.ms)
        { const int sp_ = span_begin(ctx, 12, ctx->stream); hipLaunchKernelGGL(k_ms_to_lr, dim3(num_frames, (p.S + 255) / 256), dim3(256), 0, ctx->stream, p); span_end(ctx, sp_, ctx->stream); }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream)); ctx->ev_valid = 1; }
    return LNN_OK;
}

/* host-buffer forms: staging buffers are allocated per call (the block-at-a-time API is latency-, not
 * throughput-oriented; batch callers use the device entry points) */
extern "C" int LINNEAmd_EncodeFramesHost(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *pcm, const uint32_t *num_samples, uint32_t num_frames,
        int32_t *residual, int32_t *params, double *stats)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    if (!shape || !pcm || !residual || !params || !stats) return LNN_INVALID_ARGUMENT;
    if (num_frames == 0) return LNN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t CS = (uint64_t)shape->num_channels * shape->num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (uint64_t)shape->num_channels * num_frames,
                   sb = sizeof(double) * LINNE_AMD_STAT_WORDS * (uint64_t)shape->num_channels * num_frames;
    int32_t *d_pcm = NULL, *d_res = NULL, *d_prm = NULL; double *d_st = NULL;
    int ret = LNN_NG;
    if (hipMalloc((void **)&d_pcm, nb) != hipSuccess || hipMalloc((void **)&d_res, nb) != hipSuccess
            || hipMalloc((void **)&d_prm, pb) != hipSuccess || hipMalloc((void **)&d_st, sb) != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "hipMalloc of staging buffers failed");
        goto done;
    }
    if (hipMemcpyAsync(d_pcm, pcm, nb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "H2D failed"); goto done; }
    if (hipMemsetAsync(d_prm, 0, pb, ctx->stream) != hipSuccess || hipMemsetAsync(d_st, 0, sb, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "memset failed"); goto done; }
    ret = LINNEAmd_EncodeFramesDevice(ctx, shape, d_pcm, num_samples, num_frames, d_res, d_prm, d_st);
    if (ret != LNN_OK) goto done;
    ret = LNN_NG;
    if (hipMemcpyAsync(residual, d_res, nb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess
            || hipMemcpyAsync(params, d_prm, pb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess
            || hipMemcpyAsync(stats, d_st, sb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "D2H failed"); goto done; }
    {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "stream sync: %s", hipGetErrorString(e)); goto done; }
    }
    ret = LNN_OK;
done:
    hipStreamSynchronize(ctx->stream);
    if (d_pcm) hipFree(d_pcm);
    if (d_res) hipFree(d_res);
    if (d_prm) hipFree(d_prm);
    if (d_st) hipFree(d_st);
    return ret;
}

extern "C" int LINNEAmd_DecodeFramesHost(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        int32_t *data, const uint32_t *num_samples, uint32_t num_frames, const int32_t *params)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    if (!shape || !data || !params) return LNN_INVALID_ARGUMENT;
    if (num_frames == 0) return LNN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t CS = (uint64_t)shape->num_channels * shape->num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (uint64_t)shape->num_channels * num_frames;
    int32_t *d_data = NULL, *d_prm = NULL;
    int ret = LNN_NG;
    if (hipMalloc((void **)&d_data, nb) != hipSuccess || hipMalloc((void **)&d_prm, pb) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "hipMalloc of staging buffers failed"); goto done; }
    if (hipMemcpyAsync(d_data, data, nb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess
            || hipMemcpyAsync(d_prm, params, pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "H2D failed"); goto done; }
    ret = LINNEAmd_DecodeFramesDevice(ctx, shape, d_data, num_samples, num_frames, d_prm);
    if (ret != LNN_OK) goto done;
    ret = LNN_NG;
    if (hipMemcpyAsync(data, d_data, nb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "D2H failed"); goto done; }
    {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "stream sync: %s", hipGetErrorString(e)); goto done; }
    }
    ret = LNN_OK;
done:
    hipStreamSynchronize(ctx->stream);
    if (d_data) hipFree(d_data);
    if (d_prm) hipFree(d_prm);
    return ret;
}


extern "C" int LINNEAmd_RicePlanDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_residual, const uint32_t *h_num_samples, uint32_t num_frames, uint8_t *d_plan)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_residual || !d_plan) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HostShape hs;
    int ret = shape_info(shape, &hs);
    if (ret != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->rice_nsteps == 0) ctx->rice_nsteps = lnn_rice_k2_steps(ctx->rice_steps);
    int m;
    if ((ret = meta_acquire(ctx, num_frames, &m)) != LNN_OK) return ret;
    uint32_t *nsm = ctx->meta_h[m];
    for (uint32_t f = 0; f < num_frames; f++) {
        const uint32_t n = h_num_samples ? h_num_samples[f] : shape->num_samples_per_block;
        if (n == 0 || n > shape->num_samples_per_block) { snprintf(ctx->err, sizeof(ctx->err), "frame %u: num_samples %u out of range", f, n); return LNN_INVALID_ARGUMENT; }
        nsm[f] = n;
    }
    if ((ret = ensure_buf(ctx, (void **)&ctx->d_plan_nsmp, &ctx->plan_nsmp_cap, sizeof(uint32_t) * (uint64_t)num_frames)) != LNN_OK) return ret;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_plan_nsmp, nsm, sizeof(uint32_t) * num_frames, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->meta_ev[m], ctx->stream));
    ctx->meta_used[m] = 1;
    RicePlanArgs a; memset(&a, 0, sizeof(a));
    a.resid = d_residual; a.nsmp = ctx->d_plan_nsmp; a.plan = d_plan; a.C = shape->num_channels; a.S = shape->num_samples_per_block;
    a.nsteps = ctx->rice_nsteps;
    for (uint32_t i = 0; i < 32; i++) a.steps[i] = ctx->rice_steps[i];
    const uint64_t CF = (uint64_t)num_frames * shape->num_channels;
    for (uint64_t c0 = 0; c0 < CF; ) {        /* grid.x limit: split very large batches on frame boundaries */
        uint64_t cnt = CF - c0;
        const uint64_t lim = (0x7FFFFFFFull / a.C) * a.C;
        if (cnt > lim) cnt = lim;
        RicePlanArgs b = a;
        b.resid = d_residual + c0 * a.S; b.plan = d_plan + c0 * LINNE_AMD_RICE_PLAN_BYTES; b.nsmp = ctx->d_plan_nsmp + c0 / a.C;
        const int sp_ = span_begin(ctx, 17, ctx->stream);
        hipLaunchKernelGGL(k_rice_plan, dim3((uint32_t)cnt), dim3(RICE_THREADS), 0, ctx->stream, b);
        span_end(ctx, sp_, ctx->stream);
        c0 += cnt;
    }
    HIPCHK(ctx, hipGetLastError());
    return LNN_OK;
}

/* ================================================================================================
 * staging slots: pinned host buffers + device buffers for a group of frames.  Submit enqueues H2D (copy-in
 * stream), the kernels (context stream) and D2H (copy-out stream) chained by events and returns at once, so a
 * caller that rotates over a few slots overlaps its own host work (entropy stage), PCIe and the kernels.
 * ============================================================================================== */
struct LINNEAmdSlot {
    LINNEAmdContext *ctx; struct LINNEAmdShape shape; uint32_t max_frames; int for_encode;
    int32_t *h_pcm, *h_data, *h_prm; double *h_st; uint8_t *h_plan;
    int32_t *d_pcm, *d_data, *d_prm; double *d_st; uint8_t *d_plan;
    hipEvent_t ev_in, ev_k, ev_done; int pending;
};

static int ctx_copy_streams(LINNEAmdContext *ctx)
{
    if (ctx->has_copy) return LNN_OK;
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
    ctx->has_copy = 1;
    return LNN_OK;
}

extern "C" void LINNEAmd_SlotDestroy(struct LINNEAmdSlot *s)
{
    if (!s) return;
    hipSetDevice(s->ctx->device);
    if (s->pending) hipEventSynchronize(s->ev_done);
    if (s->h_pcm) hipHostFree(s->h_pcm);
    if (s->h_data) hipHostFree(s->h_data);
    if (s->h_prm) hipHostFree(s->h_prm);
    if (s->h_st) hipHostFree(s->h_st);
    if (s->h_plan) hipHostFree(s->h_plan);
    if (s->d_plan) hipFree(s->d_plan);
    if (s->d_pcm) hipFree(s->d_pcm);
    if (s->d_data) hipFree(s->d_data);
    if (s->d_prm) hipFree(s->d_prm);
    if (s->d_st) hipFree(s->d_st);
    if (s->ev_in) hipEventDestroy(s->ev_in);
    if (s->ev_k) hipEventDestroy(s->ev_k);
    if (s->ev_done) hipEventDestroy(s->ev_done);
    free(s);
}

extern "C" struct LINNEAmdSlot *LINNEAmd_SlotCreate(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        uint32_t max_frames, int for_encode)
{
    HostShape hs;
    if (!ctx) return NULL;
    ctx->err[0] = 0;
    if (!shape || max_frames == 0 || shape_info(shape, &hs) != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "SlotCreate: invalid shape"); return NULL; }
    if (hipSetDevice(ctx->device) != hipSuccess || ctx_copy_streams(ctx) != LNN_OK) return NULL;
    LINNEAmdSlot *s = (LINNEAmdSlot *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->ctx = ctx; s->shape = *shape; s->max_frames = max_frames; s->for_encode = for_encode;
    const uint64_t CS = (uint64_t)shape->num_channels * shape->num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * max_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (uint64_t)shape->num_channels * max_frames,
                   sb = sizeof(double) * LINNE_AMD_STAT_WORDS * (uint64_t)shape->num_channels * max_frames;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_data, nb, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_prm, pb, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_data, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_prm, pb);
    if (for_encode) {
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_pcm, nb, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_st, sb, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_pcm, nb);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_st, sb);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_plan, (uint64_t)LINNE_AMD_RICE_PLAN_BYTES * shape->num_channels * max_frames, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_plan, (uint64_t)LINNE_AMD_RICE_PLAN_BYTES * shape->num_channels * max_frames);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_k, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "SlotCreate: %s", hipGetErrorString(e)); LINNEAmd_SlotDestroy(s); return NULL; }
    return s;
}

extern "C" int32_t *LINNEAmd_SlotPcm(struct LINNEAmdSlot *s) { return s ? s->h_pcm : NULL; }
extern "C" int32_t *LINNEAmd_SlotData(struct LINNEAmdSlot *s) { return s ? s->h_data : NULL; }
extern "C" int32_t *LINNEAmd_SlotParams(struct LINNEAmdSlot *s) { return s ? s->h_prm : NULL; }
extern "C" double *LINNEAmd_SlotStats(struct LINNEAmdSlot *s) { return s ? s->h_st : NULL; }
extern "C" uint8_t *LINNEAmd_SlotRicePlan(struct LINNEAmdSlot *s) { return s ? s->h_plan : NULL; }
extern "C" uint32_t LINNEAmd_SlotCapacity(const struct LINNEAmdSlot *s) { return s ? s->max_frames : 0; }

extern "C" int LINNEAmd_SlotWait(struct LINNEAmdSlot *s)
{
    if (!s) return LNN_INVALID_ARGUMENT;
    if (!s->pending) return LNN_OK;
    HIPCHK(s->ctx, hipEventSynchronize(s->ev_done));
    s->pending = 0;
    return LNN_OK;
}

extern "C" int LINNEAmd_SlotEncodeSubmit(struct LINNEAmdSlot *s, const uint32_t *num_samples, uint32_t num_frames)
{
    if (!s || !s->for_encode) return LNN_INVALID_ARGUMENT;
    LINNEAmdContext *ctx = s->ctx;
    if (num_frames == 0 || num_frames > s->max_frames) { snprintf(ctx->err, sizeof(ctx->err), "SlotEncodeSubmit: %u frames in a slot of %u", num_frames, s->max_frames); return LNN_INVALID_ARGUMENT; }
    int ret = LINNEAmd_SlotWait(s);
    if (ret != LNN_OK) return ret;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t C = s->shape.num_channels, CS = C * s->shape.num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * C * num_frames, sb = sizeof(double) * LINNE_AMD_STAT_WORDS * C * num_frames;
    HIPCHK(ctx, hipMemcpyAsync(s->d_pcm, s->h_pcm, nb, hipMemcpyHostToDevice, ctx->copy_in));
    HIPCHK(ctx, hipEventRecord(s->ev_in, ctx->copy_in));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_in, 0));
    HIPCHK(ctx, hipMemsetAsync(s->d_prm, 0, pb, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(s->d_st, 0, sb, ctx->stream));
    if ((ret = LINNEAmd_EncodeFramesDevice(ctx, &s->shape, s->d_pcm, num_samples, num_frames, s->d_data, s->d_prm, s->d_st)) != LNN_OK) return ret;
    if ((ret = LINNEAmd_RicePlanDevice(ctx, &s->shape, s->d_data, num_samples, num_frames, s->d_plan)) != LNN_OK) return ret;
    HIPCHK(ctx, hipEventRecord(s->ev_k, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_out, s->ev_k, 0));
    HIPCHK(ctx, hipMemcpyAsync(s->h_data, s->d_data, nb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipMemcpyAsync(s->h_prm, s->d_prm, pb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipMemcpyAsync(s->h_st, s->d_st, sb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipMemcpyAsync(s->h_plan, s->d_plan, (uint64_t)LINNE_AMD_RICE_PLAN_BYTES * C * num_frames, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipEventRecord(s->ev_done, ctx->copy_out));
    s->pending = 1;
    return LNN_OK;
}

extern "C" int LINNEAmd_SlotDecodeSubmit(struct LINNEAmdSlot *s, const uint32_t *num_samples, uint32_t num_frames)
{
    if (!s) return LNN_INVALID_ARGUMENT;
    LINNEAmdContext *ctx = s->ctx;
    if (num_frames == 0 || num_frames > s->max_frames) { snprintf(ctx->err, sizeof(ctx->err), "SlotDecodeSubmit: %u frames in a slot of %u", num_frames, s->max_frames); return LNN_INVALID_ARGUMENT; }
    int ret = LINNEAmd_SlotWait(s);
    if (ret != LNN_OK) return ret;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t C = s->shape.num_channels, CS = C * s->shape.num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * C * num_frames;
    HIPCHK(ctx, hipMemcpyAsync(s->d_data, s->h_data, nb, hipMemcpyHostToDevice, ctx->copy_in));
    HIPCHK(ctx, hipMemcpyAsync(s->d_prm, s->h_prm, pb, hipMemcpyHostToDevice, ctx->copy_in));
    HIPCHK(ctx, hipEventRecord(s->ev_in, ctx->copy_in));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_in, 0));
    if ((ret = LINNEAmd_DecodeFramesDevice(ctx, &s->shape, s->d_data, num_samples, num_frames, s->d_prm)) != LNN_OK) return ret;
    HIPCHK(ctx, hipEventRecord(s->ev_k, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_out, s->ev_k, 0));
    HIPCHK(ctx, hipMemcpyAsync(s->h_data, s->d_data, nb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipEventRecord(s->ev_done, ctx->copy_out));
    s->pending = 1;
    return LNN_OK;
}

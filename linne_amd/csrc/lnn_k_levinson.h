/* lnn_k_levinson.h -- ridge + Levinson-Durbin kernel (k_levinson_lds).
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_LEVINSON_H_INCLUDED
#define LNN_K_LEVINSON_H_INCLUDED

/* ridge + Levinson-Durbin of every (trial, unit) problem (lpc.c:327-366, 578-633 with zero AF iterations), lanes = jobs: a wavefront solves the SAME (trial, unit) problem of 64
 * consecutive jobs, each lane running the scalar recursion of `levinson` above on its own column of two LDS arrays
 * (a[i][lane], r[i][lane]: conflict-free 8-byte accesses).  Every wave instruction therefore advances 64 problems; the
 * ordered sum a[0]r[k+1] + ... + a[k]r[1] (lpc.c:295-297) is one chain per lane.  grid = (job groups, units of the trial). */
__device__ __forceinline__ void levinson_problem(const Plan &p, uint32_t layer, uint32_t t, uint32_t unit, double *lev_lds, uint32_t lane, uint32_t job0)
{
    uint32_t job = job0 + lane;
    const bool inrange = job < p.J;
    if (!inrange) job = p.J - 1;
    const DevClass &c = job_class(p, job);
    const uint32_t P = p.P[layer], u = 1u << t, np = P >> t, P0 = p.P[0];
    const bool have = inrange && t < c.ntrials[layer];
    const uint32_t n = c.na / u;
    double *sa = lev_lds + lane, *sr = lev_lds + (size_t)(np + 2) * 64 + lane;      /* element i at [i * 64] */
    const double reg = p.job_reg ? p.job_reg[job] : p.regs[job % p.R];
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
                for (; i + 8 <= k + 1; i += 8) {                /* eight terms per trip: sixteen reads in flight, the adds stay in order */
                    const double a0 = pa[0], a1_ = pa[64], a2 = pa[128], a3 = pa[192], a4 = pa[256], a5 = pa[320], a6 = pa[384], a7 = pa[448];
                    const double q0 = pr[0], q1 = *(pr - 64), q2 = *(pr - 128), q3 = *(pr - 192), q4 = *(pr - 256), q5 = *(pr - 320), q6 = *(pr - 384), q7 = *(pr - 448);
                    const double m0 = a0 * q0, m1 = a1_ * q1, m2 = a2 * q2, m3 = a3 * q3, m4 = a4 * q4, m5 = a5 * q5, m6 = a6 * q6, m7 = a7 * q7;
                    gamma += m0; gamma += m1; gamma += m2; gamma += m3; gamma += m4; gamma += m5; gamma += m6; gamma += m7;
                    pa += 512; pr -= 512;
                }
                for (; i + 4 <= k + 1; i += 4) {
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
            while (i + 6 < j) {                              /* four pairs per trip (i+3 < j-3, written without unsigned underflow): eight loads in flight, then the updates */
                double *pi = sa + (size_t)i * 64, *pj = sa + (size_t)j * 64;
                const double ai0 = pi[0], ai1 = pi[64], ai2 = pi[128], ai3 = pi[192];
                const double aj0 = pj[0], aj1 = *(pj - 64), aj2 = *(pj - 128), aj3 = *(pj - 192);
                pi[0] = ai0 + gamma * aj0;   pj[0] = aj0 + gamma * ai0;
                pi[64] = ai1 + gamma * aj1;  *(pj - 64) = aj1 + gamma * ai1;
                pi[128] = ai2 + gamma * aj2; *(pj - 128) = aj2 + gamma * ai2;
                pi[192] = ai3 + gamma * aj3; *(pj - 192) = aj3 + gamma * ai3;
                i += 4; j -= 4;
            }
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
    if (!have) return;                                         /* (lanes of a wave only: the caller's loops are wave-uniform) */
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

/* grid = (job groups, units of trial t).  Wave 0 of a block solves problem (t, blockIdx.y).  A block may carry riding waves
 * (blockDim 64 * (1 + nride), `ride` < LNN_MAXT): together they solve every problem of the trials ride, ride+1, ... of the
 * same 64 jobs, dealt out in turn, one after the other in each wave's own LDS columns behind wave 0's.  The long layer's
 * one-unit trial holds 130 KB of LDS -- one wave per CU, on one of its four SIMDs -- so the many tiny problems of its short
 * trials (latency, not arithmetic) ride along on the other three instead of taking launches of their own. */
#define LEV_MAXRIDE 3
__global__ __launch_bounds__(64 * (1 + LEV_MAXRIDE)) void k_levinson_lds(Plan p, uint32_t layer, uint32_t t, uint32_t ride)
{
    extern __shared__ __attribute__((aligned(16))) double lev_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, job0 = blockIdx.x * 64;
    if (wave == 0) { levinson_problem(p, layer, t, blockIdx.y, lev_lds, lane, job0); return; }
    const uint32_t nride = (blockDim.x >> 6) - 1u;
    double *mine = lev_lds + (size_t)(2 * (p.P[layer] >> t) + 3) * 64 + (size_t)(wave - 1u) * (2 * (p.P[layer] >> ride) + 3) * 64;
    const uint32_t maxu = p.P[layer] < (uint32_t)LNN_MAXU ? p.P[layer] : (uint32_t)LNN_MAXU;
    uint32_t q = 0;
    for (uint32_t tt = ride, u = 1u << ride; u <= maxu; tt++, u <<= 1)
        for (uint32_t unit = 0; unit < u; unit++, q++)
            if (q % nride == wave - 1u) levinson_problem(p, layer, tt, unit, mine, lane, job0);
}

/* The same recursion for a SMALL batch (block-at-a-time calls: a handful of jobs): a WAVE per (job, trial, unit) problem, all
 * problems of a layer in one launch.  What is serial in the recursion -- the ordered sum a[0]r[k+1] + ... + a[k]r[1] -- stays one
 * chain, but its products are formed by the lanes side by side (lane i: a[i] r[k+1-i], into LDS) and so is the update
 * a[i] += gamma a[k+1-i]; the chain then is k + 1 adds fed from LDS.  An order-128 problem takes ~50 us (the lanes = jobs form
 * above, with 8 of its 64 lanes busy: 0.32 ms, and its four launches follow one another).  Every operation is the one
 * levinson_problem performs, on the same operands, in the same order.  grid = (jobs, problems of the layer). */
__global__ __launch_bounds__(64) void k_levinson_wave(Plan p, uint32_t layer)
{
    __shared__ __attribute__((aligned(16))) double sa[2][LNN_MAXP + 2], sr[LNN_MAXP + 2], sprod[LNN_MAXP + 2];
    const uint32_t job = blockIdx.x, lane = threadIdx.x;
    const uint32_t t = 31u - (uint32_t)__clz((int)(blockIdx.y + 1u)), unit = blockIdx.y + 1u - (1u << t);
    const DevClass &c = job_class(p, job);
    if (t >= c.ntrials[layer]) return;
    const uint32_t P = p.P[layer], u = 1u << t, np = P >> t, P0 = p.P[0];
    const uint32_t n = c.na / u;
    const double reg = p.job_reg ? p.job_reg[job] : p.regs[job % p.R];
    const uint32_t ajob = (layer == 0) ? job - job % p.R : job;
    const double *r = p.acorr + ((size_t)ajob * LNN_MAXT + t) * LNN_ACW + (size_t)unit * (np + 1);
    double *h = p.tcoef + ((size_t)job * LNN_MAXT + t) * LNN_MAXP + (size_t)unit * np;
    const bool last = (layer + 1 == p.L);
    for (uint32_t i = lane; i <= np; i += 64u) sr[i] = r[i];
    for (uint32_t i = lane; i < np + 2u; i += 64u) { sa[0][i] = 0.0; sa[1][i] = 0.0; }
    __builtin_amdgcn_wave_barrier();
    double tail = 0.0; int tail_set = 0;
    const double r0 = sr[0] * (1.0 + reg);                    /* lpc.c:358 */
    const bool zero = (n < np) || (fabs(r0) < (double)FLT_EPSILON);      /* lpc.c:349-355, 271-276 */
    uint32_t cur = 0;
    if (!zero) {
        const double r1 = sr[1];
        double ek = r0;
        const double a1 = -r1 / r0;
        ek += r1 * a1;
        if (lane == 0) { sa[0][0] = 1.0; sa[0][1] = a1; }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = 1; k < np; k++) {
            const double *a = sa[cur];
            double *an = sa[cur ^ 1u];
            for (uint32_t i = lane; i <= k; i += 64u) sprod[i] = a[i] * sr[k + 1u - i];
            __builtin_amdgcn_wave_barrier();
            double gamma = 0.0;
            {
                uint32_t i = 0;
                for (; i + 8u <= k + 1u; i += 8u) {
                    const lnn_d2 m0 = *(const lnn_d2 *)(sprod + i), m1 = *(const lnn_d2 *)(sprod + i + 2), m2 = *(const lnn_d2 *)(sprod + i + 4), m3 = *(const lnn_d2 *)(sprod + i + 6);
                    gamma += m0.x; gamma += m0.y; gamma += m1.x; gamma += m1.y; gamma += m2.x; gamma += m2.y; gamma += m3.x; gamma += m3.y;
                }
                for (; i <= k; i++) gamma += sprod[i];
            }
            gamma /= -ek;
            ek *= (1.0 - gamma * gamma);
            for (uint32_t i = lane; i <= k + 1u; i += 64u) {
                double v;
                if (i == 0u) v = 1.0 + gamma * 0.0;                /* u[0]   + gamma*v[0]   */
                else if (i == k + 1u) v = 0.0 + gamma * 1.0;       /* u[k+1] + gamma*v[k+1] */
                else v = a[i] + gamma * a[k + 1u - i];
                an[i] = v;
            }
            __builtin_amdgcn_wave_barrier();
            cur ^= 1u;
            if (last && k == P0) { tail = -gamma; tail_set = 1; }
        }
    }
    if (zero) {
        for (uint32_t k = lane; k < np; k += 64u) h[k] = 0.0;
        tail = 0.0; tail_set = (np >= P0) ? 1 : 0;             /* zero branches write parcor[0..order] */
    } else {
        for (uint32_t k = lane; k < np; k += 64u) h[k] = sa[cur][np - k];
        if (!(last && np > P0)) { tail = 0.0; tail_set = 0; }
    }
    if (last && lane == 0) {
        p.ptail[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = tail;
        p.ptail_set[((size_t)job * LNN_MAXT + t) * LNN_MAXU + unit] = (uint8_t)tail_set;
    }
}

#endif

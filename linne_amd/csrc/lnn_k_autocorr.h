/* lnn_k_autocorr.h -- Welch window + autocorrelation kernels (k_autocorr2, k_autocorr_lane) and their launchers.
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_AUTOCORR_H_INCLUDED
#define LNN_K_AUTOCORR_H_INCLUDED

/* ------------------------------------------------------------------------------------------------
 * analysis, one layer at a time over every job = (channel-frame, regulariser pass)
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ const DevClass &job_class(const Plan &p, uint32_t job) { return p.cls[p.cls_of_frame[(job / p.R) / p.C]]; }
/* Quirk Q1 for an ODD analysis length (an odd num_samples_per_block): only the one-unit trial exists then, its Welch window never
 * writes the middle sample (lpc.c:200-204), no other Welch call of the frame does either, and what the window buffer holds there
 * is what the block-type estimate left (linne_encoder.c:494-503 -> lpc.c:188-195): the LAST channel's raw sample at that index
 * under the SIN window -- a function of the frame alone */
__device__ __forceinline__ double q1_stale_first_trial(const Plan &p, uint32_t job, const DevClass &c)
{
    const uint32_t f = (job / p.R) / p.C, mid = c.na >> 1;
    if (mid >= c.n) return 0.0;
    const size_t i = ((size_t)p.frame_map[f] * p.C + (p.C - 1u)) * p.S + mid;
    return ((double)pcm_at(p, i) * p.scale) * p.sintab[c.sin_off + mid];
}

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
                if (t < c.ntrials[layer] && !hist_takes(p, layer, c)) {
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
                if (t < c.ntrials[layer] && !hist_takes(p, layer, c)) {
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
                    if ((g_n & 1u) && g_u == 1u) g_stale = q1_stale_first_trial(p, job, c);
                    else if (g_n & 1u) {     /* Q1: stale middle sample = previous trial's last unit at local index m */
                        const uint32_t m = g_n >> 1, n2 = 2 * g_n, si = (g_u / 2 - 1) * n2 + m;
                        const double xv = L0 ? ((double)g_xi[si] * p.scale) : g_xd[si];
                        const double wgt = c.trial_div[layer][t - 1] * (double)m * (double)(n2 - 1 - m);
                        g_stale = xv * wgt;
                    }
                }
            }
        }
    }

    if (__all(!active && !gen)) return;             /* every job of the wave is k_autocorr_hist's (or beyond the batch) */

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
template <int K, bool L0, int D = 1>          /* D: groups of U = 4 steps by which the loads run ahead of the multiply-adds */
__device__ __forceinline__ void autocorr_lane(const Plan &p, uint32_t layer, uint32_t cur, uint32_t q_end, uint32_t job, uint32_t t, bool active)
{
    constexpr int U = 4;                      /* a group must fit in the zero zone after a unit (>= 4 steps): units are flushed between groups */
    constexpr uint32_t np = K - 1;
    const DevClass &c = job_class(p, job);
    const uint32_t u = 1u << t;
    const uint32_t n = active ? (c.na / u) : 1u;
    const uint32_t upl = n + (np > 4 ? np : 4);
    const double *wt = p.wtab + (active ? c.wt_off[layer][t] : 0u);
    const int32_t *xi = p.xint + (size_t)(job / p.R) * p.S;
    const double *xd = p.sig + ((size_t)job * 2 + cur) * p.S;
    double stale = 0.0;
    if (active && (n & 1u) && u == 1u) stale = q1_stale_first_trial(p, job, c);
    else if (active && (n & 1u)) {                  /* Q1, as in k_autocorr2 */
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
    double r[K], w[K + U], fx[D][U], fw[D][U]; uint32_t fl[D][U];
#pragma unroll
    for (int j = 0; j < K; j++) {
        r[j] = 0.0;
        double rx, rw; uint32_t rl;
        gen_issue(rx, rw, rl);
        w[j] = gen_finish(rx, rw, rl);
    }
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
        for (int j = 0; j < U; j++) gen_issue(fx[d][j], fw[d][j], fl[d][j]);
    }
    uint32_t a_unit = 0, flush_pos = n;
    double *out = p.acorr + ((size_t)job * LNN_MAXT + t) * LNN_ACW;
#pragma unroll 1
    for (uint32_t q0 = 0; q0 < q_end; q0 += U * D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (active && q0 + (uint32_t)(d * U) >= flush_pos) {   /* in the zero zone after a unit: store its lags, restart */
                double *o = out + (size_t)a_unit * K;
#pragma unroll
                for (int j = 0; j < K; j++) { o[j] = r[j]; r[j] = 0.0; }
                a_unit++;
                flush_pos = (a_unit < u) ? (flush_pos + upl) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int j = 0; j < U; j++) w[K + j] = gen_finish(fx[d][j], fw[d][j], fl[d][j]);
#pragma unroll
            for (int j = 0; j < U; j++) gen_issue(fx[d][j], fw[d][j], fl[d][j]);     /* the loads of group +D fly during the MACs */
#pragma unroll
            for (int tt = 0; tt < U; tt++) {
#pragma unroll
                for (int j = 0; j < K; j++) r[j] += w[tt] * w[tt + j];
            }
#pragma unroll
            for (int j = 0; j < K; j++) w[j] = w[j + U];
        }
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

/* Layers of order <= 4 in the fast form.  The 64 rows of a block are few waves' worth of work (layer 0: rows are channel-
 * frames), and one wave cannot issue FP64 instructions back to back (tools/ubench/dp_rate.hip: a lone wave gets ~40 % of
 * its SIMD's rate), so the block spreads the lags over many lean waves: wave -> (trial TT, lags J0 .. J0+JN-1).  The rows'
 * samples pass once through a transposed LDS tile (coalesced 16-byte loads shared by the block's waves, converted and scaled
 * on the way in; lane = row reads); a wave windows each sample for its trial (the weights of a tile sit in LDS, read by
 * broadcast) and multiplies it with the trial's last values, kept in a register ring: lag j adds v[m-j] * v[m] when sample
 * m arrives -- the reference's products in the reference's order (the pairs that would reach past a unit's end simply never
 * form; in the padded-stream kernels they add +0.0).  No window generator, no stream bookkeeping: 1 + 2 JN multiply/adds
 * per sample.  Units end at multiples of the finest unit (a multiple of 4 samples), checked once per 4 samples. */
template <int P, bool L0, int TT, int J0, int JN, int NW>
__device__ __forceinline__ void autocorr_rows(const Plan &p, uint32_t layer, uint32_t cur, uint32_t row0, uint32_t nrows, uint32_t rstride,
        uint32_t na, const DevClass &c0, uint32_t wave, uint32_t lane, double (*xt)[32][65], double (*wtile)[32])
{
    constexpr int NT = AcCfg<P>::NT, T = 32, NLD = L0 ? 8 : 16, PT = P >> TT;      /* PT: order of my trial = ring length (divides 4) */
    constexpr int NSLOT = (NLD + NW - 1) / NW;
    static_assert(PT <= 32 && T % PT == 0 && J0 + JN <= PT + 1, "register rings: sample m lives in slot m % PT, and the tile loop is unrolled over T samples");
    typedef typename std::conditional<L0, int4, lnn_d2>::type XV;
    const uint32_t ntiles = na / T, seg = na >> (NT - 1), nt = na >> TT;
    uint32_t myrow = row0 + lane; if (myrow >= nrows) myrow = nrows - 1;
    const bool store = (row0 + lane) < nrows;
    double *out = p.acorr + ((size_t)myrow * rstride * LNN_MAXT + TT) * LNN_ACW + J0;
    /* tile loads: L0 -- instruction k covers rows 8k + lane/8, samples 4(lane%8)..+3; else rows 4k + lane/16, samples 2(lane%16)..+1;
     * wave w issues the instructions k = w, w + NW, ... */
    const uint32_t lrow = L0 ? (lane >> 3) : (lane >> 4), lsmp = L0 ? 4u * (lane & 7u) : 2u * (lane & 15u);
    const void *src[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; i++) {
        const uint32_t k = wave + (uint32_t)i * NW;
        uint32_t r = row0 + (L0 ? 8u : 4u) * (k < (uint32_t)NLD ? k : 0u) + lrow; if (r >= nrows) r = nrows - 1;
        if (L0) src[i] = p.xint + (size_t)r * p.S + lsmp;                   /* rows are channel-frames */
        else src[i] = p.sig + ((size_t)r * 2 + cur) * p.S + lsmp;
    }
    const double *wt = p.wtab + (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.wt_off[layer][TT]);
    /* the weights of a tile, all trials: wave t (< NT) fetches trial t's */
    const double *wts = p.wtab + (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.wt_off[layer][wave < (uint32_t)NT ? wave : 0u]);
    const uint32_t wnt = na >> (wave < (uint32_t)NT ? wave : 0u);
    uint32_t wloc = 0;
    (void)wt; (void)nt;
    XV pre[NSLOT]; double prew = 0.0;
    auto issue = [&](uint32_t tile_idx) {
#pragma unroll
        for (int i = 0; i < NSLOT; i++)
            if (wave + (uint32_t)i * NW < (uint32_t)NLD) pre[i] = *(const XV *)((L0 ? (const char *)src[i] + (size_t)tile_idx * T * 4 : (const char *)src[i] + (size_t)tile_idx * T * 8));
        if (wave < (uint32_t)NT) {
            uint32_t l = wloc + (lane & (T - 1u)); if (l >= wnt) l -= wnt;      /* wnt >= seg >= T */
            prew = wts[l];
            wloc += T; if (wloc >= wnt) wloc -= wnt;
        }
    };
    auto commit = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            const uint32_t k = wave + (uint32_t)i * NW;
            if (k < (uint32_t)NLD) {
                const uint32_t r = (L0 ? 8u : 4u) * k + lrow;
                if (L0) {
                    const int4 v = *(const int4 *)&pre[i];
                    xt[buf][lsmp][r] = (double)v.x * p.scale; xt[buf][lsmp + 1][r] = (double)v.y * p.scale;
                    xt[buf][lsmp + 2][r] = (double)v.z * p.scale; xt[buf][lsmp + 3][r] = (double)v.w * p.scale;
                } else { const lnn_d2 v = *(const lnn_d2 *)&pre[i]; xt[buf][lsmp][r] = v.x; xt[buf][lsmp + 1][r] = v.y; }
            }
        }
        if (wave < (uint32_t)NT && lane < T) wtile[buf * NT + wave][lane] = prew;
    };
    double r[JN], ring[PT], q[JN];
#pragma unroll
    for (int j = 0; j < JN; j++) { r[j] = 0.0; q[j] = 0.0; }
#pragma unroll
    for (int j = 0; j < PT; j++) ring[j] = 0.0;
    /* The two tiles form a ring of 2T samples.  Hand-pipelined: in one step the adds of sample m-1's products, the products
     * of sample m, the windowing of sample m+1 (independent of each other), with the LDS reads of the next group in flight. */
    auto read_group = [&](uint32_t slot, double *xv, double *wv) {         /* slot: ring position of the group, multiple of 4 */
        const uint32_t buf = slot / T, o = slot % T;
#pragma unroll
        for (int i = 0; i < 4; i++) xv[i] = xt[buf][o + i][lane];
        const lnn_d2 a = *(const lnn_d2 *)&wtile[buf * NT + TT][o], b = *(const lnn_d2 *)&wtile[buf * NT + TT][o + 2];
        wv[0] = a.x; wv[1] = a.y; wv[2] = b.x; wv[3] = b.y;
    };
    issue(0); commit(0);
    __syncthreads();
    double xv[4], wv[4], vcur;
    read_group(0, xv, wv);
    vcur = xv[0] * wv[0];
    uint32_t to_b = seg, bcount = 0, unit = 0;
#pragma unroll 1
    for (uint32_t ti = 0; ti < ntiles; ti++) {
        const uint32_t buf = ti & 1u;
        if (ti + 1 < ntiles) issue(ti + 1);
#pragma unroll
        for (int g = 0; g < T / 4; g++) {
            if (g == T / 4 - 1) {                                   /* the next group lies in the other tile: publish it */
                if (ti + 1 < ntiles) commit(buf ^ 1u);
                __syncthreads();
            }
            double nx[4], nw[4];
            read_group((buf * T + 4 * (uint32_t)g + 4) % (2 * T), nx, nw);      /* past the last tile: stale data, never used */
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int j = 0; j < JN; j++) r[j] += q[j];          /* adds of the previous sample's products */
                const double v = vcur;
#pragma unroll
                for (int j = 0; j < JN; j++) q[j] = (J0 + j == 0) ? (v * v) : (ring[((4 * g + i - (J0 + j)) % PT + PT) % PT] * v);
                ring[(4 * g + i) % PT] = v;
                vcur = (i < 3) ? (xv[(i + 1) & 3] * wv[(i + 1) & 3]) : (nx[0] * nw[0]);          /* window the next one */
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { xv[i] = nx[i]; wv[i] = nw[i]; }
            to_b -= 4;
            if (to_b == 0) {                                        /* a finest unit ends here; my trial's unit with every 2^(NT-1-TT)-th */
                to_b = seg; bcount++;
                if ((bcount & ((1u << (NT - 1 - TT)) - 1u)) == 0) {
#pragma unroll
                    for (int j = 0; j < JN; j++) { r[j] += q[j]; q[j] = 0.0; }           /* the unit's last products */
                    if (store) {
                        double *o = out + (size_t)unit * (PT + 1);
#pragma unroll
                        for (int j = 0; j < JN; j++) o[j] = r[j];
                    }
#pragma unroll
                    for (int j = 0; j < JN; j++) r[j] = 0.0;
#pragma unroll
                    for (int j = 0; j < PT; j++) ring[j] = 0.0;
                    unit++;
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
template <> struct AcsWaves<4>  { static constexpr int NW = 10; };
template <> struct AcsWaves<2>  { static constexpr int NW = 5; };

#define ACS_ROWS(P_) ((P_) <= 4 || p.rows16)      /* LINNE_AMD_ROWS16=0: orders 8 and 16 keep the shared-tile form */
template <int P, bool L0>
__global__ __launch_bounds__(64 * AcsWaves<P>::NW, (P >= 8) ? 4 : 5) void k_autocorr_lane(Plan p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    using Cfg = AcCfg<P>;
    constexpr int NT = Cfg::NT, NW = AcsWaves<P>::NW;
    __shared__ double tile[2][ACS_T][65];
    __shared__ __attribute__((aligned(16))) double wts_mem[2 * NT * 32];     /* autocorr_rows: the weights of two tiles, every trial */
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t rstride = L0 ? p.R : 1u;
    /* my block of 64 rows of one class run (RowRuns); the blocks are taken in reverse so that a ragged last frame, whose
     * block may need the slow general form, starts first and runs beside the others */
    const RowRuns &rr = p.runs[L0 ? 0 : 1];
    const uint32_t b = gridDim.x - 1u - blockIdx.x;
    uint32_t run = 0;
    while (run + 1 < rr.n && b >= rr.blk_begin[run + 1]) run++;
    const uint32_t row0 = rr.row_begin[run] + (b - rr.blk_begin[run]) * 64u, nrows = rr.row_begin[run + 1];
    uint32_t row = row0 + lane;
    const bool inrange = row < nrows;
    if (!inrange) row = nrows - 1;
    const uint32_t job = row * rstride;
    const uint32_t ci = p.cls_of_frame[(job / p.R) / p.C];
    const uint32_t ci0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
    const DevClass &c0 = p.cls[ci0];
    const uint32_t na = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.na);
    const bool fast = __all(ci == ci0) && c0.ntrials[layer] == (uint32_t)NT && (na % (4u << (NT - 1))) == 0 && (na % ACS_T) == 0;
    constexpr int NWS = (P == 4) ? 3 : (P == 2) ? 2 : NW;     /* waves of the shared-tile form (orders <= 4 run it only when autocorr_rows cannot) */
#define ACS_RUN(T_, K_, J0_, JN_) autocorr_shared<K_, J0_, JN_, L0, NWS>(p, layer, cur, row0, nrows, rstride, na, \
        (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.wt_off[layer][T_]), T_, wave, lane, tile)
    if constexpr (P <= 16) {
        if (fast && (na >> (NT - 1)) >= 32u && (p.S & 3u) == 0 && ACS_ROWS(P)) {
            double (*wtile)[32] = (double (*)[32])wts_mem;
#define ROWS_RUN(T_, J0_, JN_) autocorr_rows<P, L0, T_, J0_, JN_, NW>(p, layer, cur, row0, nrows, rstride, na, c0, wave, lane, tile, wtile)
            if constexpr (P == 16) switch (wave) {              /* as in the shared-tile form: waves w and w+4 share a SIMD */
                case 0: ROWS_RUN(0, 10, 7); break; case 1: ROWS_RUN(0, 0, 5); break; case 2: ROWS_RUN(0, 5, 5); break; case 3: ROWS_RUN(1, 0, 5); break;
                case 4: ROWS_RUN(4, 0, 2); break;  case 5: ROWS_RUN(1, 5, 4); break; case 6: ROWS_RUN(2, 0, 5); break; default: ROWS_RUN(3, 0, 3); break;
            } else if constexpr (P == 8) switch (wave) {
                case 0: ROWS_RUN(0, 0, 5); break; case 1: ROWS_RUN(0, 5, 4); break; case 2: ROWS_RUN(1, 0, 5); break; case 3: ROWS_RUN(2, 0, 3); break;
                default: ROWS_RUN(3, 0, 2); break;
            } else if constexpr (P == 4) switch (wave) {        /* few rows (channel-frames), few lags: one lag per wave keeps more SIMDs busy */
                case 0: ROWS_RUN(0, 0, 1); break; case 1: ROWS_RUN(0, 1, 1); break; case 2: ROWS_RUN(0, 2, 1); break; case 3: ROWS_RUN(0, 3, 1); break;
                case 4: ROWS_RUN(0, 4, 1); break; case 5: ROWS_RUN(1, 0, 1); break; case 6: ROWS_RUN(1, 1, 1); break; case 7: ROWS_RUN(1, 2, 1); break;
                case 8: ROWS_RUN(2, 0, 1); break; default: ROWS_RUN(2, 1, 1); break;
            } else switch (wave) {
                case 0: ROWS_RUN(0, 0, 1); break; case 1: ROWS_RUN(0, 1, 1); break; case 2: ROWS_RUN(0, 2, 1); break;
                case 3: ROWS_RUN(1, 0, 1); break; default: ROWS_RUN(1, 1, 1); break;
            }
#undef ROWS_RUN
            return;
        }
    }
    if (fast) {                                             /* every wave of the block sees the same rows: same decision */
        if (P == 16) switch (wave) {                        /* waves w and w+4 share a SIMD: 9 / 9 / 10 / 8 lags per SIMD */
            case 0: ACS_RUN(0, 17, 10, 7); break; case 1: ACS_RUN(0, 17, 0, 5); break;  case 2: ACS_RUN(0, 17, 5, 5); break;
            case 3: ACS_RUN(1, 9, 0, 5); break;   case 4: ACS_RUN(4, 2, 0, 2); break;   case 5: ACS_RUN(1, 9, 5, 4); break;
            case 6: ACS_RUN(2, 5, 0, 5); break;   default: ACS_RUN(3, 3, 0, 3); break;
        } else if (P == 8) switch (wave) {
            case 0: ACS_RUN(0, 9, 0, 5); break;   case 1: ACS_RUN(0, 9, 5, 4); break;   case 2: ACS_RUN(1, 5, 0, 5); break;
            case 3: ACS_RUN(2, 3, 0, 3); break;   default: ACS_RUN(3, 2, 0, 2); break;
        } else if (P == 4) switch (wave) {
            case 0: ACS_RUN(0, 5, 0, 5); break;   case 1: ACS_RUN(1, 3, 0, 3); break;   case 2: ACS_RUN(2, 2, 0, 2); break;   default: break;
        } else switch (wave) {
            case 0: ACS_RUN(0, 3, 0, 3); break;   case 1: ACS_RUN(1, 2, 0, 2); break;   default: break;
        }
        return;
    }
#undef ACS_RUN
    /* general form: the first NT waves take one whole trial each */
    if (wave >= (uint32_t)NT) return;
    const uint32_t t = wave;
    const bool active = inrange && (t < job_class(p, job).ntrials[layer]);
    /* a wave of this form waits for its loads every group of 4 steps: the few-lag layers, whose groups hold little arithmetic
     * and whose kernel has registers to spare, keep 4 groups of loads in flight */
    constexpr int D = 1;
    const uint32_t q_end = na_max + Cfg::MAXPAD + 8 + 4 * D;
    switch (P >> t) {       /* wave-uniform: the trial fixes the number of lags */
    case 16: if (P >= 16) autocorr_lane<17, L0, D>(p, layer, cur, q_end, job, t, active); break;
    case 8:  if (P >= 8)  autocorr_lane<9, L0, D>(p, layer, cur, q_end, job, t, active); break;
    case 4:  if (P >= 4)  autocorr_lane<5, L0, D>(p, layer, cur, q_end, job, t, active); break;
    case 2:  autocorr_lane<3, L0, D>(p, layer, cur, q_end, job, t, active); break;
    default: autocorr_lane<2, L0, D>(p, layer, cur, q_end, job, t, active); break;
    }
}

template <int P> static void launch_autocorr_small(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    const RowRuns &rr = p.runs[layer == 0 ? 0 : 1];
    const dim3 grid(rr.blk_begin[rr.n]);
    if (layer == 0) hipLaunchKernelGGL((k_autocorr_lane<P, true>), grid, dim3(64 * AcsWaves<P>::NW), 0, st, p, layer, cur, na_max);
    else hipLaunchKernelGGL((k_autocorr_lane<P, false>), grid, dim3(64 * AcsWaves<P>::NW), 0, st, p, layer, cur, na_max);
}

/* Short layers when the batch is small and every unit length is even: one block per row (a job; for layer 0 a channel-frame,
 * whose lags serve every regulariser pass).  The lag products of a chunk of samples go to LDS, then lane c of wave 0 adds the
 * products of chain c = (trial, lag) in sample order and stores a unit's lag when the unit ends: the 10240-step dependent chain
 * is nothing but adds fed from LDS.  The three stages -- window the samples of chunk k + 2 (waves 1-3), multiply the pairs of
 * chunk k + 1 (waves 1-3), add up chunk k (wave 0) -- run side by side on double buffers with ONE barrier per chunk, so a row
 * takes about as long as its chains (~60 us; the single-buffered form with three barriers per chunk took 0.6 ms, the waves of
 * k_autocorr_lane ~1 ms) -- but only a few lanes are busy, so the lane kernel wins once the batch fills the chip. */
template <int P> struct ApCfg {
    static constexpr int NT = (P >= 16) ? 5 : (P >= 8) ? 4 : (P >= 4) ? 3 : 2;
    static constexpr int nch() { int s = 0; for (int t = 0; t < NT; t++) s += (P >> t) + 1; return s; }      /* 36, 19, 10, 5 */
    static constexpr int NCH = nch();
    static constexpr int CHUNK = (P >= 16) ? 128 : (P >= 8) ? 128 : 256;      /* (64 for P = 16 was measured: 162 trips of ~1.1 us, whatever a trip held) */
    static constexpr int trial_of(int ch) { int t = 0; while (ch >= (P >> t) + 1) { ch -= (P >> t) + 1; t++; } return t; }
    static constexpr int lag_of(int ch) { int t = 0; while (ch >= (P >> t) + 1) { ch -= (P >> t) + 1; t++; } return ch; }
};
template <int P, bool L0, int NTH>       /* NTH threads: wave 0 adds, the others produce (512 for the 16-tap layer: its 36 chains x 128 samples of products per trip
                                          * took three producer waves as long as the adder's 128 adds) */
__global__ __launch_bounds__(NTH) void k_autocorr_prod(Plan p, uint32_t layer, uint32_t cur)
{
    using Cfg = ApCfg<P>;
    constexpr int NT = Cfg::NT, NCH = Cfg::NCH, CHUNK = Cfg::CHUNK, NPROD = NTH - 64;      /* producer threads: waves 1 .. */
    static_assert(NCH <= 64, "one wave adds up all chains");
    __shared__ double sv[2][NT][CHUNK + P];
    __shared__ uint32_t srem[2][NT][CHUNK + P];
    /* rows padded by two doubles: adder lane c walks row c, and a lane stride of CHUNK doubles (512 B .. 2 KB) would put every lane
     * on the same LDS bank -- an NCH-way conflict on every read of the dependent chain; two keep a row 16-byte aligned (ds_read_b128) */
    __shared__ __attribute__((aligned(16))) double sprod[2][NCH][CHUNK + 2];
    const uint32_t row = blockIdx.x, tid = threadIdx.x, job = L0 ? row * p.R : row;
    const DevClass &c = job_class(p, job);
    const uint32_t na = c.na;
    const int32_t *xi = p.xint + (size_t)(job / p.R) * p.S;
    const double *xd = p.sig + ((size_t)job * 2 + cur) * p.S;
    const uint32_t nchunks = (na + CHUNK - 1) / CHUNK;
    const bool producer = tid >= 64u;
    const uint32_t ptid = tid - 64u;
    /* my chain (wave 0, lanes below NCH): trial, lag, unit length */
    uint32_t ct = 0, clag = tid;
    while (ct + 1 < (uint32_t)NT && clag >= ((uint32_t)P >> ct) + 1) { clag -= ((uint32_t)P >> ct) + 1; ct++; }
    const uint32_t cn = na >> ct, cnp = (uint32_t)P >> ct;
    double r = 0.0;
    uint32_t cloc = 0, cunit = 0;
    double *cout = p.acorr + ((size_t)job * LNN_MAXT + ct) * LNN_ACW + clag;
    /* a producer's share of a chunk's CHUNK + P samples: the sample and its window weight under every trial, requested one
     * iteration before they are multiplied and written to LDS (a load waited for on the spot would cost every chunk a trip to
     * memory); beside each windowed sample goes the number of samples left in its unit: a pair (i, i + lag) counts while lag does
     * not exceed it */
    constexpr int NPRE = (CHUNK + P + NPROD - 1) / NPROD;
    double px[NPRE], pw[NPRE][NT];
    uint32_t wloc[NPRE][NT], prem[NPRE][NT];                    /* place inside the unit of my samples in the chunk to be requested next; samples left, of the requested ones */
#pragma unroll
    for (int m = 0; m < NPRE; m++)
#pragma unroll
        for (int t = 0; t < NT; t++) wloc[m][t] = (ptid + (uint32_t)m * NPROD) % (na >> t);
    auto prefetch = [&](uint32_t kk) {
        const uint32_t base = kk * CHUNK;
#pragma unroll
        for (int m = 0; m < NPRE; m++) {
            const uint32_t i = ptid + (uint32_t)m * NPROD, g = base + i;
            const bool in = i < (uint32_t)(CHUNK + P) && g < na;
            px[m] = in ? (L0 ? ((double)xi[g] * p.scale) : xd[g]) : 0.0;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint32_t nt = na >> t;
                pw[m][t] = in ? p.wtab[c.wt_off[layer][t] + wloc[m][t]] : 0.0;
                prem[m][t] = nt - 1u - wloc[m][t];
                uint32_t l = wloc[m][t] + (uint32_t)CHUNK;
                while (l >= nt) l -= nt;
                wloc[m][t] = l;
            }
        }
    };
    /* a producer's share of a chunk's NCH x CHUNK pairs: element e = ptid + 192 m is chain e / CHUNK at sample e % CHUNK */
    constexpr int NPM = (NCH * CHUNK + NPROD - 1) / NPROD;
    uint32_t pe[NPM];                                            /* packed: sample (12 bits) | chain << 12 | lag << 20 | trial << 28; 0xFFFFFFFF: none */
#pragma unroll
    for (int m = 0; m < NPM; m++) {
        const uint32_t e = ptid + (uint32_t)m * NPROD, ch = e / (uint32_t)CHUNK, i = e % (uint32_t)CHUNK;
        uint32_t t = 0, lag = ch;
        while (t + 1 < (uint32_t)NT && lag >= ((uint32_t)P >> t) + 1) { lag -= ((uint32_t)P >> t) + 1; t++; }
        pe[m] = (ch < (uint32_t)NCH) ? (i | (ch << 12) | (lag << 20) | (t << 28)) : 0xFFFFFFFFu;
    }
    if (producer) prefetch(0);
    /* iteration k: window chunk k, multiply chunk k - 1, add chunk k - 2 */
    for (uint32_t k = 0; k < nchunks + 2u; k++) {
        if (producer) {
            if (k < nchunks) {                                      /* the windowed samples base .. base + CHUNK + P - 1 of every trial */
                double (*dst)[CHUNK + P] = sv[k & 1u];
                uint32_t (*drem)[CHUNK + P] = srem[k & 1u];
#pragma unroll
                for (int m = 0; m < NPRE; m++) {
                    const uint32_t i = ptid + (uint32_t)m * NPROD;
                    if (i < (uint32_t)(CHUNK + P)) {
#pragma unroll
                        for (int t = 0; t < NT; t++) { dst[t][i] = px[m] * pw[m][t]; drem[t][i] = prem[m][t]; }     /* (0.0 * 0.0 beyond the frame's end) */
                    }
                }
                if (k + 1u < nchunks && !LNN_DBG_IS(p, 103u)) prefetch(k + 1u);
            }
            if (k >= 1u && k - 1u < nchunks && !LNN_DBG_IS(p, 102u)) {                      /* the pairs of chunk k - 1: both samples inside the same unit, else +0.0 (no effect on the bits) */
                const double (*src)[CHUNK + P] = sv[(k - 1u) & 1u];
                const uint32_t (*rem)[CHUNK + P] = srem[(k - 1u) & 1u];
                double (*dst)[CHUNK + 2] = sprod[(k - 1u) & 1u];
                /* all reads first, then all writes: the compiler cannot know that `src` and `dst` never overlap and would wait out
                 * every element's LDS trip before starting the next */
                double av[NPM], bv[NPM]; uint32_t rv[NPM];
#pragma unroll
                for (int m = 0; m < NPM; m++) {
                    const uint32_t e = (pe[m] != 0xFFFFFFFFu) ? pe[m] : 0u;
                    const uint32_t i = e & 0xFFFu, lag = (e >> 20) & 0xFFu, t = e >> 28;
                    av[m] = src[t][i]; bv[m] = src[t][i + lag]; rv[m] = rem[t][i];
                }
#pragma unroll
                for (int m = 0; m < NPM; m++) {
                    const uint32_t e = pe[m];
                    if (e != 0xFFFFFFFFu) {
                        const uint32_t i = e & 0xFFFu, ch = (e >> 12) & 0xFFu, lag = (e >> 20) & 0xFFu;
                        dst[ch][i] = (lag <= rv[m]) ? av[m] * bv[m] : 0.0;
                    }
                }
            }
        } else if (k >= 2u && tid < (uint32_t)NCH && !LNN_DBG_IS(p, 101u)) {                /* add up chunk k - 2 */
            const uint32_t base = (k - 2u) * CHUNK;
            const uint32_t cnt = (na - base < (uint32_t)CHUNK) ? (na - base) : (uint32_t)CHUNK;
            const double *q = sprod[k & 1u][tid];
            uint32_t i = 0;
            while (i < cnt) {                                       /* runs that end at the chunk's or the unit's end */
                const uint32_t seg = (cnt - i < cn - cloc) ? (cnt - i) : (cn - cloc), end = i + seg;
                /* The chain is all this lane does, and a lone wave issues an instruction every ~9 cycles: what counts is the number
                 * of instructions per add.  Sixteen products per trip, in two register sets: the next eight are requested before the
                 * current eight are added, nothing is copied (1.7 instructions per add; the plain loop's 8 reads + 8 adds + waits came to
                 * 30 cycles per add, 127 us for the 4-tap layer of one block).  Unit lengths and chunks are even, so i is: 16-byte reads */
#define AP_LOAD(A, OFF) { const lnn_d2 *q2_ = (const lnn_d2 *)(q + i + (OFF)); A##0 = q2_[0]; A##1 = q2_[1]; A##2 = q2_[2]; A##3 = q2_[3]; }
#define AP_ADD(A) { r += (A##0).x; r += (A##0).y; r += (A##1).x; r += (A##1).y; r += (A##2).x; r += (A##2).y; r += (A##3).x; r += (A##3).y; }
                if (i + 16 <= end) {
                    lnn_d2 a0, a1, a2, a3, b0, b1, b2, b3;
                    AP_LOAD(a, 0)
                    for (; i + 16 <= end; i += 16) {
                        AP_LOAD(b, 8)
                        AP_ADD(a)
                        if (i + 24 <= end) AP_LOAD(a, 16)
                        AP_ADD(b)
                    }
                }
                if (i + 8 <= end) { lnn_d2 a0, a1, a2, a3; AP_LOAD(a, 0) AP_ADD(a) i += 8; }
#undef AP_LOAD
#undef AP_ADD
                for (; i < end; i++) r += q[i];
                cloc += seg;
                if (cloc == cn) { cout[(size_t)cunit * (cnp + 1)] = r; r = 0.0; cloc = 0; cunit++; }
            }
        }
        __syncthreads();
    }
}

template <int P> static void launch_autocorr2(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, uint32_t na_max)
{
    using Cfg = AcCfg<P>;
    const uint32_t blocks = (p.J + Cfg::JPW - 1) / Cfg::JPW;
    if (layer == 0) hipLaunchKernelGGL((k_autocorr2<P, true>), dim3(blocks), dim3(64), 0, st, p, layer, cur, na_max);
    else hipLaunchKernelGGL((k_autocorr2<P, false>), dim3(blocks), dim3(64), 0, st, p, layer, cur, na_max);
}
#include "lnn_k_autocorr_hist.h"

/* the long layer's lanes = jobs kernels, for the frames they take (hist_takes); which: 0 / 1 the trials of order P and P/2
 * (k_autocorr_hist), 2 the shorter ones (k_autocorr_sub).  Returns false when the layer has no such kernel. */
static bool launch_autocorr_hist(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, int which)
{
    const RowRuns &rr = p.runs[1];
    const dim3 grid(rr.blk_begin[rr.n]);
    if (p.P[layer] == 128u) {
        if (which == 0) hipLaunchKernelGGL((k_autocorr_hist<128, 0>), grid, dim3(64 * HIST_WAVES(128)), 0, st, p, layer, cur);
        else if (which == 1) hipLaunchKernelGGL((k_autocorr_hist<128, 1>), grid, dim3(64 * HIST_WAVES(64)), 0, st, p, layer, cur);
        else hipLaunchKernelGGL((k_autocorr_sub<128>), grid, dim3(640), 0, st, p, layer, cur);
        return true;
    }
    if (p.P[layer] == 64u) {
        if (which == 0) hipLaunchKernelGGL((k_autocorr_hist<64, 0>), grid, dim3(64 * HIST_WAVES(64)), 0, st, p, layer, cur);
        else if (which == 2) hipLaunchKernelGGL((k_autocorr_sub<64>), grid, dim3(640), 0, st, p, layer, cur);
        else return false;
        return true;
    }
    return false;
}

/* Long layers when the batch is small (block-at-a-time calls) and every unit length is even: one block per (row, trial), lane =
 * lag.  A unit's windowed samples are staged AW_TILE at a time with np zeros behind the unit's end (a pair that leaves the unit
 * adds +-0.0: no effect on the bits), the next tile's samples and weights requested before this one is worked on; a lane adds
 * xs[j] * xs[j + lag] in sample order -- one chain per lag, as the reference has it -- reading xs[j] at one address for the whole
 * wave and xs[j + lag] at consecutive ones.  A trial takes ~50 us, the eight trials of a job run side by side; k_autocorr2,
 * whose lanes own five lags each and walk the whole frame, takes 0.5 ms however few jobs there are. */
#define AW_TILE 512
template <int P>
__global__ __launch_bounds__(64 * ((P + 64) / 64)) void k_autocorr_wide(Plan p, uint32_t layer, uint32_t cur)
{
    constexpr int NTHR = 64 * ((P + 64) / 64);                         /* lanes for P + 1 lags */
    constexpr int NPRE = (AW_TILE + P + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) double xs[AW_TILE + P + 8];
    const uint32_t row = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const bool l0 = (layer == 0);
    const uint32_t job = l0 ? row * p.R : row;
    const DevClass &c = job_class(p, job);
    if (t >= c.ntrials[layer]) return;
    const uint32_t na = c.na, u = 1u << t, np = (uint32_t)P >> t, nt = na >> t;
    const int32_t *xi = p.xint + (size_t)(job / p.R) * p.S;
    const double *xd = p.sig + ((size_t)job * 2 + cur) * p.S;
    const double *wt = p.wtab + c.wt_off[layer][t];
    double *out = p.acorr + ((size_t)job * LNN_MAXT + t) * LNN_ACW + tid;
    const bool mine = tid <= np;                                        /* my lag */
    const uint32_t tiles_per_unit = (nt + AW_TILE - 1) / AW_TILE, ntiles = u * tiles_per_unit;
    double px[NPRE], pwv[NPRE];
    auto prefetch = [&](uint32_t ti) {                                  /* tile ti: unit ti / tiles_per_unit, places loc0 .. loc0 + AW_TILE + np - 1 */
        const uint32_t un = ti / tiles_per_unit, loc0 = (ti - un * tiles_per_unit) * AW_TILE;
#pragma unroll
        for (int m = 0; m < NPRE; m++) {
            const uint32_t i = tid + (uint32_t)m * NTHR, loc = loc0 + i;
            const bool in = i < (uint32_t)AW_TILE + np && loc < nt;
            const uint32_t g = un * nt + loc;
            px[m] = in ? (l0 ? ((double)xi[g] * p.scale) : xd[g]) : 0.0;
            pwv[m] = in ? wt[loc] : 0.0;
        }
    };
    prefetch(0);
    double acc = 0.0;
    for (uint32_t ti = 0; ti < ntiles; ti++) {
        const uint32_t un = ti / tiles_per_unit, loc0 = (ti - un * tiles_per_unit) * AW_TILE;
        __syncthreads();                                                /* the tile before is through */
#pragma unroll
        for (int m = 0; m < NPRE; m++) { const uint32_t i = tid + (uint32_t)m * NTHR; if (i < (uint32_t)AW_TILE + np) xs[i] = px[m] * pwv[m]; }
        if (ti + 1 < ntiles) prefetch(ti + 1);
        __syncthreads();
        if (mine) {
            const uint32_t cnt = (nt - loc0 < (uint32_t)AW_TILE) ? (nt - loc0) : (uint32_t)AW_TILE;
            const double *xa = xs, *xb = xs + tid;
            uint32_t j = 0;
            /* 32 (16) samples a trip: every trip waits out one LDS round trip before its first add (the adds themselves are ~18 cycles
             * apart: the FP64 unit's dependent latency), so the trip is made long */
            for (; j + 32 <= cnt; j += 32) {
                double m[32];
#pragma unroll
                for (int q = 0; q < 32; q += 2) { const lnn_d2 a = *(const lnn_d2 *)(xa + j + q); m[q] = a.x * xb[j + q]; m[q + 1] = a.y * xb[j + q + 1]; }
#pragma unroll
                for (int q = 0; q < 32; q++) acc += m[q];
            }
            for (; j + 16 <= cnt; j += 16) {
                double m[16];
#pragma unroll
                for (int q = 0; q < 16; q += 2) { const lnn_d2 a = *(const lnn_d2 *)(xa + j + q); m[q] = a.x * xb[j + q]; m[q + 1] = a.y * xb[j + q + 1]; }
#pragma unroll
                for (int q = 0; q < 16; q++) acc += m[q];
            }
            for (; j + 8 <= cnt; j += 8) {
                const lnn_d2 a0 = *(const lnn_d2 *)(xa + j), a1 = *(const lnn_d2 *)(xa + j + 2), a2 = *(const lnn_d2 *)(xa + j + 4), a3 = *(const lnn_d2 *)(xa + j + 6);
                const double b0 = xb[j], b1 = xb[j + 1], b2 = xb[j + 2], b3 = xb[j + 3], b4 = xb[j + 4], b5 = xb[j + 5], b6 = xb[j + 6], b7 = xb[j + 7];
                const double m0 = a0.x * b0, m1 = a0.y * b1, m2 = a1.x * b2, m3 = a1.y * b3, m4 = a2.x * b4, m5 = a2.y * b5, m6 = a3.x * b6, m7 = a3.y * b7;
                acc += m0; acc += m1; acc += m2; acc += m3; acc += m4; acc += m5; acc += m6; acc += m7;
            }
            for (; j < cnt; j++) acc += xa[j] * xb[j];
            if (loc0 + cnt == nt) { out[(size_t)un * (np + 1)] = acc; acc = 0.0; }     /* the unit is through */
        }
    }
}

static void dispatch_autocorr2(hipStream_t st, const Plan &p, uint32_t layer, uint32_t cur, uint32_t na_max, bool prod_ok)
{
    /* the product form has a short dependent chain but few busy lanes: it wins while the batch is too small to fill the chip
     * with k_autocorr_lane's long-running waves */
    if (prod_ok && p.P[layer] <= 16u) {
        const uint32_t rows = (layer == 0) ? p.J / p.R : p.J;
        const bool small = (layer == 0) ? (rows < 6144u) : (rows <= 512u);
        if (small) {
#define LNN_AP(PP, NTH_) do { if (layer == 0) hipLaunchKernelGGL((k_autocorr_prod<PP, true, NTH_>), dim3(rows), dim3(NTH_), 0, st, p, layer, cur); \
                        else hipLaunchKernelGGL((k_autocorr_prod<PP, false, NTH_>), dim3(rows), dim3(NTH_), 0, st, p, layer, cur); } while (0)
            switch (p.P[layer]) { case 2: LNN_AP(2, 256); break; case 4: LNN_AP(4, 256); break; case 8: LNN_AP(8, 512); break; default: LNN_AP(16, 512); break; }
#undef LNN_AP
            return;
        }
    }
    if (prod_ok && p.P[layer] >= 32u) {          /* (the host sets the bit for a long layer only when the batch is small: wide_ok) */
        const uint32_t rows = (layer == 0) ? p.J / p.R : p.J;
        uint32_t ntr = 0; for (uint32_t u = 1; u <= (p.P[layer] < (uint32_t)LNN_MAXU ? p.P[layer] : (uint32_t)LNN_MAXU); u <<= 1) ntr++;
        switch (p.P[layer]) {
        case 32: hipLaunchKernelGGL((k_autocorr_wide<32>), dim3(rows, ntr), dim3(64), 0, st, p, layer, cur); break;       /* (block = the kernel's launch bound: 64 lanes per 64 lags, rounded up from P + 1) */
        case 64: hipLaunchKernelGGL((k_autocorr_wide<64>), dim3(rows, ntr), dim3(128), 0, st, p, layer, cur); break;
        default: hipLaunchKernelGGL((k_autocorr_wide<128>), dim3(rows, ntr), dim3(192), 0, st, p, layer, cur); break;
        }
        return;
    }
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


#endif

/* lnn_k_prep.h -- k_prep (copy, MS, pre-emphasis) and k_stats (block-type statistics).
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_PREP_H_INCLUDED
#define LNN_K_PREP_H_INCLUDED

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

/* the general form: every phase streams the channel through global memory (xint <-> xtmp); any block length, and the ordered
 * double chains for material whose pre-emphasis correlations leave the exact-integer range */
__device__ void prep_general(const Plan &p, int64_t *sh, int32_t &sh_coef, double (*sh_prod)[PREP_CHUNK])
{
    const uint32_t f = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;     /* one block per (frame, channel) */
    const DevClass &c = p.cls[p.cls_of_frame[f]];
    const uint32_t n = c.n, S = p.S, C = p.C;
    const uint32_t fo = p.frame_map[f];                      /* the caller's frame behind row f of the class-sorted chunk */
    const size_t inbase = (size_t)fo * C * S;                /* input PCM: int32, or int16 when the caller staged 16-bit samples (Plan.pcm16) */
    int32_t *src = p.xint + ((size_t)f * C + ch) * S, *dst = p.xtmp + ((size_t)f * C + ch) * S;
    int32_t *rec = p.prm + ((size_t)fo * C + ch) * LINNE_AMD_PARAM_WORDS;

    /* copy with zero padding (linne_encoder.c:613-621); LR -> MS on channels 0/1 (linne_utility.c:120-132): each of
     * the two blocks derives its own channel from L and R */
    for (uint32_t s = tid; s < S; s += PREP_THREADS) {
        int32_t v = 0;
        if (s < n) {
            v = pcm_at(p, inbase + (size_t)ch * S + s);
            if (p.ms && ch < 2) {
                const uint32_t l = (uint32_t)pcm_at(p, inbase + s), r = (uint32_t)pcm_at(p, inbase + (size_t)S + s);
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

/* one round's tile of the fast form: samples t0 .. t0 + 2562 of the channel after copy / zero padding / LR -> MS (0 outside
 * [0, n)), coalesced, ALL of a thread's loads issued before the first is used (a load waited for on the spot costs the block a trip
 * to memory per element: 11 trips per round) */
template <int FMT, bool MS>           /* FMT: 0 int32, 1 int16, 2 packed 3-byte samples */
__device__ __forceinline__ void prep2_fill_t(const Plan &p, int32_t *tile, int64_t t0, uint32_t n, size_t inbase, uint32_t ch, uint32_t tid)
{
    constexpr uint32_t NQ = (PREP_THREADS * 10u + 3u + PREP_THREADS - 1u) / PREP_THREADS;      /* 11 */
    const int16_t *p16 = (const int16_t *)p.pcm;
    const size_t b0 = MS ? inbase : inbase + (size_t)ch * p.S, b1 = inbase + (size_t)p.S;
    int32_t a[NQ], b[NQ];
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++) {
        const int64_t s = t0 + (int64_t)(tid + PREP_THREADS * q);
        const size_t idx = (s >= 0 && s < (int64_t)n) ? (size_t)s : 0u;                       /* (n >= 1: index 0 is always readable) */
        a[q] = FMT == 1 ? (int32_t)p16[b0 + idx] : (FMT == 2 ? pcm24_at(p.pcm, b0 + idx) : p.pcm[b0 + idx]);
        b[q] = MS ? (FMT == 1 ? (int32_t)p16[b1 + idx] : (FMT == 2 ? pcm24_at(p.pcm, b1 + idx) : p.pcm[b1 + idx])) : 0;
    }
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++) {
        const uint32_t i = tid + PREP_THREADS * q;
        const int64_t s = t0 + (int64_t)i;
        int32_t v = a[q];
        if (MS) { const int32_t side = (int32_t)((uint32_t)b[q] - (uint32_t)a[q]); v = (ch == 1) ? side : (int32_t)((uint32_t)a[q] + (uint32_t)(side >> 1)); }
        if (s < 0 || s >= (int64_t)n) v = 0;
        if (i < PREP_THREADS * 10u + 3u) tile[i] = v;
    }
}
__device__ __forceinline__ void prep2_fill(const Plan &p, int32_t *tile, int64_t t0, uint32_t n, size_t inbase, uint32_t ch, bool ms, uint32_t tid)
{
    if (p.pcm16 == 1u)      { if (ms) prep2_fill_t<1, true>(p, tile, t0, n, inbase, ch, tid); else prep2_fill_t<1, false>(p, tile, t0, n, inbase, ch, tid); }
    else if (p.pcm16 == 2u) { if (ms) prep2_fill_t<2, true>(p, tile, t0, n, inbase, ch, tid); else prep2_fill_t<2, false>(p, tile, t0, n, inbase, ch, tid); }
    else                    { if (ms) prep2_fill_t<0, true>(p, tile, t0, n, inbase, ch, tid); else prep2_fill_t<0, false>(p, tile, t0, n, inbase, ch, tid); }
}

/* k_prep: copy / zero padding, MS, the two pre-emphasis stages (linne_encoder.c:613-641, linne_utility.c:120-212), one block per
 * (frame, channel).
 *
 * Fast form (blocks of up to 10 240 samples whose correlations stay exact integers -- all 16-bit material): the channel lives in
 * REGISTERS.  A thread owns four runs of 10 consecutive samples (one per 2560-sample round) with a halo of two samples in front and
 * one behind, so that everything a stage needs of its neighbours -- x[s + 1] for the cross products, the previous stage's
 * output at s - 1 for the next stage's filter -- is recomputed locally: the block talks only in its reductions (max |x|, sum x^2,
 * the two correlation sums: ONE combined reduction per stage) and through the LDS tile that makes the global loads and stores
 * coalesced.  Global traffic: the PCM in, the pre-emphasised channel out -- 80 KB per channel-frame where the general form
 * (five passes over xint / xtmp) moved 369 KB (profiles/pmc_latest.json before round 3).  Every value is the general form's:
 * same integer arithmetic, and the correlation sums are exact integers whatever the order (the `exact` test is the same).
 * A block whose sums are not exact starts over in the general form. */
#define PREP2_RUN 10u
#define PREP2_ROUND (PREP_THREADS * PREP2_RUN)          /* 2560 samples */
#define PREP2_MAXROUNDS 4u
__global__ __launch_bounds__(PREP_THREADS) void k_prep(Plan p)
{
    __shared__ int64_t sh[4][PREP_THREADS / 64];
    __shared__ int32_t sh_coef;
    __shared__ int32_t sh_exact;
    /* the tile of the fast form and the product buffers of the general form are never live together */
    __shared__ __attribute__((aligned(16))) double sh_prod[2][PREP_CHUNK];
    int32_t *tile = (int32_t *)&sh_prod[0][0];                 /* PREP2_ROUND + 3 words */
    static_assert(sizeof(double) * 2 * PREP_CHUNK >= sizeof(int32_t) * (PREP2_ROUND + 3), "tile fits the product buffers");
    const uint32_t f = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const DevClass &c = p.cls[p.cls_of_frame[f]];
    const uint32_t n = c.n, S = p.S, C = p.C;
    if (S > PREP2_ROUND * PREP2_MAXROUNDS || p.prep_general) { prep_general(p, sh[0], sh_coef, sh_prod); return; }
    const uint32_t fo = p.frame_map[f];
    const size_t inbase = (size_t)fo * C * S;
    int32_t *out = p.xint + ((size_t)f * C + ch) * S;
    int32_t *rec = p.prm + ((size_t)fo * C + ch) * LINNE_AMD_PARAM_WORDS;
    const uint32_t rounds = (S + PREP2_ROUND - 1u) / PREP2_ROUND;
    const bool ms = p.ms && ch < 2;
    auto input_at = [&](int64_t s) -> int32_t {                 /* the channel after copy / zero padding / LR -> MS; 0 outside [0, n) */
        if (s < 0 || s >= (int64_t)n) return 0;
        if (!ms) return pcm_at(p, inbase + (size_t)ch * S + (size_t)s);
        const uint32_t l = (uint32_t)pcm_at(p, inbase + (size_t)s), r = (uint32_t)pcm_at(p, inbase + (size_t)S + (size_t)s);
        const int32_t side = (int32_t)(r - l);
        return (ch == 1) ? side : (int32_t)(l + (uint32_t)(side >> 1));
    };
    /* x[r][j] = sample r * 2560 + 10 tid + j, j = -2 .. 10 (index j + 2) */
    int32_t x[PREP2_MAXROUNDS][PREP2_RUN + 3];
#pragma unroll
    for (uint32_t r = 0; r < PREP2_MAXROUNDS; r++) {
        if (r < rounds) {
            const int64_t t0 = (int64_t)r * PREP2_ROUND - 2;
            __syncthreads();
            prep2_fill(p, tile, t0, n, inbase, ch, ms, tid);
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < PREP2_RUN + 3u; j++) x[r][j] = tile[PREP2_RUN * tid + j];
        } else {
#pragma unroll
            for (uint32_t j = 0; j < PREP2_RUN + 3u; j++) x[r][j] = 0;
        }
    }
    int32_t coefs[2], firsts[2];
    /* the reduction of one stage: max |v|, sum v^2 (s < n), sum v[s]^2 and sum v[s] v[s+1] (s < n - 1) */
    auto stage_coef = [&](auto value_at, uint32_t stage) -> bool {     /* value_at(r, j): the stage's input at sample r * 2560 + 10 tid + j, j = 0 .. 10 */
        int64_t mx = 0; uint64_t s0 = 0, s1 = 0, sq = 0;
#pragma unroll
        for (uint32_t r = 0; r < PREP2_MAXROUNDS; r++) {
            if (r >= rounds) continue;
#pragma unroll
            for (uint32_t j = 0; j < PREP2_RUN; j++) {
                const uint32_t s = r * PREP2_ROUND + PREP2_RUN * tid + j;
                if (s < n) {
                    const int64_t a = value_at(r, j); const int64_t av = a < 0 ? -a : a;
                    mx = av > mx ? av : mx;
                    sq += (uint64_t)(a * a);
                    if (s + 1 < n) { const int64_t b = value_at(r, j + 1); s0 += (uint64_t)(a * a); s1 += (uint64_t)(a * b); }
                }
            }
        }
        /* one combined reduction: wave shuffles, then one LDS hop */
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int64_t m2 = __shfl_xor(mx, o); mx = m2 > mx ? m2 : mx;
            sq += (uint64_t)__shfl_xor((int64_t)sq, o); s0 += (uint64_t)__shfl_xor((int64_t)s0, o); s1 += (uint64_t)__shfl_xor((int64_t)s1, o);
        }
        __syncthreads();
        if ((tid & 63u) == 0) { sh[0][tid >> 6] = mx; sh[1][tid >> 6] = (int64_t)sq; sh[2][tid >> 6] = (int64_t)s0; sh[3][tid >> 6] = (int64_t)s1; }
        __syncthreads();
        if (tid == 0) {
            int64_t m = sh[0][0]; uint64_t q = 0, a0 = 0, a1 = 0;
            for (uint32_t w = 0; w < PREP_THREADS / 64; w++) { m = sh[0][w] > m ? sh[0][w] : m; q += (uint64_t)sh[1][w]; a0 += (uint64_t)sh[2][w]; a1 += (uint64_t)sh[3][w]; }
            /* the same test as the general form: mx^2 n < 2^62 (the 64-bit sums cannot wrap) and sum x^2 < 2^53 (every partial sum of
             * the reference's double chains is an exactly representable integer, so any order gives its bits) */
            const bool exact = (((double)m * (double)m * (double)n) < 4.0e18) && (q < (1ull << 53));
            int32_t coef = 0;
            if (exact) {
                const double c0 = (double)(int64_t)a0;
                double c1 = (double)(int64_t)a1;
                c1 /= c0;
                if ((c0 < 1e-6) || (c1 < 0.0)) coef = 0;
                else { coef = (int32_t)round_away(c1 * 32.0); if (coef >= 16) coef = 15; }
            }
            sh_coef = coef; sh_exact = exact ? 1 : 0;
        }
        __syncthreads();
        coefs[stage] = sh_coef;
        return sh_exact != 0;
    };
    /* a channel whose sums are not exact integers (loud 24-bit material) needs the reference's ordered chains: 2 x 10 239 dependent
     * adds.  With p.prep_defer it leaves this block as it came (after copy / zero padding / LR -> MS) into xtmp and goes onto
     * k_prep_slow's list, where a lane runs a channel-frame's chains -- here they would occupy two lanes of the block while 254 wait
     * (77 us per stage and block, eight blocks per CU: 9.6 ms for the 45 000 channel-frames of the 8-channel stress case) */
    auto defer = [&]() {
        int32_t *raw = p.xtmp + ((size_t)f * C + ch) * S;
#pragma unroll
        for (uint32_t r = 0; r < PREP2_MAXROUNDS; r++) {
            if (r >= rounds) continue;
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < PREP2_RUN; j++) tile[PREP2_RUN * tid + j] = x[r][j + 2];
            __syncthreads();
            for (uint32_t i = tid; i < PREP2_ROUND; i += PREP_THREADS) { const uint32_t s = r * PREP2_ROUND + i; if (s < S) raw[s] = tile[i]; }
        }
        if (tid == 0) p.prep_slow_rows[atomicAdd(p.prep_slow_n, 1u)] = f * C + ch;
    };
    /* stage 0 works on x, stage 1 on y[s] = x[s] - mulshr5(x[s ? s - 1 : 0], coef0) for s < n (0 beyond), recomputed where needed */
    const int32_t x_first = input_at(0);
    auto xv = [&](uint32_t r, uint32_t j) -> int32_t { return x[r][j + 2]; };
    if (!stage_coef(xv, 0u)) { if (p.prep_defer) { defer(); return; } __syncthreads(); prep_general(p, sh[0], sh_coef, sh_prod); return; }
    const int32_t cf0 = coefs[0];
    auto yat = [&](uint32_t r, int32_t j) -> int32_t {          /* j = -1 .. 10 */
        const int64_t s = (int64_t)r * PREP2_ROUND + (int64_t)PREP2_RUN * tid + j;
        if (s < 0 || s >= (int64_t)n) return 0;
        const int32_t prev = (s == 0) ? x[r][j + 2] : x[r][j + 1];
        return (int32_t)((uint32_t)x[r][j + 2] - (uint32_t)mulshr5(prev, cf0));
    };
    auto yv = [&](uint32_t r, uint32_t j) -> int32_t { return yat(r, (int32_t)j); };
    firsts[0] = x_first;
    const int32_t y_first = (n > 0) ? (int32_t)((uint32_t)x_first - (uint32_t)mulshr5(x_first, cf0)) : 0;
    if (!stage_coef(yv, 1u)) { if (p.prep_defer) { defer(); return; } __syncthreads(); prep_general(p, sh[0], sh_coef, sh_prod); return; }
    const int32_t cf1 = coefs[1];
    firsts[1] = y_first;
    if (tid == 0) {
        rec[LINNE_AMD_PRM_PREV + 0] = firsts[0]; rec[LINNE_AMD_PRM_PCOEF + 0] = cf0;
        rec[LINNE_AMD_PRM_PREV + 1] = firsts[1]; rec[LINNE_AMD_PRM_PCOEF + 1] = cf1;
    }
    /* z[s] = y[s] - mulshr5(y[s ? s - 1 : 0], coef1), through the tile to coalesced stores */
#pragma unroll
    for (uint32_t r = 0; r < PREP2_MAXROUNDS; r++) {
        if (r >= rounds) continue;
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < PREP2_RUN; j++) {
            const uint32_t s = r * PREP2_ROUND + PREP2_RUN * tid + j;
            int32_t z = 0;
            if (s < n) { const int32_t yc = yat(r, (int32_t)j), yp = (s == 0) ? yc : yat(r, (int32_t)j - 1); z = (int32_t)((uint32_t)yc - (uint32_t)mulshr5(yp, cf1)); }
            tile[PREP2_RUN * tid + j] = z;
        }
        __syncthreads();
        for (uint32_t i = tid; i < PREP2_ROUND; i += PREP_THREADS) { const uint32_t s = r * PREP2_ROUND + i; if (s < S) out[s] = tile[i]; }
    }
}

/* k_prep_slow: the pre-emphasis of the channel-frames k_prep could not finish with integer sums (its list: Plan.prep_slow_rows),
 * lanes = channel-frames.  linne_utility.c:158-193: corr0 = sum x[i]^2 and corr1 = sum x[i] x[i+1], i < n - 1, are double chains in
 * sample order -- 10 239 dependent adds each, per stage.  A block takes 64 listed rows and walks them in tiles of 64 samples
 * (coalesced 16-byte loads from xtmp, transposed through LDS, two buffers, one barrier per tile):
 *   waves 0, 3   load tile t + 1 / request tile t + 5 (four tiles in flight in registers), half the rows each;
 *   wave 1   corr0's chain of its 64 rows over tile t; wave 2: corr1's (a step: LDS read, conversion, multiply, add -- the products
 *            and their order are the reference's; what lies at or behind n - 1 is not added);
 *   pass A   the chains of stage 0 on x -> coefficient 0; pass B: of stage 1 on y[s] = x[s] - mulshr5(x[s ? s - 1 : 0], c0), formed
 *            on the fly -> coefficient 1; pass C (all waves, 16 bytes per lane): z[s] = y[s] - mulshr5(y[s ? s - 1 : 0], c1) from
 *            x[s - 2 .. s] into xint -- what prep_general leaves there, by the same integer arithmetic (linne_utility.c:196-212).
 * Blocks beyond the list's end leave at once (16-bit material: all of them). */
#define PS_WAVES 4
__global__ __launch_bounds__(64 * PS_WAVES, 2) void k_prep_slow(Plan p)      /* (two blocks per CU: up to 512 blocks = 32 768 listed rows run at once; a chunk of the 8-channel stress case lists 22 500) */
{
    __shared__ __attribute__((aligned(16))) int32_t tile[2][64][68];      /* [tile mod 2][row][sample], rows 68 words apart: 16-byte accesses both ways -- the loader's (16 lanes of a row: 64 banks) and the chains' (lane = row: 16 lanes x 4 words hit banks 4 lane + j, all different) */
    __shared__ double xch[64];                                    /* corr1 on its way to the wave that holds corr0 */
    __shared__ int32_t cfs[2][64];                                /* the rows' coefficients */
    __shared__ uint32_t rowid[64], nlen[64];
    const uint32_t cnt = *p.prep_slow_n, base = blockIdx.x * 64u;
    if (base >= cnt) return;
    const uint32_t lane = threadIdx.x & 63u, S = p.S, C = p.C;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nv = (cnt - base < 64u) ? cnt - base : 64u;
    const uint32_t row = p.prep_slow_rows[base + (lane < nv ? lane : nv - 1u)];      /* f * C + ch of the chunk */
    const uint32_t f = row / C, ch = row - f * C;
    const uint32_t n = p.cls[p.cls_of_frame[f]].n;
    if (role == 0u) { rowid[lane] = row; nlen[lane] = n; }
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t a = (uint32_t)__shfl_xor((int)nmax, o); nmax = a > nmax ? a : nmax; }
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    const uint32_t ntiles = (nmax + 63u) / 64u;
    __syncthreads();
    /* the loaders (waves 0 and 3, half the rows each): instruction k of loader h takes rows 4 (8 h + k) + (lane >> 4), samples
     * 4 (lane & 15) .. + 3 of the tile */
    const bool loader = (role == 0u || role == 3u);
    const uint32_t rq = lane >> 4, i4 = 4u * (lane & 15u), r0l = (role == 3u) ? 32u : 0u;
    uint32_t roff[8];                                             /* 32-bit BYTE offsets from xtmp (one uniform base + an offset register per load: eight 64-bit pointers were spilled, and a reload in front of each load made it wait for the one before); the host sets prep_defer only for chunks whose xtmp is below 4 GB */
    const char *const xb = (const char *)p.xtmp;
    /* FOUR tiles on their way in registers: a tile is 1-2 us of chain work, and beside the previous group's Rice kernels and copies
     * (whole streams) a load takes several -- with one tile in flight the launch took 3.2-3.8 ms there instead of 0.7 */
    lnn_v4i pre[4][8];
    if (loader) {
#pragma unroll
        for (int k = 0; k < 8; k++) roff[k] = rowid[r0l + 4 * k + rq] * S * 4u;
    }
    auto issue = [&](uint32_t t, lnn_v4i (&pr)[8]) {
        const uint32_t s0 = t * 64u + i4, sb = 4u * ((s0 < S) ? s0 : 0u);      /* (S is a multiple of 4.  What a tile holds behind S >= n is never added: any readable address will do, and no load sits behind a branch) */
#pragma unroll
        for (int k = 0; k < 8; k++) pr[k] = *(const lnn_v4i *)(xb + (roff[k] + sb));
    };
    auto commit = [&](uint32_t t, const lnn_v4i (&pr)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) *(lnn_v4i *)&tile[t & 1u][r0l + 4 * k + rq][i4] = pr[k];
    };
    /* the chains are the launch's critical path -- a dependent add per step -- and the launch often runs beside the previous group's
     * Rice kernels (whole streams) or the other half's analysis: their steps go first */
    if (loader) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    int32_t first0 = 0, first1 = 0;                               /* the stages' first input samples (the decoder's initial state) */
    int32_t c0 = 0;
#pragma unroll 1
    for (uint32_t pass = 0; pass < 2u; pass++) {
        double acc = 0.0, prevd = 0.0;                            /* (index -1: a zero whose products add +0.0) */
        int32_t xprev = 0;
        if (loader) {
            /* the loaders' own walk over the tiles (the same barriers as the chains'): iteration t writes tile t + 1 into LDS and
             * requests tile t + 5 into the registers that held it; unrolled over the four register stages */
            uint32_t z = 0;
            asm volatile("" : "+s"(z));                            /* (a zero the compiler cannot see through: the first tiles' 40 addresses are the same in both passes, and it kept them -- 80 registers, spilled -- instead of adding an offset to a base) */
            if (ntiles) { issue(z, pre[0]); commit(0, pre[0]); }
            if (ntiles > 1u) issue(z + 1u, pre[1]);
            if (ntiles > 2u) issue(z + 2u, pre[2]);
            if (ntiles > 3u) issue(z + 3u, pre[3]);
            if (ntiles > 4u) issue(z + 4u, pre[0]);
            __syncthreads();
#pragma unroll 1
            for (uint32_t t4 = 0; t4 < ntiles; t4 += 4u) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t4 + (uint32_t)u;
                    if (t < ntiles) {
                        if (t + 1u < ntiles) { commit(t + 1u, pre[(u + 1) % 4]); if (t + 5u < ntiles) issue(t + 5u, pre[(u + 1) % 4]); }
                        __syncthreads();
                    }
                }
            }
        } else {
        __syncthreads();
#pragma unroll 1
        for (uint32_t t = 0; t < ntiles; t++) {
            {
                const int32_t *tl = tile[t & 1u][lane];
                if (t == 0u) { xprev = tl[0]; if (pass) first1 = (int32_t)((uint32_t)xprev - (uint32_t)mulshr5(xprev, c0)); else first0 = xprev; }
                /* no row ends inside this tile: the rows that have all of it run the plain steps, and so do the rows that ended before it --
                 * behind a row's end xtmp holds zeros, and from one sample behind it on they make the stage's input and its products +0.0,
                 * which leave a chain's bits alone.  The one tile a row's end n falls into (n / 64; the tile that STARTS at n when n is a
                 * multiple of 64: its first step would add the product of index n - 1) asks every step */
                const bool whole = !__any((n >> 6) == t);
                /* step g: the products of index g - 1 = (value before) x (value before | this value), added while g < n */
#define PS_STEP(SQ_, GUARD_, PASS_, V_, S_) { \
                    const int32_t v = (V_); \
                    int32_t in_ = v; \
                    if (PASS_) { in_ = (int32_t)((uint32_t)v - (uint32_t)mulshr5(xprev, c0)); if ((GUARD_) && 64u * t + (S_) >= n) in_ = 0; xprev = v; } \
                    const double vd = (double)in_; \
                    const double prod = (SQ_) ? prevd * prevd : prevd * vd; \
                    if (!(GUARD_) || 64u * t + (S_) < n) acc += prod; \
                    prevd = vd; }
                /* a whole tile: the row's 64 samples first (a read waited for inside the chain costs the wave its latency per step), then 64
                 * steps of straight-line code, eight at a time (the conversions do not depend on the chain: left alone the scheduler forms
                 * all 64 first, in 128 registers) */
#define PS_CHAIN(SQ_, PASS_) { lnn_v4i vv[16]; \
                _Pragma("unroll") for (int k = 0; k < 16; k++) vv[k] = *(const lnn_v4i *)(tl + 4 * k); \
                __builtin_amdgcn_sched_barrier(0); \
                _Pragma("unroll") for (uint32_t s = 0; s < 64u; s++) { PS_STEP(SQ_, false, PASS_, vv[s >> 2][s & 3u], s) if ((s & 7u) == 7u) __builtin_amdgcn_sched_barrier(0); } }
                /* a tile in which a row ends (a ragged last frame's, once per pass): step by step */
#define PS_TAIL(SQ_, PASS_) { _Pragma("unroll 1") for (uint32_t s = 0; s < 64u; s++) PS_STEP(SQ_, true, PASS_, tl[s], s) }
#define PS_CHAIN2(SQ_, PASS_) { if (whole) PS_CHAIN(SQ_, PASS_) else PS_TAIL(SQ_, PASS_) }
                if (role == 1u) { if (pass) PS_CHAIN2(true, true) else PS_CHAIN2(true, false) }
                else            { if (pass) PS_CHAIN2(false, true) else PS_CHAIN2(false, false) }
#undef PS_CHAIN2
#undef PS_TAIL
#undef PS_CHAIN
#undef PS_STEP
            }
            __syncthreads();
        }
        }
        /* the stage's coefficient (linne_utility.c:176-190), in the lanes of wave 1 */
        if (role == 2u) xch[lane] = acc;
        __syncthreads();
        if (role == 1u) {
            const double corr0 = acc;
            double corr1 = xch[lane];
            int32_t coef;
            corr1 /= corr0;
            if ((corr0 < 1e-6) || (corr1 < 0.0)) coef = 0;
            else { coef = (int32_t)round_away(corr1 * 32.0); if (coef >= 16) coef = 15; }
            cfs[pass][lane] = coef;
        }
        __syncthreads();
        c0 = cfs[0][lane];
    }
    if (role == 1u && lane < nv) {
        int32_t *rec = p.prm + ((size_t)p.frame_map[f] * C + ch) * LINNE_AMD_PARAM_WORDS;
        rec[LINNE_AMD_PRM_PREV + 0] = first0; rec[LINNE_AMD_PRM_PCOEF + 0] = cfs[0][lane];
        rec[LINNE_AMD_PRM_PREV + 1] = first1; rec[LINNE_AMD_PRM_PCOEF + 1] = cfs[1][lane];
    }
    /* pass C: both stages' filters, four samples per lane */
#pragma unroll 1
    for (uint32_t r = 0; r < nv; r++) {
        const uint32_t nr = nlen[r];
        const int32_t a0 = cfs[0][r], a1 = cfs[1][r];
        const int32_t *xin = p.xtmp + (size_t)rowid[r] * S;
        int32_t *out = p.xint + (size_t)rowid[r] * S;
        for (uint32_t g0 = 4u * threadIdx.x; g0 < S; g0 += 4u * 64u * PS_WAVES) {
            const lnn_v4i v = *(const lnn_v4i *)(xin + g0);
            int32_t xm1 = 0, xm2 = 0;
            if (g0) { xm1 = xin[g0 - 1u]; xm2 = xin[g0 - 2u]; }
            /* y[g] = x[g] - mulshr5(x[g ? g - 1 : 0], a0) for g < n, 0 beyond; z likewise from y */
            int32_t yprev = g0 ? (int32_t)((uint32_t)xm1 - (uint32_t)mulshr5(xm2, a0)) : 0;      /* y[g0 - 1] (g0 >= 4: its own predecessor is x[g0 - 2]) */
            if (g0 && g0 - 1u >= nr) yprev = 0;
            int32_t xp = xm1;
            lnn_v4i z;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t g = g0 + (uint32_t)e;
                const int32_t xg = v[e];
                int32_t y = (int32_t)((uint32_t)xg - (uint32_t)mulshr5(g ? xp : xg, a0));
                if (g >= nr) y = 0;
                int32_t zz = (int32_t)((uint32_t)y - (uint32_t)mulshr5(g ? yprev : y, a1));
                if (g >= nr) zz = 0;
                z[e] = zz; xp = xg; yprev = y;
            }
            *(lnn_v4i *)(out + g0) = z;
        }
    }
}
#undef PS_WAVES

/* block-type statistics (linne_encoder.c:494-503 -> lpc.c:810-848): SIN-window autocorrelation of the RAW channel at
 * order P0 = layer-0 size, then Levinson-Durbin.  One block per (frame, channel): all threads window a chunk of samples
 * and form the lag products into LDS, then lane `lag` adds its products in sample order -- one chain per lag, as in the
 * reference.  Independent of the analysis, so it runs on a side stream concurrently with it. */
#define STAT_THREADS 256
#define STAT_CHUNK   512
__global__ __launch_bounds__(STAT_THREADS) void k_stats(Plan p)
{
    __shared__ double sv[STAT_CHUNK + 8];
    __shared__ double sprod[5][STAT_CHUNK];
    __shared__ double sh_r[8];
    const uint32_t f = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const DevClass &c = p.cls[p.cls_of_frame[f]];
    const uint32_t n = c.n, S = p.S, C = p.C;
    const uint32_t fo = p.frame_map[f];
    const size_t xbase = ((size_t)fo * C + ch) * S;
    const uint32_t P0 = p.P[0];                              /* 2 or 4 */
    const double *sinw = p.sintab + c.sin_off;
    double r = 0.0;
    for (uint32_t base = 0; base < n; base += STAT_CHUNK) {
        __syncthreads();
        for (uint32_t i = tid; i < STAT_CHUNK + P0; i += STAT_THREADS) {
            const uint32_t g = base + i;
            sv[i] = (g < n) ? ((double)pcm_at(p, xbase + g) * p.scale) * sinw[g] : 0.0;
        }
        __syncthreads();
        for (uint32_t idx = tid; idx < (P0 + 1) * STAT_CHUNK; idx += STAT_THREADS) {
            const uint32_t lag = idx / STAT_CHUNK, i = idx % STAT_CHUNK;
            sprod[lag][i] = sv[i] * sv[i + lag];
        }
        __syncthreads();
        if (tid <= P0 && tid < n) {                          /* lag = tid: terms i < n - lag */
            const uint32_t lag = tid, total = n - lag;
            const uint32_t cnt = (total > base) ? ((total - base < STAT_CHUNK) ? (total - base) : STAT_CHUNK) : 0u;
            const double *q = sprod[lag];
            uint32_t i = 0;
            for (; i + 8 <= cnt; i += 8) {
                const double q0 = q[i], q1 = q[i + 1], q2 = q[i + 2], q3 = q[i + 3], q4 = q[i + 4], q5 = q[i + 5], q6 = q[i + 6], q7 = q[i + 7];
                r += q0; r += q1; r += q2; r += q3; r += q4; r += q5; r += q6; r += q7;
            }
            for (; i < cnt; i++) r += q[i];
        }
    }
    __syncthreads();
    if (tid <= P0) sh_r[tid] = r;
    __syncthreads();
    if (tid == 0) {
        double a[8], pc[8], rl[8];
        double *st = p.stats + ((size_t)fo * C + ch) * LINNE_AMD_STAT_WORDS;
        for (uint32_t i = 0; i <= P0; i++) rl[i] = sh_r[i];
        const double r0 = rl[0] * (1.0 + 0.0);
        const int zero = (n < P0) || (fabs(r0) < (double)FLT_EPSILON);
        for (uint32_t i = 0; i < 8; i++) pc[i] = 0.0;
        if (!zero) levinson(rl, r0, P0, a, pc);
        st[LINNE_AMD_ST_R0] = rl[0];
        st[LINNE_AMD_ST_K1 + 0] = pc[1]; st[LINNE_AMD_ST_K1 + 1] = pc[2]; st[LINNE_AMD_ST_K1 + 2] = pc[3];
        st[LINNE_AMD_ST_ZERO] = zero ? 1.0 : 0.0;
    }
}

/* k_stats_rows: the same statistics with lanes = channel-frames, for batches (k_stats keeps the small ones: it finishes a single
 * block sooner).  A block takes 64 channel-frames and P0 + 1 waves; all waves load: 16 samples of 64 rows per tile, coalesced
 * 16-byte (int16: 8-byte) loads, converted, scaled and SIN-windowed once, transposed into LDS (row stride 65).  Wave `lag` then
 * walks the positions: lane = channel-frame adds v[m - lag] * v[m] to its chain in sample order -- the reference's products in the
 * reference's order (lpc.c:215-249 after :188-195); the last `lag` values ride in registers (the position loop is unrolled over
 * the ring's turn).  Zeros stand in front of sample 0 and behind a frame's end: adding +-0.0 leaves a chain's bits unchanged.
 * 485 blocks serve the 60-minute stereo track where k_stats launches 31 008, each of which has five lanes adding while 251 wait:
 * 3.1 ms beside k_prep (and 2.6 ms of the step lost to the crowding) became a fraction of a millisecond. */
#define STATR_T 16                      /* positions per tile (64 was measured: slower -- 69 KB of LDS leave two blocks per CU and the kernel lives beside k_prep) */
template <int LAG, typename IssueNext, typename CommitNext>
__device__ __forceinline__ double stats_rows_chain(const double (*xt)[STATR_T][65], uint32_t lane, uint32_t ntiles, IssueNext &&issue_next, CommitNext &&commit_next)
{
    double r = 0.0, ring[LAG ? LAG : 1];
#pragma unroll
    for (int k = 0; k < (LAG ? LAG : 1); k++) ring[k] = 0.0;
    constexpr int TURN = LAG ? LAG : 1;                     /* the tile length (16) is a multiple of 1, 2 and 4: the ring index is a constant; 3 needs the modulo below */
#pragma unroll 1
    for (uint32_t ti = 0; ti < ntiles; ti++) {
        const uint32_t buf = ti & 1u;
        issue_next(ti);                                     /* the next tile's loads are requested now and land in LDS behind this tile's work */
        if (LAG == 3) {
#pragma unroll 1
            for (uint32_t m = 0; m < (uint32_t)STATR_T; m++) {
                const double v = xt[buf][m][lane];
                r += ring[0] * v;
                ring[0] = ring[1]; ring[1] = ring[2]; ring[2] = v;
            }
        } else {
#pragma unroll 1
            for (uint32_t m0 = 0; m0 < (uint32_t)STATR_T; m0 += 16u) {
#pragma unroll
                for (int m = 0; m < 16; m++) {              /* 16 is a multiple of the ring's turn: its index is a constant */
                    const double v = xt[buf][m0 + m][lane];
                    if (LAG == 0) r += v * v;
                    else { r += ring[m % TURN] * v; ring[m % TURN] = v; }
                }
            }
        }
        commit_next(ti);
        __syncthreads();
    }
    return r;
}

template <int NW, int FMT>           /* NW = P0 + 1 waves (3 or 5); FMT: how the PCM is staged -- 0 int32, 1 int16, 2 packed little-endian 3-byte samples.  The host launches it only for S % 4 == 0 (a piece of four samples is then 16 / 8 / 12 bytes at a 4-byte boundary) */
__global__ __launch_bounds__(64 * NW, 2) void k_stats_rows(Plan p)
{
    constexpr uint32_t NPIECE = 64u * (STATR_T / 4u), NTH = 64u * NW, NP = (NPIECE + NTH - 1u) / NTH;     /* loader pieces (a row's 4 consecutive samples) per thread and tile */
    __shared__ __attribute__((aligned(16))) double xt[2][STATR_T][65];
    __shared__ double rl[5][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t P0 = p.P[0], C = p.C, S = p.S, nrows = p.F * C, row0 = blockIdx.x * 64u;
    /* my row as a chain owner (lane) */
    uint32_t row = row0 + lane; const bool mine = row < nrows; if (!mine) row = nrows - 1u;
    const DevClass &c = p.cls[p.cls_of_frame[row / C]];
    const uint32_t my_n = mine ? c.n : 0u;
    uint32_t n_blk = my_n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)n_blk, o); n_blk = v > n_blk ? v : n_blk; }
    n_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_blk);
    const uint32_t ntiles = (n_blk + STATR_T - 1u) / STATR_T;
    /* my loader pieces: piece pc = row pc / 16, samples 4 (pc % 16) .. + 3 of the tile; everything that does not depend on the tile
     * is set up once.  A tile's loads are ALL issued before any is used (no branches between them: addresses are clamped, the
     * values masked afterwards) */
    size_t pbase[NP]; uint32_t pn[NP], plsm[NP], plrow[NP]; const double *psw[NP];
#pragma unroll
    for (uint32_t q = 0; q < NP; q++) {
        const uint32_t pc = tid + q * NTH, pcc = pc < NPIECE ? pc : 0u;
        plrow[q] = pcc / (STATR_T / 4u); plsm[q] = 4u * (pcc % (STATR_T / 4u));
        uint32_t r = row0 + plrow[q]; const bool have = (pc < NPIECE) && r < nrows; if (r >= nrows) r = nrows - 1u;
        const uint32_t fr = r / C, ch = r % C;
        const DevClass &lc = p.cls[p.cls_of_frame[fr]];
        pn[q] = have ? lc.n : 0u;
        pbase[q] = ((size_t)p.frame_map[fr] * C + ch) * S;
        psw[q] = p.sintab + lc.sin_off;
    }
    int4 raw[NP]; double sv[NP][4];
    auto issue = [&](uint32_t ti) {
#pragma unroll
        for (uint32_t q = 0; q < NP; q++) {
            const uint32_t s = ti * STATR_T + plsm[q], sc = (s + 3u < S) ? s : 0u;          /* (S % 4 == 0: a piece is inside the row or behind its end) */
            if (FMT == 1) { const short4 v = *(const short4 *)((const int16_t *)p.pcm + pbase[q] + sc); raw[q] = make_int4(v.x, v.y, v.z, v.w); }
            else if (FMT == 2) {                            /* twelve bytes = four samples, least significant byte first */
                const uint32_t *w = (const uint32_t *)((const uint8_t *)p.pcm + 3u * (pbase[q] + sc));
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
                raw[q] = make_int4((int32_t)(w0 << 8) >> 8, (int32_t)(((w0 >> 24) | (w1 << 8)) << 8) >> 8, (int32_t)(((w1 >> 16) | (w2 << 16)) << 8) >> 8, (int32_t)w2 >> 8);
            }
            else raw[q] = *(const int4 *)(p.pcm + pbase[q] + sc);
            const uint32_t last = pn[q] ? pn[q] - 1u : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) sv[q][j] = psw[q][(s + j < pn[q]) ? (s + j) : last];
        }
    };
    auto commit = [&](uint32_t ti, uint32_t buf) {
#pragma unroll
        for (uint32_t q = 0; q < NP; q++) {
            if (tid + q * NTH < NPIECE) {
                const uint32_t s = ti * STATR_T + plsm[q];
                const int32_t v[4] = { raw[q].x, raw[q].y, raw[q].z, raw[q].w };
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) xt[buf][plsm[q] + j][plrow[q]] = (s + j < pn[q]) ? ((double)v[j] * p.scale) * sv[q][j] : 0.0;   /* lpc.c:192: the window on the scaled sample */
            }
        }
    };
    if (ntiles == 0) return;
    issue(0); commit(0, 0);
    __syncthreads();
    auto issue_next = [&](uint32_t ti) { if (ti + 1u < ntiles) issue(ti + 1u); };
    auto commit_next = [&](uint32_t ti) { if (ti + 1u < ntiles) commit(ti + 1u, (ti + 1u) & 1u); };        /* (that buffer was last read in tile ti - 1, a barrier ago) */
    double r;
    switch (wave) {
    case 0: r = stats_rows_chain<0>(xt, lane, ntiles, issue_next, commit_next); break;
    case 1: r = stats_rows_chain<1>(xt, lane, ntiles, issue_next, commit_next); break;
    case 2: r = stats_rows_chain<2>(xt, lane, ntiles, issue_next, commit_next); break;
    case 3: r = stats_rows_chain<3>(xt, lane, ntiles, issue_next, commit_next); break;
    default: r = stats_rows_chain<4>(xt, lane, ntiles, issue_next, commit_next); break;
    }
    rl[wave][lane] = r;
    __syncthreads();
    if (wave == 0 && mine) {
        double a[8], pc[8], rr[8];
        const uint32_t fr = row / C, ch = row % C;
        double *st = p.stats + ((size_t)p.frame_map[fr] * C + ch) * LINNE_AMD_STAT_WORDS;
        for (uint32_t i = 0; i <= P0; i++) rr[i] = rl[i][lane];
        const double r0 = rr[0] * (1.0 + 0.0);
        const int zero = (my_n < P0) || (fabs(r0) < (double)FLT_EPSILON);
        for (uint32_t i = 0; i < 8; i++) pc[i] = 0.0;
        if (!zero) levinson(rr, r0, P0, a, pc);
        st[LINNE_AMD_ST_R0] = rr[0];
        st[LINNE_AMD_ST_K1 + 0] = pc[1]; st[LINNE_AMD_ST_K1 + 1] = pc[2]; st[LINNE_AMD_ST_K1 + 2] = pc[3];
        st[LINNE_AMD_ST_ZERO] = zero ? 1.0 : 0.0;
    }
}


#endif

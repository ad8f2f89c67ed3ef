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

__global__ __launch_bounds__(PREP_THREADS) void k_prep(Plan p)
{
    __shared__ int64_t sh[PREP_THREADS / 64];
    __shared__ int32_t sh_coef;
    __shared__ double sh_prod[2][PREP_CHUNK];
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


#endif

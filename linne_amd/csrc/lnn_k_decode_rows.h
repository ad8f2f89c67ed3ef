/* lnn_k_decode_rows.h -- the THROUGHPUT form of the synthesis: k_synth_rows (a layer, linne_lpc_synthesize.c:8-83, four channel-frames
 * per wave, one per 16-lane DPP row), k_synth_rows8 (the layers of <= 16 taps, eight per wave) and k_deemph_lr (de-emphasis + MS -> LR
 * behind layer 0).
 * Part of the single translation unit lnn_device.hip (included there behind lnn_k_decode.h); not a stand-alone header.
 *
 * The lanes = channel-frames kernels (k_synth_small / k_synth_big) are few, long waves: a pass costs one wave's 10 240-step
 * recurrence whatever the batch, and the long layer's 128 taps per step are FP64 multiply-adds on the vector unit.  Here the
 * recurrence is split by the distance d of a tap as in k_synth_pipe (y[t] = r[t] - ((half + sum_d c_d y[t - d]) >> rshift), blocks
 * of 16 outputs, lane i of a row = output i of the block):
 *   d <= i            the block's own outputs: lane i holds output i's sum; when y_j is final every later lane of the ROW adds its
 *                     tap times y_j -- the broadcast is a DPP row_newbcast, so the four rows run their own recurrences in the same
 *                     instructions (shift, subtract, broadcast, two 24-bit multiply-adds per step and four channel-frames);
 *   i < d <= i + 16   the previous block's outputs: layers of <= 16 taps add them in the same steps into the NEXT block's sums (second
 *                     coefficient set); longer layers leave them to the matrix unit with the older ones (one more MFMA at the block's start);
 *   d > i + 16        older samples (layers of more than 16 taps): v_mfma_i32_16x16x64_i8 with K = 4 channel-frames x 16 samples --
 *                     A = the outputs' signed base-256 digit planes (row 4 q + b = plane b of channel-frame q, non-zero only in
 *                     that channel-frame's K group), B = each channel-frame's Toeplitz slice of its 8-bit coefficients (held by
 *                     its own 16 lanes), one MFMA per 16 samples of history, issued a block ahead; C comes out with the four
 *                     planes of output i of channel-frame q in lane (q, i): recombined by shifts, modulo 2^32 like the reference.
 * Blocks are aligned to the frame (sample 16 m + i), not to the unit: a lane is `pred` (its sample is predicted from its unit's
 * coefficients) or not (the unit's first np samples, what lies behind the last unit, a layer linne_decoder.c skips), and a row's
 * coefficient registers are built for ONE state -- all 16 lanes predicting in unit u, or none (zeros: the steady code then copies).
 * A block in which a row's lanes disagree (a unit's boundary inside it) takes the generic routine for that row: tap by tap over the
 * row's lanes from the previous outputs (registers for d <= 16, the digit ring beyond), a DPP row sum per sample.  The layers of <= 16
 * taps, where a unit of fewer than 16 samples makes every unit's first block such a block, build the registers per LANE instead and
 * have no generic rows.  An outer loop iteration starts at a block in which some lane's class changes (every lane knows the next
 * such block of its own: m_event) and writes the registers -- there and nowhere else, so the inner loop keeps them in place.
 * Frames of any length and unit count take this kernel; the steady code is what a frame spends its time in.
 * The samples travel in 64-sample chunks: one 16-byte load and store per lane and chunk, staged through LDS. */
#ifndef LNN_K_DECODE_ROWS_H_INCLUDED
#define LNN_K_DECODE_ROWS_H_INCLUDED

template <int J> __device__ __forceinline__ int32_t row_bcast(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true); }   /* row_newbcast:J (every lane has a source: nothing is bound) */
__device__ __forceinline__ uint32_t row_sum_all(uint32_t v)      /* wrap-around sum over the 16 lanes of a row, in every lane of it */
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);   /* row_ror:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);   /* row_ror:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);   /* row_ror:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);   /* row_ror:1 */
    return v;
}

#define SR_STEPS(MUL_) \
    SR_STEP(0, MUL_) SR_STEP(1, MUL_) SR_STEP(2, MUL_) SR_STEP(3, MUL_) SR_STEP(4, MUL_) SR_STEP(5, MUL_) SR_STEP(6, MUL_) SR_STEP(7, MUL_) \
    SR_STEP(8, MUL_) SR_STEP(9, MUL_) SR_STEP(10, MUL_) SR_STEP(11, MUL_) SR_STEP(12, MUL_) SR_STEP(13, MUL_) SR_STEP(14, MUL_) SR_STEP(15, MUL_)

#define SR_PAD 160u      /* zeros in front of the staged coefficients: distances up to 128 + 16 + 15 beyond np */
#define SR_CST 320u      /* SR_PAD + 128 coefficients + 16 zeros (distances <= 0), rounded */
#define SR_RINGP 272      /* bytes from a digit plane to the next (256 used) */
template <int NCH, int PB = 16>      /* NCH: 16-sample chunks of older history on the matrix unit: 0 for layers of <= 16 taps, 1 for 32, 3 for 64, 7
                                       * for 128.  PB: the layer's order if it is below 16, else 16: output j of a block reaches the next block's sums
                                       * only from distance 16 + i - j <= PB, i.e. j >= 16 - PB -- the other steps leave that multiply-add out */
__global__ __launch_bounds__(64, 4) void k_synth_rows(DecPlan p, uint32_t layer)
{
    __shared__ __attribute__((aligned(16))) int8_t ring[4][4][SR_RINGP];     /* [channel-frame][digit plane][sample mod 256]; planes 272 bytes apart = 4 banks: the 16 planes' reads of a chunk (one lane each) and the rows' digit stores do not collide */
    __shared__ __attribute__((aligned(16))) int8_t zeros[256];               /* what the A operand's other K groups read */
    __shared__ __attribute__((aligned(16))) int32_t stg_in[2][4][80], stg_out[4][80];    /* (rows 16 words apart modulo the 64 banks: the four rows' reads of a block do not collide) */
    __shared__ __attribute__((aligned(16))) int8_t cst[4][SR_CST];           /* a row's coefficients while its registers are built */
    __shared__ int8_t call[4][16];                                           /* NCH = 0: the layer's coefficients of each row (P <= 16 of them) */
    const uint32_t lane = threadIdx.x, i = lane & 15u, q = lane >> 4, S = p.S;
    const uint32_t nrows = p.F * p.C;
    uint32_t cf = 4u * blockIdx.x + q;
    const bool have = cf < nrows;
    if (!have) cf = nrows - 1u;
    const uint32_t n = have ? p.nsmp[cf / p.C] : 0u;
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    int32_t *g = p.data + (size_t)cf * S;
    const uint32_t P = p.P[layer];
    const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + layer], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + layer];
    const uint32_t np = units ? P / units : 0u, ns = units ? n / units : 0u;
    const bool skip = (units == 0 || np == 0 || ns < np);                    /* linne_decoder.c: such a layer leaves the data unchanged */
    const uint32_t half = 1u << ((rs - 1u) & 31u), sh_ = rs & 31u;
    const int32_t *crec = rec + LINNE_AMD_PRM_COEF + p.coef_off[layer];
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)nmax, o); nmax = other > nmax ? other : nmax; }
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    const uint32_t nchunk = (nmax + 63u) / 64u;
    for (uint32_t k = lane; k < 256u; k += 64u) { zeros[k] = 0; }
    for (uint32_t k = lane; k < 4u * SR_CST / 4u; k += 64u) ((uint32_t *)&cst[0][0])[k] = 0u;
    if (!NCH) call[q][i] = (i < P && units) ? (int8_t)crec[i] : (int8_t)0;
    if (NCH) for (uint32_t k = lane; k < 16u * SR_RINGP / 4u; k += 64u) ((uint32_t *)&ring[0][0][0])[k] = 0u;
    /* the A operand of lane l: row l & 15 = plane b of channel-frame qa, K group l >> 4: its own channel-frame's, or zeros */
    const int8_t *abase = (((lane & 15u) >> 2) == q) ? &ring[(lane & 15u) >> 2][lane & 3u][0] : &zeros[0];

    /* place of my sample in its unit; the state my row's coefficient registers are built for */
    uint32_t unit = 0, m_event = 0u, half_l = 0u;               /* m_event: the first block in which my sample's class may be another */
    bool pred = false;
    int32_t ccA[16], ccB[16];
    lnn_v4i tz[NCH ? NCH : 1];
#pragma unroll
    for (int j = 0; j < 16; j++) { ccA[j] = 0; ccB[j] = 0; }
#pragma unroll
    for (int c = 0; c < (NCH ? NCH : 1); c++) tz[c] = lnn_v4i{ 0, 0, 0, 0 };
    int32_t yprev = 0;
    uint32_t nxt = 0;                                              /* layers of <= 16 taps: what the previous block adds (NCH > 0: the matrix unit's, with the old taps) */
    lnn_v4i wacc = { 0, 0, 0, 0 }, tzp = { 0, 0, 0, 0 };           /* NCH > 0: the digit planes of this block's matrix-unit sums so far (chunks 0 .. NCH-1); the Toeplitz slice of the PREVIOUS block's taps (distances 16 + i - e) */

    auto window = [&](uint32_t m) -> lnn_v4i {                  /* matrix-unit part of block m's sums without the block before it: the 16 NCH samples that end 16 before it (digit planes) */
        lnn_v4i acc4 = { 0, 0, 0, 0 };
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const lnn_v4i a = *(const lnn_v4i *)(abase + ((16u * m - 32u - 16u * (uint32_t)c) & 255u));
            acc4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, tz[c], acc4, 0, 0, 0);
        }
        return acc4;
    };

    /* 64-sample chunks: chunk c + 1 is requested while chunk c is worked on */
    auto fetch = [&](uint32_t c) -> lnn_v4i {
        lnn_v4i v = { 0, 0, 0, 0 };
        const uint32_t s0 = 64u * c + 4u * i;
        if (s0 + 3u < S) v = *(const lnn_v4i *)(g + s0);
        else { if (s0 < S) v[0] = g[s0]; if (s0 + 1u < S) v[1] = g[s0 + 1u]; if (s0 + 2u < S) v[2] = g[s0 + 2u]; }
        return v;
    };
    lnn_v4i pre = { 0, 0, 0, 0 };
    if (nchunk) { const lnn_v4i first = fetch(0); *(lnn_v4i *)&stg_in[0][q][4u * i] = first; }
    if (nchunk > 1u) pre = fetch(1);
    const uint32_t nblk = (nmax + 15u) / 16u;

    /* Outer loop: a block in which some lane's class changes -- the rows' classes, their coefficient registers (written HERE and
     * nowhere else, so the inner loop holds them in place), what the previous outputs add.  Inner loop: blocks while every lane
     * keeps its class. */
    uint32_t m = 0;
#pragma unroll 1
    while (m < nblk) {
        bool gen = false;
        {
            /* a lane's class changes here: every row's class in this block -- all lanes predicting in one unit, none
             * predicting, or mixed (the generic routine) -- and for how many blocks every lane keeps its own */
            {
                const uint32_t t = 16u * m + i;
                unit = skip ? units : t / (ns ? ns : 1u);
                uint32_t ahead = 0xFFFFFFFFu;               /* behind the last unit (or a skipped layer): copied to the end */
                pred = false;
                if (unit < units) {
                    const uint32_t tl = t - unit * ns;
                    pred = tl >= np;
                    ahead = ((pred ? ns : np) - tl + 15u) / 16u;
                }
                m_event = (ahead == 0xFFFFFFFFu) ? ahead : m + ahead;
            }
            const uint64_t bp = __ballot(pred);
            const uint32_t rb = (uint32_t)(bp >> (16u * q)) & 0xFFFFu;
            const uint32_t u0 = (uint32_t)row_bcast<0>((int32_t)unit);
            const uint64_t bu = __ballot(pred && unit != u0);
            const bool same_unit = ((uint32_t)(bu >> (16u * q)) & 0xFFFFu) == 0u;
            const bool all_pred = (rb == 0xFFFFu) && same_unit, all_pass = (rb == 0u);
            gen = !(all_pred || all_pass);
            if (!NCH) {
                /* layers of <= 16 taps (a unit of np < 16 samples' start is a mixed block every time): the registers are built per
                 * LANE from the layer's whole coefficient vector (unit u's at u np) -- every tap of a predicting lane lies in its own
                 * unit (d <= np <= its place in the unit), so mixed rows run the steady code too */
                gen = false;
                const uint32_t cb0 = unit * np + np - i;         /* tap of distance d of my unit: call[cb0 + i - d] */
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int32_t da = (int32_t)i - j; const uint32_t db = 16u + i - (uint32_t)j;
                    const bool va = pred && da >= 1 && (uint32_t)da <= np, vb = pred && db <= np;
                    ccA[j] = va ? (int32_t)call[q][va ? cb0 + (uint32_t)j : 0u] : 0;
                    ccB[j] = vb ? (int32_t)call[q][vb ? cb0 + (uint32_t)j - 16u : 0u] : 0;
                }
                half_l = pred ? half : 0u;
            } else
            {   /* the coefficient registers of every row for the state it is in now (rows that keep theirs get the same values
                 * again); a mixed row: zeros, no state */
                const bool bpred = pred && !gen;
                /* the unit's coefficients (8 bits by format), staged as bytes between two runs of zeros: tap of distance d =
                 * cu[np - d] (linne_lpc_synthesize.c:30) sits at SR_PAD + np - d, and every distance outside 1 .. np reads a zero */
                if (bpred) {
                    const int32_t *cu = crec + (size_t)unit * np;
                    int32_t cv[8];
#pragma unroll
                    for (int kk = 0; kk < 8; kk++) { const uint32_t k = i + 16u * (uint32_t)kk; cv[kk] = (k < np) ? cu[k] : 0; }
#pragma unroll
                    for (int kk = 0; kk < 8; kk++) { const uint32_t k = i + 16u * (uint32_t)kk; if (k < np) cst[q][SR_PAD + k] = (int8_t)cv[kk]; }
                }
                const int8_t *cb = &cst[q][0] + (bpred ? SR_PAD + np - i : SR_PAD - 17u);      /* (no state: zeros whatever the distance) */
#pragma unroll
                for (int j = 0; j < 16; j++) ccA[j] = cb[j];                                   /* distances i - j */
                {   /* distances 16 + i - e: the block before, element e -- as a slice for the matrix unit like the older ones */
                    uint32_t w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
                    for (int e = 0; e < 16; e++) w[e >> 2] |= ((uint32_t)cb[e - 16] & 0xFFu) << (8 * (e & 3));
                    tzp = lnn_v4i{ (int)w[0], (int)w[1], (int)w[2], (int)w[3] };
                }
                __builtin_amdgcn_sched_barrier(0);          /* (a group of reads at a time: the registers are the steady code's) */
#pragma unroll
                for (int cc = 0; cc < NCH; cc++) {
                    uint32_t w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
                    for (int e = 0; e < 16; e++)            /* element e of chunk cc lies d = 32 + 16 cc + i - e samples before output i */
                        w[e >> 2] |= ((uint32_t)cb[e - 32 - 16 * cc] & 0xFFu) << (8 * (e & 3));
                    tz[cc] = lnn_v4i{ (int)w[0], (int)w[1], (int)w[2], (int)w[3] };
                    __builtin_amdgcn_sched_barrier(0);
                }
                half_l = bpred ? half : 0u;
            }
            if (gen) m_event = m + 1u;                      /* a mixed row has no state: the next block builds */
            /* what the previous block adds to this one's sums, and the matrix-unit part, with the registers as they are now
             * (rows that kept their state get the values they had: the sums are associative) */
            if (!NCH) {
                nxt = 0;
#define SR_STEP(J, MUL_) if ((J) >= 16 - PB) { const int32_t sv = row_bcast<J>(yprev); nxt += sp_mul8(ccB[J], sv & 0xFFFF, sv >> 16); }
                SR_STEPS(0)
#undef SR_STEP
            } else wacc = window(m);
        }
#pragma unroll 1
        do {
            const uint32_t c = m >> 2, k = m & 3u;
            if (k == 0u && c + 1u < nchunk) { *(lnn_v4i *)&stg_in[(c + 1u) & 1u][q][4u * i] = pre; if (c + 2u < nchunk) pre = fetch(c + 2u); }
            const int32_t res = stg_in[c & 1u][q][16u * k + i];
            /* NCH > 0: the block before this one joins the sums on the matrix unit too (its digits are in the ring since it ended; the same 16
             * bytes are chunk 0 of the NEXT block's window): the steps carry no multiply-add for it */
            lnn_v4i wa = { 0, 0, 0, 0 };
            uint32_t mpart = 0;
            if (NCH) {
                wa = *(const lnn_v4i *)(abase + ((16u * m - 16u) & 255u));
                wacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, tzp, wacc, 0, 0, 0);
                mpart = (uint32_t)wacc[0] + ((uint32_t)wacc[1] << 8) + ((uint32_t)wacc[2] << 16) + ((uint32_t)wacc[3] << 24);
            }
            const uint32_t acc0 = half_l + mpart + nxt;
            uint32_t acc = acc0;
            nxt = 0;
            /* speculation as in k_synth_pipe: every output of the block fits 24 bits -- one full-rate multiply-add per sum.  The
             * NEXT block's matrix-unit part rides between the steps, a chunk every two of them: each MFMA waits for the one before
             * it (one accumulator), and a wave that issued the seven back to back stood still for their latencies */
            lnn_v4i acc4 = { 0, 0, 0, 0 };
#define SR_STEP(J, MUL_) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); const int32_t sv = row_bcast<J>(y); \
                acc += (uint32_t)__mul24(ccA[J], sv); if (!NCH && (J) >= 16 - PB) asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(nxt) : "v"(ccB[J]), "v"(sv)); }     /* (a multiply-add per tap: the compiler's tree of products and three-operand adds is half as many again) */
#define SR_WIN(CC_) if ((CC_) < NCH) { const lnn_v4i a_ = wa; \
                if ((CC_) + 1 < NCH) wa = *(const lnn_v4i *)(abase + ((16u * m - 16u - 16u * (uint32_t)((CC_) + 1)) & 255u)); \
                acc4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_, tz[(CC_) < NCH ? (CC_) : 0], acc4, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            SR_STEP(0, 0) SR_STEP(1, 0) SR_WIN(0) SR_STEP(2, 0) SR_STEP(3, 0) SR_WIN(1) SR_STEP(4, 0) SR_STEP(5, 0) SR_WIN(2) SR_STEP(6, 0) SR_STEP(7, 0) SR_WIN(3)
            SR_STEP(8, 0) SR_STEP(9, 0) SR_WIN(4) SR_STEP(10, 0) SR_STEP(11, 0) SR_WIN(5) SR_STEP(12, 0) SR_STEP(13, 0) SR_WIN(6) SR_STEP(14, 0) SR_STEP(15, 0)
#undef SR_WIN
#undef SR_STEP
            int32_t yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
            const bool fits = gen || (((int32_t)((uint32_t)yout << 8) >> 8) == yout);
            if (!__all(fits)) {
                acc = acc0; nxt = 0;
#define SR_STEP(J, MUL_) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); const int32_t sv = row_bcast<J>(y); \
                const int32_t sl = sv & 0xFFFF, shh = sv >> 16; acc += sp_mul8(ccA[J], sl, shh); if (!NCH && (J) >= 16 - PB) nxt += sp_mul8(ccB[J], sl, shh); }
                SR_STEPS(0)
#undef SR_STEP
                yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
            }
            if (NCH && __any(gen)) {                             /* (the layers of <= 16 taps have no mixed rows: registers per lane) */
                /* the generic routine, sample by sample: the row's lanes share the taps (lane k: distances k + 1, k + 17, ...) */
                int32_t ycur = 0;
                const uint32_t rowbase = 16u * q;
#pragma unroll 1
                for (uint32_t j = 0; j < 16u; j++) {
                    const uint32_t src = (rowbase + j) * 4u;
                    const int32_t resj = __builtin_amdgcn_ds_bpermute((int)src, res);
                    const uint32_t pj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)pred);
                    const uint32_t uj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)unit);
                    const int32_t *cu = crec + (size_t)uj * np;
                    uint32_t sum = 0;
                    {
                        const uint32_t d = i + 1u;
                        const int32_t hsel = (i < j) ? ycur : yprev;
                        const int32_t val = __builtin_amdgcn_ds_bpermute((int)((rowbase + ((j - d) & 15u)) * 4u), hsel);
                        if (gen && pj && d <= np) sum = (uint32_t)cu[np - d] * (uint32_t)val;
                    }
                    if (NCH) {
                        for (uint32_t d = i + 17u; d <= (uint32_t)(16 * NCH + 16); d += 16u) {
                            const uint32_t ix = (16u * m + j - d) & 255u;
                            const uint32_t val = (uint32_t)(int32_t)ring[q][0][ix] + ((uint32_t)(int32_t)ring[q][1][ix] << 8)
                                               + ((uint32_t)(int32_t)ring[q][2][ix] << 16) + ((uint32_t)(int32_t)ring[q][3][ix] << 24);
                            if (gen && pj && d <= np) sum += (uint32_t)cu[np - d] * val;
                        }
                    }
                    sum = row_sum_all(sum);
                    const int32_t y = pj ? (int32_t)((uint32_t)resj - (uint32_t)((int32_t)(half + sum) >> sh_)) : resj;
                    if (i == j) ycur = y;
                }
                if (gen) yout = ycur;
            }
            stg_out[q][16u * k + i] = yout;
            if (NCH) {
                const uint32_t dg = sp_digits(yout), ix = (16u * m + i) & 255u;
                ring[q][0][ix] = (int8_t)dg; ring[q][1][ix] = (int8_t)(dg >> 8); ring[q][2][ix] = (int8_t)(dg >> 16); ring[q][3][ix] = (int8_t)(dg >> 24);
            }
            yprev = yout;
            wacc = acc4;
            m++;
            if (k == 3u || m == nblk) {
                /* the chunk's outputs: 16 bytes per lane; nothing behind a frame's end is written */
                const uint32_t s0 = 64u * c + 4u * i;
                const lnn_v4i v = *(const lnn_v4i *)&stg_out[q][4u * i];
                if (have) {
                    if (s0 + 3u < n) *(lnn_v4i *)(g + s0) = v;
                    else { if (s0 < n) g[s0] = v[0]; if (s0 + 1u < n) g[s0 + 1u] = v[1]; if (s0 + 2u < n) g[s0 + 2u] = v[2]; }
                }
            }
        } while (m < nblk && __all(m < m_event));
    }
}

/* k_synth_rows8<PB>: k_synth_rows for the layers of <= 16 taps with EIGHT channel-frames per wave, one per half of a DPP row, in
 * blocks of 8 outputs (lane i of a half = sample 8 m + i): a step's five to seven instructions -- shift, subtract, TWO broadcasts
 * (row_newbcast:j into the whole row, row_newbcast:8+j over its upper half: bank mask 0xc), the multiply-add into this block's sum,
 * the ones into the next block's (distances 8 + i - j) and, for orders above 8, the block after that (16 + i - j) -- serve eight
 * channel-frames, not four.  No matrix unit, no generic routine: every lane's coefficient registers are built from its own (unit,
 * predicting or not), as in k_synth_rows<0>.  PB = 4 / 8 / 16 bounds the layer's order: taps that cannot exist are left out. */
template <int J> __device__ __forceinline__ int32_t half_bcast(int32_t v)
{
    const int32_t lo = __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true);         /* lane J of the row -> all of it */
    return __builtin_amdgcn_update_dpp(lo, v, 0x150 + 8 + J, 0xf, 0xc, false);               /* lane 8 + J -> lanes 8 .. 15 */
}
#define SR8_STEPS SR8_STEP(0) SR8_STEP(1) SR8_STEP(2) SR8_STEP(3) SR8_STEP(4) SR8_STEP(5) SR8_STEP(6) SR8_STEP(7)
template <int PB>
__global__ __launch_bounds__(64, 4) void k_synth_rows8(DecPlan p, uint32_t layer)
{
    constexpr bool FAR = PB > 8;                                   /* taps beyond distance 8 + i: the block before the previous one */
    constexpr int JB = (PB >= 8) ? 0 : 8 - PB;                    /* first step whose output reaches the next block (8 + i - j <= PB) */
    __shared__ __attribute__((aligned(16))) int32_t stg_in[2][8][72], stg_out[8][72];    /* (rows 8 words apart modulo the 64 banks) */
    __shared__ int8_t call[8][16];                                 /* the layer's coefficients of each channel-frame (unit u's at u np) */
    const uint32_t lane = threadIdx.x, i = lane & 7u, q = lane >> 3, S = p.S;
    const uint32_t nrows = p.F * p.C;
    uint32_t cf = 8u * blockIdx.x + q;
    const bool have = cf < nrows;
    if (!have) cf = nrows - 1u;
    const uint32_t n = have ? p.nsmp[cf / p.C] : 0u;
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    int32_t *g = p.data + (size_t)cf * S;
    const uint32_t P = p.P[layer];
    const uint32_t units = (uint32_t)rec[LINNE_AMD_PRM_UNITS + layer], rs = (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + layer];
    const uint32_t np = units ? P / units : 0u, ns = units ? n / units : 0u;
    const bool skip = (units == 0 || np == 0 || ns < np);          /* linne_decoder.c: such a layer leaves the data unchanged */
    const uint32_t half = 1u << ((rs - 1u) & 31u), sh_ = rs & 31u;
    const int32_t *crec = rec + LINNE_AMD_PRM_COEF + p.coef_off[layer];
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)nmax, o); nmax = other > nmax ? other : nmax; }
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    const uint32_t nchunk = (nmax + 63u) / 64u, nblk = (nmax + 7u) / 8u;
    call[q][i] = (i < P && units) ? (int8_t)crec[i] : (int8_t)0;
    call[q][i + 8u] = (i + 8u < P && units) ? (int8_t)crec[i + 8u] : (int8_t)0;

    uint32_t unit = 0, m_event = 0u, half_l = 0u;
    bool pred = false;
    int32_t ccA[8], ccB[8], ccC[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { ccA[j] = 0; ccB[j] = 0; ccC[j] = 0; }
    int32_t yprev = 0, yprev2 = 0;
    uint32_t accB = 0, accC = 0, ncp = 0;                          /* what the previous block / the one before it add to this block; what the previous block adds to the next */

    /* 64-sample chunks, 32 bytes per lane: chunk c + 1 is requested while chunk c is worked on */
    auto fetch = [&](uint32_t c, lnn_v4i &v0, lnn_v4i &v1) {
        const uint32_t s0 = 64u * c + 8u * i;
        v0 = lnn_v4i{ 0, 0, 0, 0 }; v1 = lnn_v4i{ 0, 0, 0, 0 };
        if (s0 + 3u < S) v0 = *(const lnn_v4i *)(g + s0);           /* (S is a multiple of 4: a group of four is whole or absent) */
        if (s0 + 7u < S) v1 = *(const lnn_v4i *)(g + s0 + 4u);
    };
    lnn_v4i pre0 = { 0, 0, 0, 0 }, pre1 = { 0, 0, 0, 0 };
    if (nchunk) { fetch(0, pre0, pre1); *(lnn_v4i *)&stg_in[0][q][8u * i] = pre0; *(lnn_v4i *)&stg_in[0][q][8u * i + 4u] = pre1; }
    if (nchunk > 1u) fetch(1, pre0, pre1);

    uint32_t m = 0;
#pragma unroll 1
    while (m < nblk) {
        {   /* a lane's class changes here: every lane's class in this block, for how many blocks it keeps it, its registers */
            const uint32_t t = 8u * m + i;
            unit = skip ? units : t / (ns ? ns : 1u);
            uint32_t ahead = 0xFFFFFFFFu;                          /* behind the last unit (or a skipped layer): copied to the end */
            pred = false;
            if (unit < units) {
                const uint32_t tl = t - unit * ns;
                pred = tl >= np;
                ahead = ((pred ? ns : np) - tl + 7u) / 8u;
            }
            m_event = (ahead == 0xFFFFFFFFu) ? ahead : m + ahead;
            const uint32_t cb0 = unit * np + np - i;               /* tap of distance d of my unit: call[cb0 + i - d]; all of them lie in the unit (d <= np <= my place in it) */
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int32_t da = (int32_t)i - j; const uint32_t db = 8u + i - (uint32_t)j, dc = 16u + i - (uint32_t)j;
                const bool va = pred && da >= 1 && (uint32_t)da <= np, vb = pred && db <= np, vc = FAR && pred && dc <= np;
                ccA[j] = va ? (int32_t)call[q][va ? cb0 + (uint32_t)j : 0u] : 0;
                ccB[j] = vb ? (int32_t)call[q][vb ? cb0 + (uint32_t)j - 8u : 0u] : 0;
                ccC[j] = vc ? (int32_t)call[q][vc ? cb0 + (uint32_t)j - 16u : 0u] : 0;
            }
            half_l = pred ? half : 0u;
            /* what the blocks before add, with the registers as they are now (lanes that kept their class get the values they had) */
            accB = 0; accC = 0; ncp = 0;
#define SR8_STEP(J) { const int32_t sv = half_bcast<J>(yprev); const int32_t sl = sv & 0xFFFF, shh = sv >> 16; \
                if ((J) >= JB) accB += sp_mul8(ccB[J], sl, shh); if (FAR) ncp += sp_mul8(ccC[J], sl, shh); \
                if (FAR) { const int32_t s2 = half_bcast<J>(yprev2); accC += sp_mul8(ccC[J], s2 & 0xFFFF, s2 >> 16); } }
            SR8_STEPS
#undef SR8_STEP
        }
#pragma unroll 1
        do {
            const uint32_t c = m >> 3, k = m & 7u;
            if (k == 0u && c + 1u < nchunk) {
                *(lnn_v4i *)&stg_in[(c + 1u) & 1u][q][8u * i] = pre0; *(lnn_v4i *)&stg_in[(c + 1u) & 1u][q][8u * i + 4u] = pre1;
                if (c + 2u < nchunk) fetch(c + 2u, pre0, pre1);
            }
            const int32_t res = stg_in[c & 1u][q][8u * k + i];
            const uint32_t acc0 = half_l + accB + accC;
            uint32_t acc = acc0, nb = 0, nc = 0;
            /* speculation as in k_synth_rows: every output of the block fits 24 bits */
#define SR8_STEP(J) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); const int32_t sv = half_bcast<J>(y); \
                acc += (uint32_t)__mul24(ccA[J], sv); \
                if ((J) >= JB) asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(nb) : "v"(ccB[J]), "v"(sv)); \
                if (FAR) asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(nc) : "v"(ccC[J]), "v"(sv)); }
            SR8_STEPS
#undef SR8_STEP
            int32_t yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
            const bool fits = (((int32_t)((uint32_t)yout << 8) >> 8) == yout);
            if (!__all(fits)) {
                acc = acc0; nb = 0; nc = 0;
#define SR8_STEP(J) { const int32_t y = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_)); const int32_t sv = half_bcast<J>(y); \
                const int32_t sl = sv & 0xFFFF, shh = sv >> 16; acc += sp_mul8(ccA[J], sl, shh); \
                if ((J) >= JB) nb += sp_mul8(ccB[J], sl, shh); if (FAR) nc += sp_mul8(ccC[J], sl, shh); }
                SR8_STEPS
#undef SR8_STEP
                yout = (int32_t)((uint32_t)res - (uint32_t)((int32_t)acc >> sh_));
            }
            stg_out[q][8u * k + i] = yout;
            yprev2 = yprev; yprev = yout;
            accB = nb; accC = ncp; ncp = nc;
            m++;
            if (k == 7u || m == nblk) {
                /* the chunk's outputs: 32 bytes per lane; nothing behind a frame's end is written */
                const uint32_t s0 = 64u * c + 8u * i;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t sh0 = s0 + 4u * (uint32_t)h;
                    const lnn_v4i v = *(const lnn_v4i *)&stg_out[q][8u * i + 4u * (uint32_t)h];
                    if (have) {
                        if (sh0 + 3u < n) *(lnn_v4i *)(g + sh0) = v;
                        else { if (sh0 < n) g[sh0] = v[0]; if (sh0 + 1u < n) g[sh0 + 1u] = v[1]; if (sh0 + 2u < n) g[sh0 + 2u] = v[2]; }
                    }
                }
            }
        } while (m < nblk && __all(m < m_event));
    }
}
#undef SR8_STEPS

/* k_deemph_lr: what follows layer 0 when k_synth_rows took it -- the two-stage de-emphasis (linne_utility.c:215-241), a scalar
 * recurrence per channel-frame, with lanes = channel-frames, and MS -> LR (linne_utility.c:135-147) on the way out when the frames
 * of a block of 64 rows are whole (FUSE_MS: C a power of two <= 64; k_ms_to_lr otherwise).  Tiles of 64 rows x 64 samples go through
 * LDS; a load or store instruction moves 16 bytes per lane = 256 bytes of four rows (16 instructions per tile and direction).
 * A wave alone on its SIMD issues an instruction every ~9 cycles, so a pass costs what ONE wave executes per tile: the block is four
 * waves with a role each -- wave 0 requests tile t + 2 and writes tile t + 1 into LDS, wave 1 runs the recurrences over tile t (the
 * only serial part: 64 steps of eight instructions), waves 2 and 3 turn tile t - 1 into left / right and store it, half the rows
 * each -- three tile buffers, a barrier per tile. */
template <bool FUSE_MS>
#define DL_STORERS 2       /* waves that store (2 + DL_STORERS waves per block) */
__global__ __launch_bounds__(64 * (2 + DL_STORERS)) void k_deemph_lr(DecPlan p)
{
    __shared__ int32_t tile[3][64][65];                           /* [tile mod 3][row][sample]: bank = row + sample, no conflicts either way */
    const uint32_t lane = threadIdx.x & 63u, row0 = blockIdx.x * 64u, S = p.S, C = p.C;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nrows = p.F * C;
    uint32_t cf = row0 + lane;
    if (cf >= nrows) cf = nrows - 1u;
    const uint32_t n = p.nsmp[cf / C];
    const int32_t *rec = p.prm + (size_t)cf * LINNE_AMD_PARAM_WORDS;
    const int32_t c0e = rec[LINNE_AMD_PRM_PCOEF + 0], c1e = rec[LINNE_AMD_PRM_PCOEF + 1];
    int32_t zp = rec[LINNE_AMD_PRM_PREV + 1], yp = rec[LINNE_AMD_PRM_PREV + 0];
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)nmax, o); nmax = other > nmax ? other : nmax; }
    const uint32_t ntiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)((nmax + 63u) / 64u));
    /* instruction k of a tile: rows 4 k + (lane >> 4), samples 4 (lane & 15) .. + 3; rows behind the last one read the last one again */
    const uint32_t rq = lane >> 4, i4 = 4u * (lane & 15u);
    const uint32_t nv = (nrows - row0 < 64u) ? nrows - row0 : 64u;
    int32_t *blk = p.data + (size_t)row0 * S;
    lnn_v4i pre[16];
    auto issue = [&](uint32_t t) {
        const uint32_t s0 = t * 64u + i4;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t r = 4u * (uint32_t)k + rq, rr = (r < nv) ? r : nv - 1u;
            pre[k] = (s0 < S) ? *(const lnn_v4i *)(blk + rr * S + s0) : lnn_v4i{ 0, 0, 0, 0 };      /* (S is a multiple of 4) */
        }
    };
    auto commit = [&](uint32_t t) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            int32_t *w = &tile[t % 3u][4 * k + rq][i4];
            w[0] = pre[k][0]; w[1] = pre[k][1]; w[2] = pre[k][2]; w[3] = pre[k][3];
        }
    };
    if (role == 0u && ntiles) { issue(0); commit(0); if (ntiles > 1u) issue(1); }
    __syncthreads();
    /* iteration t: tile t + 1 goes into LDS, tile t is de-emphasised, tile t - 1 leaves */
    for (uint32_t t = 0; t < ntiles + 1u; t++) {
        if (role == 0u) {
            if (t + 1u < ntiles) { commit(t + 1u); if (t + 2u < ntiles) issue(t + 2u); }
        } else if (role == 1u) {
            if (t < ntiles) {
                int32_t (*tl)[65] = tile[t % 3u];
#pragma unroll 16
                for (uint32_t s = 0; s < 64u; s++) {              /* (behind a frame's end the state runs on: nothing of it is stored) */
                    const int32_t z = (int32_t)((uint32_t)tl[lane][s] + (uint32_t)mulshr5(zp, c1e));
                    const int32_t y = (int32_t)((uint32_t)z + (uint32_t)mulshr5(yp, c0e));
                    zp = z; yp = y;
                    tl[lane][s] = y;
                }
            }
        } else if (t >= 1u) {                                      /* (DL_STORERS waves: their share of the sixteen row groups each) */
            const uint32_t to = t - 1u, s0 = to * 64u + i4;
            int32_t (*tl)[65] = tile[to % 3u];
#pragma unroll
            for (int kk = 0; kk < 16 / DL_STORERS; kk++) {
                const uint32_t r = 4u * ((uint32_t)kk + (uint32_t)(16 / DL_STORERS) * (role - 2u)) + rq;
                const int32_t *w = &tl[r][i4];
                lnn_v4i v = { w[0], w[1], w[2], w[3] };
                if (FUSE_MS) {                                    /* row0 is a multiple of C: channels 0 and 1 of a frame are neighbouring rows of this block */
                    const uint32_t ch = r & (C - 1u);
                    if (ch < 2u) {
                        const int32_t *o = &tl[r ^ 1u][i4];       /* (C >= 2: rows r and r ^ 1 are the frame's channels 0 and 1) */
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t m_ = (uint32_t)(ch ? o[j] : v[j]), sd = (uint32_t)(ch ? v[j] : o[j]);
                            const uint32_t l = m_ - (uint32_t)((int32_t)sd >> 1);
                            v[j] = (int32_t)(ch ? sd + l : l);
                        }
                    }
                }
                const uint32_t nr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r & 63u) * 4u), (int)n);      /* row r's length: lane r holds it */
                if (r < nv) {
                    int32_t *gp = blk + r * S + s0;
                    if (s0 + 3u < nr) *(lnn_v4i *)gp = v;
                    else { if (s0 < nr) gp[0] = v[0]; if (s0 + 1u < nr) gp[1] = v[1]; if (s0 + 2u < nr) gp[2] = v[2]; }
                }
            }
        }
        __syncthreads();
    }
}
#undef SR_STEPS
#undef SR_PAD
#undef SR_CST
#undef SR_RINGP

#endif

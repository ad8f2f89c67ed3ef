/* lnn_k_rice.h -- k_rice_plan: partition means, Rice parameters and partition-order search.
 * Part of the single translation unit lnn_device.hip (included there, in this order); not a stand-alone header. */
#ifndef LNN_K_RICE_H_INCLUDED
#define LNN_K_RICE_H_INCLUDED

/* ================================================================================================
 * Rice planning (linne_coder.c:217-279): one block per channel-frame.
 * Integer sums are exact, the means repeat the host's divisions ((double)sum / ns, then pairwise (a + b) / 2.0), the
 * parameter is a search in the table of steps the host located with its libm (a mean inside a guard band raises the
 * record's flag and the host searches that channel-frame itself), code lengths are uint32 with wrap-around.
 * ============================================================================================== */
#define RICE_THREADS 256
#define REMIT_LDS_SAMPLES 12288u            /* frames up to this length are staged in LDS (48 KB + padding) by k_rice_plan / k_rice_emit */
struct RicePlanArgs {
    const int32_t *resid; const uint32_t *nsmp; uint8_t *plan; uint32_t C, S, nsteps;
    double steps[32];
};
__device__ __forceinline__ uint32_t rp_zz(int32_t v) { const uint32_t d = (uint32_t)v << 1; return (v < 0) ? ((0u - d) - 1u) : d; }
__device__ __forceinline__ uint32_t rp_wave_sum(uint32_t v) { for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64); return v; }    /* every lane gets the wave's total */
__device__ __forceinline__ uint32_t rp_gamma_len(uint32_t u) { return u ? (2u * (32u - (uint32_t)__clz((int)(u + 1u))) - 1u) : 1u; }   /* 2*ceil_log2(u+2)-1 */

/* LDS: a thread walks the nsf consecutive samples of its finest partition; straight from global memory that is a 4 * nsf byte
 * stride between lanes (every load instruction touches 64 cache lines), so frames up to REMIT_LDS_SAMPLES are first brought into
 * LDS with coalesced loads, zig-zagged on the way; run r starts at r * (nsf + 1) so that the lanes' strided reads spread over
 * the banks */
template <bool LDS> __global__ __launch_bounds__(RICE_THREADS) void k_rice_plan(RicePlanArgs a)
{
    extern __shared__ uint32_t zbuf[];
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
    if (LDS) { for (uint32_t s = tid; s < n; s += RICE_THREADS) zbuf[s + s / nsf] = rp_zz(x[s]); __syncthreads(); }
    for (uint32_t p = tid; p < parts; p += RICE_THREADS) {
        const int32_t *q = x + (size_t)p * nsf;
        const uint32_t *zq = zbuf + (size_t)p * (nsf + 1u);
        uint64_t sum = 0;
        for (uint32_t j = 0; j < nsf; j++) sum += LDS ? zq[j] : rp_zz(q[j]);
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
    /* The totals per order are sums of uint32 with wrap-around: their order is free, so a wave adds up its lanes' shares across
     * the lanes and issues ONE LDS atomic per order (64 lanes adding to the same word one by one was most of this kernel's time).
     * Per entry: the samples' fixed part and the parameter's own code.  Entry e = e1 - 1 belongs to order floor(log2 e1): the 64
     * entries of an aligned group of e1 >= 64 share their order. */
    for (uint32_t base = 0; base <= nent; base += RICE_THREADS) {       /* (every lane takes every turn: the wave sum needs them all) */
        const uint32_t e1 = base + tid;
        uint32_t bits = 0, o = 31u - (uint32_t)__clz((int)(e1 | 1u));
        if (e1 != 0u && e1 <= nent) {
            const uint32_t e = e1 - 1u;
            const uint32_t p = e - ((1u << o) - 1u), k = kk[e];
            bits = (n >> o) * (k + 2u) + (p ? rp_gamma_len(rp_zz((int32_t)k - (int32_t)kk[e - 1])) : 5u);
        }
        if ((e1 & ~63u) == 0u) { if (e1 != 0u && e1 <= nent) atomicAdd(&tot[o], bits); }      /* (wave-uniform branch: e1 - lane is a multiple of 64) */
        else {
            const uint32_t s = rp_wave_sum(bits);
            if ((tid & 63u) == 0u) atomicAdd(&tot[o], s);
        }
    }
    /* per finest partition: the excess of its samples under the parameter of each order's enclosing partition */
    {
        uint32_t acc[11];
#pragma unroll
        for (uint32_t o = 0; o < 11; o++) acc[o] = 0;
        for (uint32_t p = tid; p < parts; p += RICE_THREADS) {
            const int32_t *q = x + (size_t)p * nsf;
            const uint32_t *zq = zbuf + (size_t)p * (nsf + 1u);
            uint32_t kc[11];
#pragma unroll
            for (uint32_t o = 0; o < 11; o++) kc[o] = (o <= max_order) ? kk[((1u << o) - 1u) + (p >> (max_order - o))] : 0u;
            for (uint32_t j = 0; j < nsf; j++) {
                const uint32_t v = LDS ? zq[j] : rp_zz(q[j]);
#pragma unroll
                for (uint32_t o = 0; o < 11; o++) { const uint32_t k1pow = 1u << ((kc[o] + 1u) & 31u); acc[o] += ((v > k1pow) ? (v - k1pow) : 0u) >> kc[o]; }
            }
        }
#pragma unroll
        for (uint32_t o = 0; o < 11; o++)
            if (o <= max_order) { const uint32_t s = rp_wave_sum(acc[o]); if ((tid & 63u) == 0u) atomicAdd(&tot[o], s); }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0, min_bits = 0xFFFFFFFFu;
        for (uint32_t o = 0; o <= max_order; o++) if (min_bits > tot[o]) { min_bits = tot[o]; best = o; }
        best_s = best;
        rec[0] = (uint8_t)best; rec[1] = (uint8_t)flag;
        /* length of this channel's code in bits: the order field + what the search counted for the winning order
         * (k_rice_scan / k_rice_emit lay the code out from it) */
        *(uint32_t *)(rec + LINNE_AMD_RICE_PLAN_NBITS) = flag ? 0xFFFFFFFFu : (10u + min_bits);
    }
    __syncthreads();
    const uint32_t best = best_s;
    for (uint32_t p = tid; p < (1u << best); p += RICE_THREADS) rec[LINNE_AMD_RICE_PLAN_K2 + p] = kk[((1u << best) - 1u) + p];
}


/* ================================================================================================
 * Rice EMISSION on the device (linne_coder.c:281-302, bit_stream.h:240-282).  With the plan known, every
 * sample's code and its length follow from the sample alone, so the bit position of every code is an exclusive scan of the
 * lengths: k_rice_scan places the channel-frames of a batch back to back (8-byte aligned) in one buffer, k_rice_emit
 * writes each channel's code there exactly as the host's rice_emit would -- 10-bit order, per partition its parameter (5 bits,
 * then gamma-coded steps) and its samples' recursive Rice codes, MSB first -- and k_copy_out moves the used part of the
 * buffer to pinned host memory.  The host then only stitches: block header, parameter bits, the channels' code appended at
 * the running bit position, CRC16 (lnn_entropy.c).  D2H shrinks from 4 bytes per sample to the code's own size.
 * A channel whose plan is flagged (a mean inside a parameter step's guard band) or whose code would not fit `cap_bytes`
 * gets offset 0xFFFFFFFF: the host fetches its residual and codes it itself.
 * ============================================================================================== */
struct RiceEmitArgs {
    const int32_t *resid; const uint32_t *nsmp; const uint8_t *plan;
    uint32_t *offsets;                  /* [CF + 1]: byte offset of each channel-frame's code in `packed` (0xFFFFFFFF: none); [CF] = total */
    uint8_t *packed; uint64_t packed_cap;
    uint32_t C, S, CF, cap_bytes;
};

#define RSCAN_THREADS 1024
__global__ __launch_bounds__(RSCAN_THREADS) void k_rice_scan(RiceEmitArgs a)
{
    __shared__ uint64_t wsum[RSCAN_THREADS / 64];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < a.CF; base += RSCAN_THREADS) {
        const uint32_t cf = base + tid;
        uint64_t bytes = 0;
        if (cf < a.CF) {
            const uint32_t nb = *(const uint32_t *)(a.plan + (size_t)cf * LINNE_AMD_RICE_PLAN_BYTES + LINNE_AMD_RICE_PLAN_NBITS);
            const uint64_t b = (((uint64_t)nb + 63u) >> 6) << 3;           /* whole 8-byte words */
            bytes = (nb == 0xFFFFFFFFu || b > a.cap_bytes) ? 0u : b;
        }
        uint64_t incl = bytes;                               /* wave inclusive scan, then the waves' totals */
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint64_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; w++) before += wsum[w];
        const uint64_t off = before + incl - bytes;
        if (cf < a.CF) a.offsets[cf] = (bytes == 0 || off + bytes > a.packed_cap || off + bytes > 0xFFFFFFF0ull) ? 0xFFFFFFFFu : (uint32_t)off;
        __syncthreads();
        if (tid == RSCAN_THREADS - 1) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) a.offsets[a.CF] = (uint32_t)((carry > a.packed_cap) ? a.packed_cap : carry);
}

/* bit writer of one thread: its codes occupy a contiguous run of bits of the channel's (pre-zeroed) region; the run's first and
 * last 32-bit words may be shared with the neighbouring threads' runs (atomicOr), the words in between are its own (store) */
struct RiceBW {
    uint32_t *dst; uint32_t w, fill, cur, first_w;
    __device__ __forceinline__ void flush() {
        if (cur) { const uint32_t be = __builtin_bswap32(cur); if (w == first_w) atomicOr(dst + w, be); else dst[w] = be; }
        w++; cur = 0; fill = 0;
    }
    __device__ __forceinline__ void put(uint32_t val, uint32_t len) {              /* len <= 32, val < 2^len */
        while (len) {
            const uint32_t room = 32u - fill, take = len < room ? len : room;
            const uint32_t bits = (take == 32u) ? val : ((val >> (len - take)) & ((1u << take) - 1u));
            cur |= bits << (room - take);
            fill += take; len -= take;
            if (fill == 32u) flush();
        }
    }
    __device__ __forceinline__ void zeros(uint64_t z) {                              /* the region is zero already: just move on */
        const uint32_t room = 32u - fill;
        if (z < room) { fill += (uint32_t)z; return; }
        z -= room; flush();
        w += (uint32_t)(z >> 5); fill = (uint32_t)(z & 31u);
    }
    __device__ __forceinline__ void finish() { if (cur) atomicOr(dst + w, __builtin_bswap32(cur)); }
};

#define REMIT_THREADS 256
/* A thread codes `ipt` consecutive samples.  Read straight from global memory that is a 4 * ipt byte stride between lanes --
 * every load instruction touches 64 cache lines -- so the block first brings the channel-frame into LDS with coalesced loads
 * (zig-zagged on the way) and the two passes read it from there; run r starts at r * (ipt + 1) so that the lanes' strided reads
 * fall on different banks. */
template <bool LDS> __global__ __launch_bounds__(REMIT_THREADS) void k_rice_emit(RiceEmitArgs a)
{
    extern __shared__ uint32_t zbuf[];
    __shared__ uint8_t kk[1024];
    __shared__ uint64_t wsum[REMIT_THREADS / 64];
    const uint32_t cf = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t off = a.offsets[cf];
    if (off == 0xFFFFFFFFu) return;                          /* block-uniform */
    const uint8_t *rec = a.plan + (size_t)cf * LINNE_AMD_RICE_PLAN_BYTES;
    const uint32_t nbits = *(const uint32_t *)(rec + LINNE_AMD_RICE_PLAN_NBITS);
    const uint32_t n = a.nsmp[cf / a.C], best = rec[0], ns = n >> best, parts = 1u << best;
    const int32_t *x = a.resid + (size_t)cf * a.S;
    uint32_t *dst = (uint32_t *)(a.packed + off);
    const uint32_t nwords = (uint32_t)((((uint64_t)nbits + 63u) >> 6) << 1);
    const uint32_t ipt = (n + REMIT_THREADS - 1) / REMIT_THREADS;
    for (uint32_t i = tid; i < nwords; i += REMIT_THREADS) dst[i] = 0;
    for (uint32_t p = tid; p < parts; p += REMIT_THREADS) kk[p] = rec[LINNE_AMD_RICE_PLAN_K2 + p];
    if (LDS) for (uint32_t s = tid; s < n; s += REMIT_THREADS) zbuf[s + s / ipt] = rp_zz(x[s]);
    __syncthreads();
    /* a thread owns ipt consecutive samples; a partition's parameter code sits in front of the partition's first sample */
    const uint32_t s0 = tid * ipt < n ? tid * ipt : n, s1 = (s0 + ipt < n) ? s0 + ipt : n;
    const uint32_t *zrun = zbuf + (size_t)tid * (ipt + 1u);
    uint64_t mybits = 0;
    {
        uint32_t part = ns ? s0 / ns : 0, loc = ns ? s0 - part * ns : 0;
        for (uint32_t s = s0; s < s1; s++) {
            const uint32_t k2 = kk[part], k1 = k2 + 1u, k1pow = 1u << (k1 & 31u);
            if (loc == 0) mybits += part ? rp_gamma_len(rp_zz((int32_t)k2 - (int32_t)kk[part - 1])) : 15u;
            const uint32_t v = LDS ? zrun[s - s0] : rp_zz(x[s]);
            mybits += (v < k1pow) ? (k1 + 1u) : (uint64_t)(((v - k1pow) >> k2) + 2u + k2);
            if (++loc == ns) { loc = 0; part++; }
        }
    }
    uint64_t incl = mybits;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint64_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t start = incl - mybits;
    for (uint32_t w = 0; w < wave; w++) start += wsum[w];
    if (s0 >= s1) return;
    RiceBW bw; bw.dst = dst; bw.w = bw.first_w = (uint32_t)(start >> 5); bw.fill = (uint32_t)(start & 31u); bw.cur = 0;
    {
        uint32_t part = ns ? s0 / ns : 0, loc = ns ? s0 - part * ns : 0;
        for (uint32_t s = s0; s < s1; s++) {
            const uint32_t k2 = kk[part], k1 = k2 + 1u, k1pow = 1u << (k1 & 31u);       /* k2 = 31: the count modulo 32, as everywhere */
            if (loc == 0) {
                if (part == 0) bw.put((best << 5) | k2, 15u);
                else {
                    const uint32_t g = rp_zz((int32_t)k2 - (int32_t)kk[part - 1]);
                    if (g == 0) bw.put(1u, 1u);
                    else { const uint32_t nd = 32u - (uint32_t)__clz((int)(g + 1u)); bw.zeros(nd - 1u); bw.put(g + 1u, nd); }
                }
            }
            const uint32_t v = LDS ? zrun[s - s0] : rp_zz(x[s]);
            if (v < k1pow) { bw.put(1u, 1u); bw.put((k1 == 32u) ? v : (v & ((1u << (k1 & 31u)) - 1u)), k1); }
            else {
                const uint32_t d = v - k1pow;
                bw.zeros((uint64_t)(d >> k2) + 1u);
                bw.put((1u << k2) | (d & ((1u << k2) - 1u)), k2 + 1u);
            }
            if (++loc == ns) { loc = 0; part++; }
        }
    }
    bw.finish();
}

/* ================================================================================================
 * Rice DECODING on the device (the decode side of SURVEY 8f-1; linne_coder.c:306-327, bit_stream.h:305-394).  A channel's code
 * is one serial bit stream and the next channel of a block starts where this one ends, so the unit of parallel work is the
 * BLOCK: lanes = frames, every lane walks its block's channels in order from the bit position the host found behind the
 * parameter bits.  The host stage then only scans block headers and decodes the parameters (a few hundred Huffman symbols per
 * block); the compressed stream itself travels over PCIe (0.76 x 2 bytes per sample here) instead of 4-byte residuals.
 *
 * A lane's reader is 32-bit throughout: two window words and one word read ahead (w0:w1 | pre), `sh` < 32 bits of w0 already
 * consumed; peek() = the next 32 bits (one funnel shift), skip(n <= 32) = add to sh and, past 32, slide the words and load the next
 * one ahead of its use.  A sample is a peek for the zero run, a skip, a peek for the binary part, a skip.  All lanes of a wave walk
 * the SAME sample index (partition boundaries are handled inline, whatever order each lane's block chose), so 16 samples of 64
 * frames at a time go through an LDS tile and reach the residual rows in 64-byte pieces instead of 64 scattered words per store.
 *
 * Anything an encoder's stream cannot contain -- a partition order above 10 or one that does not divide the block, a parameter
 * step of more than 32 digits, a zero run into the end of the data -- ends the lane with end_bit = ~0: the host decodes that
 * stream itself (lnn_parse_block keeps the reference's behaviour for damaged streams).  Otherwise end_bit is the bit position
 * behind the last channel's code, from which the host derives the bytes the block consumed (linne_decoder.c:495-499).
 * ============================================================================================== */
struct RiceDecodeArgs {
    const uint32_t *words; uint64_t nbytes;          /* the stream segment of the group, 4-byte aligned, zero padded to 8 bytes */
    const uint64_t *bitpos;                          /* [F] bit position of the first channel's code (~0: not a COMPRESS block: skip) */
    const uint64_t *bitend;                          /* [F] bit position behind the block (its codes end before it), or NULL: the segment's end */
    const uint32_t *nsmp;                            /* [F] */
    int32_t *resid;                                  /* [F][C][S] */
    uint64_t *endbit;                                /* [F] */
    uint32_t F, C, S;
};

#define RDEC_THREADS 64
#define RDEC_TILE 16
#define RDEC_RING 64                    /* words of its stream a lane keeps in LDS */
#define RDEC_CHUNK 16                   /* words per refill (one 64-byte piece of the lane's stream) */
struct RiceBR {
    const uint32_t *words; uint32_t nwords;          /* words that may be loaded (zeros beyond) */
    const uint32_t *ring;                            /* this lane's RDEC_RING words in LDS: stream words [hi - RDEC_RING, hi) at index & (RDEC_RING - 1) */
    uint32_t hi;
    uint32_t w0, w1, pre, widx, sh;                  /* widx: index of the word after `pre` */
    __device__ __forceinline__ uint32_t load(uint32_t i) const {
        if (i < hi) return ring[i & (RDEC_RING - 1u)];                     /* (never below hi - RDEC_RING: the refill rule of k_rice_decode) */
        return (i < nwords) ? __builtin_bswap32(words[i]) : 0u;            /* beyond what is staged: a long run of zeros got ahead of the refills */
    }
    __device__ __forceinline__ void open(uint64_t bit) {
        const uint32_t i = (uint32_t)(bit >> 5);
        sh = (uint32_t)bit & 31u; w0 = load(i); w1 = load(i + 1u); pre = load(i + 2u); widx = i + 3u;
    }
    __device__ __forceinline__ uint64_t pos() const { return ((uint64_t)(widx - 3u) << 5) + sh; }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)(((((uint64_t)w0 << 32) | w1) << sh) >> 32); }
    __device__ __forceinline__ void skip(uint32_t n) {                     /* n <= 32 */
        sh += n;
        if (sh >= 32u) { sh -= 32u; w0 = w1; w1 = pre; pre = load(widx); widx++; }
    }
    /* skip(n), n <= 32, for a lane whose next word is staged (widx < hi), without a branch: every lane reads its ring (one LDS read
     * for the wave) and keeps or drops the word */
    __device__ __forceinline__ void skip_staged(uint32_t n) {
        const uint32_t nw = ring[widx & (RDEC_RING - 1u)];
        sh += n;
        const bool adv = sh >= 32u;
        sh = adv ? sh - 32u : sh; w0 = adv ? w1 : w0; w1 = adv ? pre : w1; pre = adv ? nw : pre; widx += adv ? 1u : 0u;
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {                  /* n <= 32 */
        const uint32_t v = n ? (peek() >> (32u - n)) : 0u;
        skip(n);
        return v;
    }
    /* zeros up to the next 1, which is consumed too; `bad` if the data ends first */
    __device__ __forceinline__ uint32_t zero_run(uint64_t nbits_total, bool &bad) {
        uint32_t run = 0;
        for (;;) {
            const uint32_t t = peek();
            if (t != 0u) { const uint32_t z = (uint32_t)__clz((int)t); skip(z + 1u); return run + z; }
            if (pos() >= nbits_total) { bad = true; return run; }
            skip(32u); run += 32u;
        }
    }
};

__global__ __launch_bounds__(RDEC_THREADS) void k_rice_decode(RiceDecodeArgs a)
{
    __shared__ int32_t tile[64][RDEC_TILE + 1];
    __shared__ uint32_t ring[64][RDEC_RING + 1];
    const uint32_t lane = threadIdx.x, f0 = blockIdx.x * RDEC_THREADS, f = f0 + lane;
    const bool have_frame = f < a.F;
    const uint64_t start = have_frame ? a.bitpos[f] : ~0ull;
    const bool live = have_frame && start != ~0ull;
    const uint32_t n = live ? a.nsmp[f] : 0u;
    /* a lane reads no code beyond its own block's end: a run of zeros cannot take it through the rest of the segment */
    uint64_t nbits_total = a.nbytes * 8u;
    if (live && a.bitend && a.bitend[f] < nbits_total) nbits_total = a.bitend[f];
    uint32_t nmax = n;                                                      /* the wave walks the longest block's samples */
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)nmax, m, 64); nmax = (o > nmax) ? o : nmax; }
    RiceBR r;
    r.words = a.words; r.nwords = (uint32_t)((a.nbytes + 3u) >> 2); r.ring = ring[lane];
    /* the lane's stream is staged RDEC_CHUNK words at a time: the loads of a chunk are issued at a tile boundary and land in the
     * ring at the next one, 16 samples of work later -- every lane's loads touch lines of their own, so a load waited for on the
     * spot costs the wave a full trip to memory */
    /* the contract of LINNEAmd_RiceDecodeDevice (include/linne_amd.h): the stream is readable up to the next multiple of 8 bytes.
     * Words from r.nwords on are never loaded and read as zeros, on this path as on load()'s */
    const uint32_t readable = r.nwords;
    uint4 inf[RDEC_CHUNK / 4];
    bool inflight = false;
    auto issue = [&]() {
#pragma unroll
        for (uint32_t k = 0; k < RDEC_CHUNK / 4; k++) {
            const uint32_t w = r.hi + 4u * k;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (w + 4u <= readable) v = *(const uint4 *)(a.words + w);
            else { if (w < readable) v.x = a.words[w]; if (w + 1u < readable) v.y = a.words[w + 1u]; if (w + 2u < readable) v.z = a.words[w + 2u]; }      /* (w + 3 >= readable here) */
            inf[k] = v;
        }
        inflight = true;
    };
    auto commit = [&]() {
        uint32_t *dst = ring[lane];
#pragma unroll
        for (uint32_t k = 0; k < RDEC_CHUNK / 4; k++) {
            const uint32_t o = (r.hi + 4u * k) & (RDEC_RING - 1u);
            dst[o] = __builtin_bswap32(inf[k].x); dst[o + 1u] = __builtin_bswap32(inf[k].y); dst[o + 2u] = __builtin_bswap32(inf[k].z); dst[o + 3u] = __builtin_bswap32(inf[k].w);
        }
        r.hi += RDEC_CHUNK; inflight = false;
    };
    r.hi = live ? ((uint32_t)(start >> 5) & ~(uint32_t)(RDEC_CHUNK - 1u)) : 0u;
    if (live) { issue(); commit(); issue(); commit(); issue(); commit(); }
    r.open(live ? start : 0u);
    bool bad = false;
    /* write-out of a tile: instruction i covers rows 4 i + lane / 16, columns lane % 16 */
    const uint32_t wrow = lane >> 4, wcol = lane & 15u;
    uint32_t rown[16];                                                      /* lengths of the rows this lane writes (0: not a row of this launch) */
#pragma unroll
    for (uint32_t i = 0; i < 16u; i++) { const uint32_t fr = f0 + 4u * i + wrow; rown[i] = (fr < a.F && a.bitpos[fr] != ~0ull) ? a.nsmp[fr] : 0u; }
    for (uint32_t ch = 0; ch < a.C; ch++) {
        uint32_t ns = 0, next_part = 0, k2 = 0, k1 = 1, k1pow = 2;
        bool first = true;
        if (live && !bad) {
            const uint32_t order = r.get(10);
            if (order > 10u || (n & ((1u << order) - 1u)) != 0u) bad = true;          /* no encoder writes that (lnn_entropy.c rice_emit): the host's decoder defines it */
            else ns = n >> order;
        }
        for (uint32_t s0 = 0; s0 < nmax; s0 += RDEC_TILE) {
            /* refill: a chunk goes out when the lane has at most half a ring ahead of its reader (so that the chunk, once in the
             * ring, overwrites nothing the reader still needs) */
            if (live && !bad && !inflight) {
                const uint32_t at = r.widx - 3u;                           /* the reader's first window word */
                if ((int32_t)(at - r.hi) >= (int32_t)RDEC_CHUNK) r.hi = at & ~(uint32_t)(RDEC_CHUNK - 1u);      /* (it ran ahead of the staged words) */
                if ((int32_t)(r.hi - at) <= (int32_t)(RDEC_RING / 2u) && r.hi < readable) issue();
            }
#pragma unroll 1
            for (uint32_t i = 0; i < RDEC_TILE; i++) {
                const uint32_t s = s0 + i;
                int32_t val = 0;
                const bool act = live && !bad && s < n;
                /* a partition starts in some lane: its parameter (the lanes' partitions differ in length: a few percent of the samples) */
                if (__any(act && s == next_part)) {
                    if (act && s == next_part) {
                        if (first) { k2 = r.get(5); first = false; }
                        else {
                            const uint32_t nd = r.zero_run(nbits_total, bad) + 1u;
                            if (bad || nd > 32u) bad = true;
                            else {
                                const uint32_t g = (nd == 1u) ? 0u : (uint32_t)((1ull << (nd - 1u)) + r.get(nd - 1u) - 1u);
                                k2 = (uint32_t)((int32_t)k2 + (int32_t)((g >> 1) ^ (0u - (g & 1u))));
                            }
                        }
                        k2 &= 31u; k1 = k2 + 1u; k1pow = 1u << (k1 & 31u);
                        next_part += ns;
                    }
                }
                {
                    /* The usual sample -- a zero run of at most 24, run and binary part within the 32 bits in view, the next word
                     * staged in the ring: ONE look at the stream and one branch-free step over it; when that holds in EVERY lane
                     * that has a sample (it nearly always does) the wave runs it as straight-line code, a lane without a sample
                     * stepping over 0 bits (round 4: a look and a step for the run, another pair for the binary part, each step a
                     * divergent branch around an LDS read: ~100 instructions a sample where the walk of a block is all that a
                     * launch's 8 ms are).  Anything else takes the general reader, lane by lane. */
                    const bool go = act && !bad;
                    const uint32_t t = r.peek();
                    uint32_t quot = (uint32_t)__clz((int)t), low;
                    const uint32_t kk = (quot == 0u) ? k1 : k2, used = quot + 1u + kk;
                    const bool usual = (t >> 7) != 0u && used <= 32u && r.widx < r.hi;
                    if (__all(usual || !go)) {
                        low = (kk != 0u) ? ((t << (quot + 1u)) >> (32u - kk)) : 0u;          /* (quot + 1 <= 25, 1 <= 32 - kk <= 31) */
                        r.skip_staged(go ? used : 0u);
                        const uint32_t v = (quot == 0u) ? low : (low + k1pow + ((quot - 1u) << k2));
                        val = go ? ((int32_t)(v >> 1) ^ -(int32_t)(v & 1u)) : 0;
                    } else if (go) {
                        if (usual) {
                            low = (kk != 0u) ? ((t << (quot + 1u)) >> (32u - kk)) : 0u;
                            r.skip_staged(used);
                        } else {
                            if (t >> 7) r.skip(quot + 1u);
                            else quot = r.zero_run(nbits_total, bad);
                            low = r.get((quot == 0u) ? k1 : k2);
                        }
                        const uint32_t v = (quot == 0u) ? low : (low + k1pow + ((quot - 1u) << k2));
                        val = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);
                    }
                }
                tile[lane][i] = val;
            }
            /* the chunk issued above lands now -- unless the reader ran past the staged words meanwhile (a long zero run): it
             * reads straight from memory then, and the stale chunk is dropped by moving `hi` up to the reader */
            if (inflight) {
                if (r.widx - 3u >= r.hi + RDEC_CHUNK) { inflight = false; r.hi = (r.widx - 3u) & ~(uint32_t)(RDEC_CHUNK - 1u); }
                else commit();
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t i = 0; i < 16u; i++) {
                const uint32_t row = 4u * i + wrow, s = s0 + wcol;
                if (s < rown[i]) a.resid[((size_t)(f0 + row) * a.C + ch) * a.S + s] = tile[row][wcol];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (have_frame) a.endbit[f] = !live ? 0u : ((bad || r.pos() > nbits_total) ? ~0ull : r.pos());
}


/* the used part of the packed buffer -> pinned host memory (the device knows the size, the host does not yet) */
__global__ __launch_bounds__(256) void k_copy_out(const uint4 *src, uint4 *dst, const uint32_t *total_bytes)
{
    const uint64_t n16 = ((uint64_t)*total_bytes + 15u) >> 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}


#endif

/*
 * lnn_entropy.c -- host entropy / bit-stream stage of liblinne_amd.so (north_star leaves it on the host):
 * CRC16, MSB-first bit I/O, the static Huffman code of the coefficients, the partitioned recursive Rice
 * coder, block (de)serialisation and the block-type decision.  Batch entry points run a thread pool over
 * frames (frames are independent once the sequential block-type pass is done).
 *
 * Behavioural references (file:line under /root/reference):
 *   libs/bit_stream/include/bit_stream.h:240-433      bit order, flush-to-byte semantics
 *   libs/static_huffman/src/static_huffman.c:28-165   tree construction and code assignment
 *   libs/linne_coder/src/linne_coder.c:86-327         gamma / recursive Rice / partition search
 *   libs/linne_internal/src/linne_utility.c:72-89     CRC16-IBM
 *   libs/linne_encoder/src/linne_encoder.c:480-862    block type, block layout
 *   libs/linne_decoder/src/linne_decoder.c:357-668    block parsing
 */
#include "lnn_host.h"
#include "lnn_coef_freq.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ CRC16 */
/* CRC-16/ARC (reflected 0xA001), eight bytes per step: g_crc[k][b] is the CRC contribution of byte b followed by k zero bytes */
static uint16_t g_crc[8][256];
static void crc_fold_init(void);
static void crc_init(void)
{
    uint32_t i, b, k;
    for (i = 0; i < 256; i++) {
        uint16_t c = (uint16_t)i;
        for (b = 0; b < 8; b++) c = (uint16_t)((c & 1u) ? ((c >> 1) ^ 0xA001u) : (c >> 1));
        g_crc[0][i] = c;
    }
    for (k = 1; k < 8; k++)
        for (i = 0; i < 256; i++) g_crc[k][i] = (uint16_t)((g_crc[k - 1][i] >> 8) ^ g_crc[0][g_crc[k - 1][i] & 0xFFu]);
    crc_fold_init();
}
static uint16_t crc_table(uint16_t crc, const uint8_t *data, uint64_t size)
{
    while (size >= 8) {
        uint64_t w;
        memcpy(&w, data, 8);                               /* little-endian host (x86-64) */
        w ^= crc;
        crc = (uint16_t)(g_crc[7][w & 0xFFu] ^ g_crc[6][(w >> 8) & 0xFFu] ^ g_crc[5][(w >> 16) & 0xFFu] ^ g_crc[4][(w >> 24) & 0xFFu]
                ^ g_crc[3][(w >> 32) & 0xFFu] ^ g_crc[2][(w >> 40) & 0xFFu] ^ g_crc[1][(w >> 48) & 0xFFu] ^ g_crc[0][w >> 56]);
        data += 8; size -= 8;
    }
    while (size--) crc = (uint16_t)((crc >> 8) ^ g_crc[0][(crc ^ *data++) & 0xFFu]);
    return crc;
}

#if defined(__x86_64__)
#include <immintrin.h>
/* Carry-less-multiply folding for long buffers (the CRC of a block is taken over all of its bytes: at 1.4 GB/s per thread the table
 * walk was most of the host's work per block on both sides).  A 16-byte piece of the message, loaded as it lies in memory, holds
 * the polynomial A(x) with register bit j at x^(127-j) (the CRC is bit-reflected); A(x) x^128 + D(x) is congruent, modulo the
 * generator, to lo64(A) (x) K1 + hi64(A) (x) K2 + D, where the product of two 64-bit pieces in this bit order comes out one
 * position low (so K1 = x^191 mod P, K2 = x^127 mod P; four accumulators at a time: x^575 and x^511).  What is left -- 16 folded
 * bytes and the tail -- goes through the table: the folded bytes are a message with the same remainder. */
static uint64_t g_fold[4];              /* x^575, x^511, x^191, x^127 mod P, bit j of the word at x^(63-j) */
static int g_have_clmul = 0;             /* set by lnn_tables_init */
static uint64_t xpow_repr(uint32_t e)
{
    uint32_t p = 1, i;                   /* x^0; polynomial bit i = coefficient of x^i, generator x^16 + x^15 + x^2 + 1 */
    uint64_t r = 0;
    for (i = 0; i < e; i++) { p <<= 1; if (p & 0x10000u) p ^= 0x18005u; }
    for (i = 0; i < 16; i++) if (p & (1u << i)) r |= 1ull << (63u - i);
    return r;
}
__attribute__((target("pclmul,sse4.1")))
static uint16_t crc_clmul(const uint8_t *data, uint64_t size)          /* size >= 64 */
{
    const __m128i k4 = _mm_set_epi64x((long long)g_fold[1], (long long)g_fold[0]);      /* lo: x^575 (for lo64), hi: x^511 (for hi64) */
    const __m128i k1 = _mm_set_epi64x((long long)g_fold[3], (long long)g_fold[2]);
    __m128i a0 = _mm_loadu_si128((const __m128i *)data), a1 = _mm_loadu_si128((const __m128i *)(data + 16)),
            a2 = _mm_loadu_si128((const __m128i *)(data + 32)), a3 = _mm_loadu_si128((const __m128i *)(data + 48));
    uint8_t buf[16];
    data += 64; size -= 64;
    while (size >= 64) {
#define FOLD4(a_, off_) a_ = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a_, k4, 0x00), _mm_clmulepi64_si128(a_, k4, 0x11)), _mm_loadu_si128((const __m128i *)(data + off_)))
        FOLD4(a0, 0); FOLD4(a1, 16); FOLD4(a2, 32); FOLD4(a3, 48);
#undef FOLD4
        data += 64; size -= 64;
    }
#define FOLD1(acc_, next_) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc_, k1, 0x00), _mm_clmulepi64_si128(acc_, k1, 0x11)), next_)
    a0 = FOLD1(a0, a1); a0 = FOLD1(a0, a2); a0 = FOLD1(a0, a3);
    while (size >= 16) { a0 = FOLD1(a0, _mm_loadu_si128((const __m128i *)data)); data += 16; size -= 16; }
#undef FOLD1
    _mm_storeu_si128((__m128i *)buf, a0);
    return crc_table(crc_table(0, buf, 16), data, size);
}
#endif

static void crc_fold_init(void)         /* (from lnn_tables_init's pthread_once) */
{
#if defined(__x86_64__)
    g_fold[0] = xpow_repr(575); g_fold[1] = xpow_repr(511); g_fold[2] = xpow_repr(191); g_fold[3] = xpow_repr(127);
    __builtin_cpu_init();
    g_have_clmul = (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("LINNE_AMD_NO_CLMUL")) ? 1 : 0;
#endif
}

uint16_t lnn_crc16(const uint8_t *data, uint64_t size)
{
#if defined(__x86_64__)
    if (size >= 128 && g_have_clmul > 0) return crc_clmul(data, size);
#endif
    return crc_table(0, data, size);
}

/* ------------------------------------------------------------------------------------------------ bits */
/* writer: fewer than 8 bits are pending in the low end of `acc` between calls; a put appends up to 56 bits, stores the
 * pending bits top-aligned as one big-endian 64-bit word and advances by the whole bytes among them -- no data-dependent
 * branch (the byte-wise path is taken only within 8 bytes of the end of the buffer) */
struct bitw { uint8_t *p, *base, *end; uint64_t acc; uint32_t n; int overflow; };
static void bw_open(struct bitw *w, uint8_t *mem, uint64_t size) { w->p = w->base = mem; w->end = mem + size; w->acc = 0; w->n = 0; w->overflow = 0; }
static inline void bw_put56(struct bitw *w, uint64_t val, uint32_t nbits)
{   /* 1 <= nbits <= 56, val < 2^nbits */
    uint64_t top;
    uint32_t nbytes;
    w->acc = (w->acc << nbits) | val;
    w->n += nbits;                                          /* <= 63 */
    top = w->acc << (64u - w->n);
    nbytes = w->n >> 3;
    if (__builtin_expect(w->end - w->p >= 8, 1)) { const uint64_t be = __builtin_bswap64(top); memcpy(w->p, &be, 8); w->p += nbytes; }
    else {
        uint32_t k;
        for (k = 0; k < nbytes; k++) { if (w->p < w->end) *w->p++ = (uint8_t)(top >> (56u - 8u * k)); else w->overflow = 1; }
    }
    w->n &= 7u;
}
static inline void bw_put(struct bitw *w, uint32_t val, uint32_t nbits)
{   /* nbits <= 32 */
    if (nbits == 0) return;
    bw_put56(w, (uint64_t)(val & (uint32_t)(0xFFFFFFFFu >> (32u - nbits))), nbits);
}
static inline void bw_zero_run_then_one(struct bitw *w, uint32_t run)   /* `run` zeros, then a 1 */
{
    while (run >= 32) { bw_put56(w, 0, 32); run -= 32; }
    bw_put56(w, 1, run + 1);
}
/* appends `nbits` bits of a byte-aligned, MSB-first source at the writer's bit position (the device's Rice code of a channel) */
static void bw_append_bits(struct bitw *w, const uint8_t *src, uint64_t nbits)
{
    while (nbits >= 32) {
        const uint32_t v = ((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3];
        bw_put56(w, v, 32);
        src += 4; nbits -= 32;
    }
    while (nbits >= 8) { bw_put56(w, *src++, 8); nbits -= 8; }
    if (nbits) bw_put56(w, (uint64_t)(*src >> (8u - nbits)), (uint32_t)nbits);
}
static void bw_flush(struct bitw *w)                        /* pads the last byte with zeros */
{
    if (w->n) { const uint8_t last = (uint8_t)(w->acc << (8u - w->n)); if (w->p < w->end) *w->p++ = last; else w->overflow = 1; w->n = 0; }
}
static uint64_t bw_bytes(const struct bitw *w) { return (uint64_t)(w->p - w->base); }     /* after bw_flush */

/* reader: a 64-bit window whose top `have` bits are the next bits of the stream; bytes are pulled in on demand and zeros
 * are supplied past the end (a corrupt or truncated stream decodes to garbage, never out of bounds) */
struct bitr { const uint8_t *p, *end, *base; uint64_t win; uint32_t have; uint64_t consumed; uint64_t nbits; };
static void br_open(struct bitr *r, const uint8_t *mem, uint64_t size) { r->p = r->base = mem; r->end = mem + size; r->win = 0; r->have = 0; r->consumed = 0; r->nbits = size * 8u; }
static inline void br_refill(struct bitr *r)
{
    if (__builtin_expect(r->end - r->p >= 8, 1)) {          /* top up to at least 57 bits with one load */
        uint64_t be;
        uint32_t adv;
        memcpy(&be, r->p, 8);
        r->win |= __builtin_bswap64(be) >> r->have;
        adv = (64u - r->have) >> 3;
        r->p += adv; r->have += adv * 8u;
        return;
    }
    while (r->have <= 56) {
        const uint64_t byte = (r->p < r->end) ? *r->p : 0u;
        r->p++;
        r->win |= byte << (56 - r->have);
        r->have += 8;
    }
}
static inline uint32_t br_get(struct bitr *r, uint32_t nbits)
{
    uint32_t v;
    if (nbits == 0) return 0;
    if (r->have < nbits) br_refill(r);
    v = (uint32_t)(r->win >> (64 - nbits));
    r->win <<= nbits; r->have -= nbits; r->consumed += nbits;
    return v;
}
static inline uint32_t br_zero_run(struct bitr *r)
{
    uint32_t run = 0;
    for (;;) {
        if (r->have < 32) br_refill(r);
        if (r->win >> 32) {
            const uint32_t z = (uint32_t)__builtin_clzll(r->win);
            r->win <<= (z + 1); r->have -= (z + 1); r->consumed += z + 1;
            return run + z;
        }
        if (r->consumed >= r->nbits) return run;             /* ran off the data: corrupt stream */
        r->win <<= 32; r->have -= 32; r->consumed += 32; run += 32;
    }
}
static uint64_t br_bytes(const struct bitr *r) { return (r->consumed + 7u) >> 3; }
#define br_pos_over(r) ((r)->consumed > (r)->nbits)

/* ------------------------------------------------------------------------------------------------ Huffman */
static struct { uint32_t root; uint16_t child[512][2]; uint32_t code[256]; uint8_t len[256]; } g_huff;
static void huff_walk(uint32_t node, uint32_t code, uint32_t len)
{
    if (node < 256) { g_huff.code[node] = code; g_huff.len[node] = (uint8_t)len; return; }
    huff_walk(g_huff.child[node][0], code << 1, len + 1);
    huff_walk(g_huff.child[node][1], (code << 1) | 1u, len + 1);
}
static void huff_init(void)
{   /* repeatedly merge the two least frequent live nodes; the scan order and strict '<' fix the ties */
    uint32_t weight[513], next, i;
    for (i = 0; i < 513; i++) weight[i] = 0;
    for (i = 0; i < 256; i++) weight[i] = lnn_coef_freq[i] ? lnn_coef_freq[i] : 1u;
    weight[512] = 0xFFFFFFFFu;
    for (next = 256; ; next++) {
        uint32_t lo = 512, lo2 = 512;
        for (i = 0; i < next; i++) {
            if (weight[i] == 0) continue;
            if (weight[i] < weight[lo]) { lo2 = lo; lo = i; }
            else if (weight[i] < weight[lo2]) lo2 = i;
        }
        if (lo2 == 512) break;
        weight[next] = weight[lo] + weight[lo2];
        weight[lo] = weight[lo2] = 0;
        g_huff.child[next][0] = (uint16_t)lo; g_huff.child[next][1] = (uint16_t)lo2;
    }
    g_huff.root = next - 1;
    huff_walk(g_huff.root, 0, 0);
}
static inline uint32_t huff_get(struct bitr *r)
{
    uint32_t node = g_huff.root;
    do { node = g_huff.child[node][br_get(r, 1)]; } while (node >= 256);
    return node;
}

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void rice_k2_init(void);
static void tables_init(void) { crc_init(); huff_init(); rice_k2_init(); }
void lnn_tables_init(void) { pthread_once(&g_once, tables_init); }

/* ------------------------------------------------------------------------------------------------ Rice */
#define RICE_LOG2_PARTS 10u
#define RICE_PARTS      (1u << RICE_LOG2_PARTS)

static inline uint32_t zz(int32_t v) { const uint32_t d = (uint32_t)v << 1; return (v < 0) ? ((0u - d) - 1u) : d; }
static inline int32_t unzz(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1u); }
static inline uint32_t ceil_log2(uint32_t x) { const uint32_t y = x - 1u; return y ? 32u - (uint32_t)__builtin_clz(y) : 0u; }

/* geometric-distribution ML estimate -> second-stage parameter (linne_coder.c:172-200); host libm */
static inline uint32_t rice_k2(double mean)
{
    const double optx = 0.5127629514437670454896078808815218508243560791015625;
    const double rho = 1.0 / (1.0 + mean);
    const double t = floor((log(log(optx) / log(1.0 - rho))) * 1.4426950408889634);
    return (uint32_t)((0 > t) ? 0 : t);
}
/* rice_k2 is a non-decreasing step function of the mean.  Its steps are located once with the same libm expression
 * (bisection on the double's bit pattern); afterwards the parameter is a table search, and the libm expression is used
 * again only within a guard band around a step, so the result is the expression's own value everywhere. */
static double g_k2_step[33]; static uint32_t g_k2_steps = 0;
static void rice_k2_init(void)
{
    uint32_t k;
    const double top = 8.0e9;                                   /* beyond any mean of zig-zagged 32-bit values */
    const uint32_t kmax = rice_k2(top);
    for (k = 1; k <= kmax && k <= 32; k++) {
        double lo = 0.0, hi = top;                              /* rice_k2(lo) < k <= rice_k2(hi) */
        int it;
        for (it = 0; it < 200; it++) {
            const double mid = lo + (hi - lo) * 0.5;
            if (mid <= lo || mid >= hi) break;
            if (rice_k2(mid) >= k) hi = mid; else lo = mid;
        }
        g_k2_step[k - 1] = hi;
    }
    g_k2_steps = (kmax < 32) ? kmax : 32;
}
uint32_t lnn_rice_k2_steps(double *steps)
{
    uint32_t k;
    lnn_tables_init();
    for (k = 0; k < g_k2_steps; k++) steps[k] = g_k2_step[k];
    return g_k2_steps;
}
static inline uint32_t rice_k2_fast(double mean)
{
    uint32_t k;
    int e;
    if (!(mean >= 0.0)) return rice_k2(mean);
    { uint64_t b; memcpy(&b, &mean, 8); e = (int)((b >> 52) & 0x7FFu) - 1022; }   /* frexp's exponent; the steps sit near 1.5 * 2^k */
    k = (e < 1) ? 0u : (uint32_t)(e - 1);
    if (k > g_k2_steps) k = g_k2_steps;
    while (k < g_k2_steps && mean >= g_k2_step[k]) k++;
    while (k > 0 && mean < g_k2_step[k - 1]) k--;
    if (k < g_k2_steps && mean >= g_k2_step[k] * (1.0 - LNN_RICE_GUARD)) return rice_k2(mean);      /* just below the next step */
    if (k > 0 && mean <= g_k2_step[k - 1] * (1.0 + LNN_RICE_GUARD)) return rice_k2(mean);            /* just above the last one  */
    return k;
}
/* sum over a partition of the second-stage excess ((v - 2^k1) >> k2 for v >= 2^k1): the only data-dependent part of a
 * partition's code length (every sample costs k2 + 2 bits before it) */
__attribute__((target_clones("avx512f", "avx2", "default")))
static uint32_t rice_excess(const uint32_t *q, uint32_t ns, uint32_t k1pow, uint32_t k2)
{
    uint32_t s, b = 0;
    for (s = 0; s < ns; s++) { const uint32_t v = q[s]; const uint32_t t = (v > k1pow) ? (v - k1pow) : 0u; b += t >> k2; }
    return b;
}
static inline uint32_t gamma_len(uint32_t u) { return u ? (2u * ceil_log2(u + 2u) - 1u) : 1u; }

#define RICE_MAX_DISTINCT 12u
struct rice_scratch {
    double mean[RICE_LOG2_PARTS + 1][RICE_PARTS]; uint8_t k2[RICE_LOG2_PARTS + 1][RICE_PARTS];
    uint32_t prefix[RICE_MAX_DISTINCT][RICE_PARTS + 1];
    uint32_t *u; uint32_t ucap; uint32_t *t; uint32_t tcap;
    int32_t *fetched; uint64_t fcap;    /* a frame's residual, brought in from the device for a channel without a device code */
};
/* prefix[i] = sum over the first i finest partitions (ns samples each) of the excess under parameter k */
__attribute__((target_clones("avx512f", "avx2", "default")))
static void rice_excess_prefix(const uint32_t *u, uint32_t parts, uint32_t ns, uint32_t k, uint32_t *t, uint32_t *prefix)
{
    const uint32_t k1pow = 1u << ((k + 1) & 31u), n = parts * ns;     /* k = 31: see rice_emit */
    uint32_t s, part, acc = 0;
    for (s = 0; s < n; s++) { const uint32_t v = u[s]; const uint32_t x = (v > k1pow) ? (v - k1pow) : 0u; t[s] = x >> k; }
    prefix[0] = 0;
    for (part = 0; part < parts; part++) {
        const uint32_t *q = t + (size_t)part * ns;
        uint32_t sum = 0;
        for (s = 0; s < ns; s++) sum += q[s];
        acc += sum;
        prefix[part + 1] = acc;
    }
}

/* writes one channel's code (linne_coder.c:281-302) for a given partition order and per-partition parameters */
static void rice_emit(struct bitw *w, const int32_t *data, uint32_t n, uint32_t best, const uint8_t *k2s)
{
    uint32_t part, s;
    {
        const uint32_t ns = n >> best;
        uint32_t prevk2 = 0;
        bw_put(w, best, RICE_LOG2_PARTS);
        for (part = 0; part < (1u << best); part++) {
            /* k2 = 31 (means beyond 3.2e9: not audio) makes the reference's `1U << k1` a shift by the type's width; the count
             * is taken modulo 32 here, on the device (k_rice_plan) and in the decoder -- what the reference's x86 build does */
            const uint32_t k2 = k2s[part], k1 = k2 + 1, k1pow = 1u << (k1 & 31u), k2mask = (1u << k2) - 1u;
            const uint64_t lead = 1ull << k1;
            const int32_t *q = data + (size_t)part * ns;
            if (part == 0) bw_put(w, k2, 5);
            else {
                const uint32_t g = zz((int32_t)k2 - (int32_t)prevk2);
                if (g == 0) bw_put(w, 1, 1);
                else { const uint32_t nd = ceil_log2(g + 2u); bw_put(w, 0, nd - 1); bw_put(w, g + 1, nd); }
            }
            prevk2 = k2;
            {
                /* v < 2^k1: '1' then k1 bits.  Otherwise 1 + ((v - 2^k1) >> k2) zeros, a '1', then k2 bits.  Both are one
                 * (value, length) pair chosen without a branch; only a code longer than 56 bits takes the slow way.  The
                 * writer's state lives in locals here: its byte stores may alias the struct, which would otherwise be
                 * re-read after every store */
                uint8_t *wp = w->p; uint64_t acc = w->acc; uint32_t nb = w->n;
                uint8_t *const fast_end = (w->end - w->p >= 16) ? w->end - 16 : w->p;
                for (s = 0; s < ns; s++) {
                    const uint32_t v = zz(q[s]), d = v - k1pow, quot = d >> k2;
                    const uint32_t m = 0u - (uint32_t)(v < k1pow);                       /* all ones when v < 2^k1 */
                    const uint64_t val = ((lead | v) & (0ull - (uint64_t)(m & 1u))) | (uint64_t)(((1u << k2) | (d & k2mask)) & ~m);
                    const uint32_t len = ((k1 + 1) & m) | ((quot + 2 + k2) & ~m);
                    if (__builtin_expect(len > 56 || wp >= fast_end, 0)) {
                        w->p = wp; w->acc = acc; w->n = nb;
                        if (len > 56) { bw_zero_run_then_one(w, 1 + quot); bw_put(w, d & k2mask, k2); } else bw_put56(w, val, len);
                        wp = w->p; acc = w->acc; nb = w->n;
                        continue;
                    }
                    acc = (acc << len) | val; nb += len;
                    { const uint64_t be = __builtin_bswap64(acc << (64u - nb)); memcpy(wp, &be, 8); }
                    wp += nb >> 3; nb &= 7u;
                }
                w->p = wp; w->acc = acc; w->n = nb;
            }
        }
    }
}

#ifdef LNN_PROF
#include <x86intrin.h>
static uint64_t g_prof[8];
#define PROF(i) do { const uint64_t t_ = __rdtsc(); g_prof[i] += t_ - prof_t; prof_t = t_; } while (0)
#define PROF_BEGIN uint64_t prof_t = __rdtsc()
#else
#define PROF(i) do { } while (0)
#define PROF_BEGIN do { } while (0)
#endif
static int rice_encode(struct bitw *w, const int32_t *data, uint32_t n, struct rice_scratch *sc)
{
    uint32_t max_order = 1, parts, order, part, s, best = 0, min_bits = 0xFFFFFFFFu;
    int32_t i;
    uint32_t *u;
    if (sc->ucap < n) { free(sc->u); sc->u = malloc(sizeof(uint32_t) * n); sc->ucap = sc->u ? n : 0; if (!sc->u) return -1; }
    u = sc->u;
    PROF_BEGIN;
    for (s = 0; s < n; s++) u[s] = zz(data[s]);
    PROF(0);
    while ((n % (1u << max_order)) == 0) max_order++;
    max_order = (max_order - 1 < RICE_LOG2_PARTS) ? max_order - 1 : RICE_LOG2_PARTS;
    parts = 1u << max_order;
    {
        const uint32_t ns = n / parts;
        for (part = 0; part < parts; part++) {
            uint64_t sum = 0;                  /* n < 2^16 values below 2^32: the double sum of the reference is this integer */
            const uint32_t *q = u + (size_t)part * ns;
            for (s = 0; s < ns; s++) sum += q[s];
            sc->mean[max_order][part] = (double)sum / ns;
        }
    }
    for (i = (int32_t)max_order - 1; i >= 0; i--)
        for (part = 0; part < (1u << i); part++)
            sc->mean[i][part] = (sc->mean[i + 1][2 * part] + sc->mean[i + 1][2 * part + 1]) / 2.0;
    PROF(1);
    /* Code length of every partitioning.  A partition with parameter k costs ns * (k + 2) bits plus the sum of its samples'
     * second-stage excess, and that sum is additive over sub-partitions: so the excess is evaluated once per DISTINCT k that
     * any (order, partition) selects -- one vectorised pass over the block and a prefix sum over the finest partitions each
     * -- and every coarser partition is a difference of two prefix entries.  (All sums are uint32 with wrap-around, like the
     * accumulators of linne_coder.c:256-279; a difference of wrapped prefixes is the wrapped sum.) */
    {
        uint32_t used = 0, distinct = 0, k;
        for (order = 0; order <= max_order; order++)
            for (part = 0; part < (1u << order); part++) {
                const uint32_t k2 = rice_k2_fast(sc->mean[order][part]) & 31u;
                sc->k2[order][part] = (uint8_t)k2;
                used |= 1u << k2;
            }
        for (k = 0; k < 32; k++) distinct += (used >> k) & 1u;
        PROF(4);
        if (distinct <= RICE_MAX_DISTINCT) {
            uint32_t slot_of[32], nslot = 0;
            if (sc->tcap < n) { free(sc->t); sc->t = malloc(sizeof(uint32_t) * n); sc->tcap = sc->t ? n : 0; if (!sc->t) return -1; }
            for (k = 0; k < 32; k++) if ((used >> k) & 1u) { slot_of[k] = nslot; rice_excess_prefix(u, parts, n / parts, k, sc->t, sc->prefix[nslot]); nslot++; }
            PROF(5);
            for (order = 0; order <= max_order; order++) {
                const uint32_t ns = n >> order, sh = max_order - order;
                uint32_t prevk2 = 0, bits = 0;
                for (part = 0; part < (1u << order); part++) {
                    const uint32_t k2 = sc->k2[order][part];
                    const uint32_t *pf = sc->prefix[slot_of[k2]];
                    bits += ns * (k2 + 2) + (pf[(part + 1) << sh] - pf[part << sh]);
                    bits += part ? gamma_len(zz((int32_t)k2 - (int32_t)prevk2)) : 5u;
                    prevk2 = k2;
                }
                if (min_bits > bits) { min_bits = bits; best = order; }
            }
        } else {
            for (order = 0; order <= max_order; order++) {
                const uint32_t ns = n >> order;
                uint32_t prevk2 = 0, bits = 0;
                for (part = 0; part < (1u << order); part++) {
                    const uint32_t k2 = sc->k2[order][part], k1pow = 1u << ((k2 + 1) & 31u);
                    const uint32_t *q = u + (size_t)part * ns;
                    bits += ns * (k2 + 2) + rice_excess(q, ns, k1pow, k2);      /* (v < 2^k1) ? k1+1 : k2+2+((v-2^k1)>>k2), k1 = k2+1 */
                    bits += part ? gamma_len(zz((int32_t)k2 - (int32_t)prevk2)) : 5u;
                    prevk2 = k2;
                }
                if (min_bits > bits) { min_bits = bits; best = order; }
            }
        }
    }
    PROF(2);
    rice_emit(w, data, n, best, sc->k2[best]);
    PROF(3);
    return 0;
}

/* returns 0, or -1 where the reference's own behaviour is undefined (a partition order of 32 or more: `1 << order`,
 * linne_coder.c:313; a parameter step whose gamma code is longer than 32 digits: a 33-bit read): only a damaged stream read
 * with the CRC check off gets here.  Orders 11 .. 31, which no encoder writes, ARE defined in the reference -- 2^order
 * partitions of n >> order samples (possibly none), each with its parameter step -- and are decoded the same way here
 * (tests/golden/corrupt_decode.json). */
static int rice_decode(struct bitr *r, int32_t *data, uint32_t n)
{
    const uint32_t order = br_get(r, RICE_LOG2_PARTS), ns = (order <= 31u) ? (n >> order) : 0u;
    uint32_t part, s, k2 = 0;
    if (order > 31u) return -1;
    for (part = 0; part < (1u << order); part++) {
        if (br_pos_over(r)) return 0;
        if (part == 0) k2 = br_get(r, 5);
        else {
            const uint32_t nd = br_zero_run(r) + 1;
            if (nd > 32) return -1;
            const uint32_t g = (nd == 1) ? 0u : (uint32_t)((1ul << (nd - 1)) + br_get(r, nd - 1) - 1);
            k2 = (uint32_t)((int32_t)k2 + unzz(g));
        }
        k2 &= 31u;
        {
            const uint32_t k1 = k2 + 1, k1pow = 1u << (k1 & 31u);        /* k2 = 31: see rice_emit */
            int32_t *q = data + (size_t)part * ns;
            {   /* the reader's state lives in locals in this loop (stores to q[] do not alias it, but it keeps the window,
                 * the count and the pointer in registers across the general-reader calls' boundaries) */
                for (s = 0; s < ns; s++) {
                    uint32_t z, zm, nb, low, v, len;
                    if (r->have < 57) br_refill(r);
                    z = (uint32_t)__builtin_clzll(r->win | 1u);
                    if (__builtin_expect(z > 24, 0)) {      /* long zero run (or the end of the data): the general reader */
                        const uint32_t quot = br_zero_run(r);
                        v = (quot == 0) ? br_get(r, k1) : (br_get(r, k2) + k1pow + ((quot - 1) << k2));
                        q[s] = unzz(v);
                        continue;
                    }
                    /* z zeros and a '1', then k1 bits (z == 0) or k2 bits: at most 25 + 31 bits, all inside the window;
                     * selected with masks, not branches (z == 0 is a coin toss) */
                    zm = 0u - (uint32_t)(z == 0);
                    nb = (k1 & zm) | (k2 & ~zm);
                    low = (uint32_t)(((r->win << (z + 1)) >> 1) >> (63u - nb));
                    v = low + ((k1pow + ((z - 1) << k2)) & ~zm);
                    len = z + 1 + nb;
                    r->win <<= len; r->have -= len; r->consumed += len;
                    q[s] = unzz(v);
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ blocks */
static void put_be16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static uint32_t get_be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }
static uint32_t get_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int lnn_shape_layers(const struct LINNEAmdShape *shape, struct lnn_layers *out)
{
    uint32_t l, off = 0;
    if (!shape || lnn_preset_info(shape->preset, &out->num_layers, out->size, &out->num_regs, out->regs) != 0) return -1;
    out->max_size = 0;
    for (l = 0; l < out->num_layers; l++) { out->offset[l] = off; off += out->size[l]; if (out->size[l] > out->max_size) out->max_size = out->size[l]; }
    out->total = off;
    return 0;
}

/* Block-type decision (linne_encoder.c:480-529) from the device statistics; the transcendental part of
 * LPCCalculator_EstimateCodeLength (lpc.c:832-859) runs here with the host libm.  *state carries what the
 * reference's calculator holds in parcor[order] (oracle quirk Q2). */
/* where a frame's PCM lives: a frames array [F][C][S] (frame f, channel ch at frames + (f * C + ch) * S) or the caller's planes
 * (channel ch of frame f at planes[ch] + first_sample + f * S) */
struct pcm_view { const int32_t *frames; const int32_t *const *planes; uint64_t first_sample; };
static inline const int32_t *pcm_channel(const struct pcm_view *v, const struct LINNEAmdShape *shape, uint64_t f, uint32_t ch)
{
    const uint64_t S = shape->num_samples_per_block;
    return v->frames ? v->frames + (f * shape->num_channels + ch) * S : v->planes[ch] + v->first_sample + f * S;
}

static uint32_t decide_block_type_view(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, uint32_t n,
        const struct pcm_view *pv, uint64_t f, const double *stats_frame, double *state);

uint32_t lnn_decide_block_type(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, uint32_t n,
        const int32_t *pcm_frame, const double *stats_frame, double *state)
{
    struct pcm_view pv; pv.frames = pcm_frame; pv.planes = NULL; pv.first_sample = 0;
    return decide_block_type_view(shape, ly, n, &pv, 0, stats_frame, state);
}

static uint32_t decide_block_type_view(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, uint32_t n,
        const struct pcm_view *pv, uint64_t f, const double *stats_frame, double *state)
{
    const uint32_t C = shape->num_channels, bits = shape->bits_per_sample, order = ly->size[0];
    uint32_t ch, s, ord;
    double mean = 0.0;
    for (ch = 0; ch < C; ch++) {
        const double *st = stats_frame + (size_t)ch * LINNE_AMD_STAT_WORDS;
        const int zero = st[LINNE_AMD_ST_ZERO] != 0.0;
        double p, ratio, len;
        if (zero) *state = 0.0;                                    /* the zero branches write parcor[0..order] */
        p = st[LINNE_AMD_ST_R0];
        p *= ldexp(1.0, (int)(2u * (bits - 1u)));                  /* pow(2, 2(bits-1)), exact */
        if (fabs(p) <= FLT_MIN) len = 0.0;
        else {
            p = log(p) * 1.4426950408889634 - log((double)n) * 1.4426950408889634;
            ratio = 0.0;
            for (ord = 1; ord < order; ord++) { const double k = zero ? 0.0 : st[LINNE_AMD_ST_K1 + ord - 1]; ratio += log(1.0 - k * k) * 1.4426950408889634; }
            ratio += log(1.0 - (*state) * (*state)) * 1.4426950408889634;
            len = 1.9426950408889634 + 0.5f * (p + ratio);
            if (len <= 0) len = 1.0;
        }
        mean += len;
    }
    mean /= C;
    mean /= bits;
    if (mean >= 0.95f) return LNN_BLOCK_RAW;
    for (ch = 0; ch < C; ch++) {
        const int32_t *x = pcm_channel(pv, shape, f, ch);
        for (s = 0; s < n; s++) if (x[s] != 0) return LNN_BLOCK_COMPRESS;
    }
    return LNN_BLOCK_SILENT;
}

/* serialises one block (linne_encoder.c:806-855); returns LNN_* and the byte count */
/* emitted: the device's Rice code of this frame's channels (packed + offsets[ch], bit lengths in the plan records); a channel
 * without one (offset 0xFFFFFFFF) is coded here from the residual, which fetch() brings in when the caller did not pass it */
struct emitted { const uint8_t *packed; const uint32_t *offsets; int (*fetch)(void *arg, uint32_t frame, int32_t *dst); void *fetch_arg; uint32_t frame; };
static int pack_block(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, uint32_t type, uint32_t n,
        const struct pcm_view *pv, uint64_t f, const int32_t *residual, const int32_t *params, const uint8_t *plan, const struct emitted *em,
        uint8_t *out, uint64_t cap, uint32_t *size_out, struct rice_scratch *sc)
{
    const uint32_t C = shape->num_channels, bits = shape->bits_per_sample, S = shape->num_samples_per_block;
    uint32_t ch, l, i, s;
    uint64_t body = 0;
    if (cap < 11) return LNN_INSUFFICIENT_BUFFER;
    put_be16(out, 0xFFFF);
    out[8] = (uint8_t)type; put_be16(out + 9, n);
    if (type == LNN_BLOCK_RAW) {                                   /* linne_encoder.c:532-591 */
        uint8_t *p = out + 11;
        if (bits != 8 && bits != 16 && bits != 24) return LNN_INVALID_FORMAT;
        if (cap - 11 < ((uint64_t)bits * n * C) / 8) return LNN_INSUFFICIENT_BUFFER;
        for (s = 0; s < n; s++)
            for (ch = 0; ch < C; ch++) {
                const uint32_t u = zz(pcm_channel(pv, shape, f, ch)[s]);
                if (bits == 24) *p++ = (uint8_t)(u >> 16);
                if (bits >= 16) *p++ = (uint8_t)(u >> 8);
                *p++ = (uint8_t)u;
            }
        body = (uint64_t)(p - (out + 11));
    } else if (type == LNN_BLOCK_COMPRESS) {                       /* linne_encoder.c:698-749 */
        struct bitw w;
        bw_open(&w, out + 11, cap - 11);
        for (ch = 0; ch < C; ch++) {
            const int32_t *rec = params + (size_t)ch * LINNE_AMD_PARAM_WORDS;
            for (l = 0; l < 2; l++) {
                bw_put(&w, zz(rec[LINNE_AMD_PRM_PREV + l]), bits + 1);
                bw_put(&w, (uint32_t)rec[LINNE_AMD_PRM_PCOEF + l], 4);
            }
        }
        for (ch = 0; ch < C; ch++) {
            const int32_t *rec = params + (size_t)ch * LINNE_AMD_PARAM_WORDS;
            for (l = 0; l < ly->num_layers; l++) {
                bw_put(&w, ceil_log2((uint32_t)rec[LINNE_AMD_PRM_UNITS + l]), 3);
                bw_put(&w, (uint32_t)rec[LINNE_AMD_PRM_RSHIFT + l], 4);
                for (i = 0; i < ly->size[l]; i++) {
                    const uint32_t sym = zz(rec[LINNE_AMD_PRM_COEF + ly->offset[l] + i]) & 255u;
                    bw_put(&w, g_huff.code[sym], g_huff.len[sym]);
                }
            }
        }
        for (ch = 0; ch < C; ch++) {
            /* the device's plan (order + parameters) when there is one and none of its means sat in a guard band */
            const uint8_t *pl = plan ? plan + (size_t)ch * LINNE_AMD_RICE_PLAN_BYTES : NULL;
            if (em && pl && em->offsets[ch] != 0xFFFFFFFFu) {          /* the code itself came from the device: append it */
                uint32_t nb; memcpy(&nb, pl + LINNE_AMD_RICE_PLAN_NBITS, 4);
                bw_append_bits(&w, em->packed + em->offsets[ch], nb);
                continue;
            }
            if (!residual) {                                         /* emit mode: the residual stayed on the device */
                if (!em || !em->fetch) return LNN_NG;
                if (sc->fcap < (uint64_t)C * S) { free(sc->fetched); sc->fetched = malloc(sizeof(int32_t) * (size_t)C * S); sc->fcap = sc->fetched ? (uint64_t)C * S : 0; if (!sc->fetched) return LNN_NG; }
                if (em->fetch(em->fetch_arg, em->frame, sc->fetched) != LNN_OK) return LNN_NG;
                residual = sc->fetched;
            }
            if (pl && pl[1] == 0 && pl[0] <= RICE_LOG2_PARTS && (n % (1u << pl[0])) == 0)
                rice_emit(&w, residual + (size_t)ch * S, n, pl[0], pl + LINNE_AMD_RICE_PLAN_K2);
            else if (rice_encode(&w, residual + (size_t)ch * S, n, sc) != 0) return LNN_NG;
        }
        bw_flush(&w);
        if (w.overflow) return LNN_INSUFFICIENT_BUFFER;
        body = bw_bytes(&w);
    }
    if (body + 5 > 0xFFFFFFFFull) return LNN_INSUFFICIENT_BUFFER;
    put_be32(out + 2, (uint32_t)body + 5);
    put_be16(out + 6, lnn_crc16(out + 8, body + 3));
    *size_out = (uint32_t)(11 + body);
    return LNN_OK;
}

/* ---- fork/join over a range (frames are independent) -------------------------------------------------- */
struct pf_job { void (*fn)(void *, uint32_t, uint32_t); void *arg; uint32_t first, count; };
static void *pf_worker(void *a) { struct pf_job *j = a; j->fn(j->arg, j->first, j->count); return NULL; }
void lnn_parallel_for(uint32_t count, uint32_t num_threads, void (*fn)(void *arg, uint32_t first, uint32_t count), void *arg)
{
    pthread_t th[64]; struct pf_job jobs[64]; int started[64];
    uint32_t t, first = 0, nt = num_threads ? num_threads : 1;
    if (count == 0) return;
    if (nt > 64) nt = 64;
    if (nt > count) nt = count;
    if (nt == 1) { fn(arg, 0, count); return; }
    for (t = 0; t < nt; t++) {
        const uint32_t c = count / nt + ((t < count % nt) ? 1u : 0u);
        jobs[t].fn = fn; jobs[t].arg = arg; jobs[t].first = first; jobs[t].count = c; first += c;
        started[t] = (t + 1 < nt) && pthread_create(&th[t], NULL, pf_worker, &jobs[t]) == 0;
        if (!started[t]) pf_worker(&jobs[t]);              /* the last share (and any the OS refused a thread for) runs here */
    }
    for (t = 0; t < nt; t++) if (started[t]) pthread_join(th[t], NULL);
}

/* ---- thread pool over frames ------------------------------------------------------------------------- */
/* Each worker owns a contiguous range of frames and a region of the pool; it serialises its blocks back to back there.
 * After the join the regions are copied to their places in the output, again in parallel. */
struct pack_share {
    const struct LINNEAmdShape *shape; const struct lnn_layers *ly;
    struct pcm_view pv; const int32_t *residual, *params; const uint32_t *nsmp; const uint8_t *types, *plan;
    const uint8_t *packed; const uint32_t *offsets; int (*fetch)(void *, uint32_t, int32_t *); void *fetch_arg;
    uint8_t *pool, *out; uint64_t per_slot; uint32_t *sizes; int *rets;
    uint32_t nshare; uint32_t share_first[65]; uint64_t share_dst[65];
};
static void pack_range(void *arg, uint32_t first, uint32_t count)
{
    struct pack_share *j = arg;
    struct rice_scratch *sc = calloc(1, sizeof(*sc));
    const uint32_t C = j->shape->num_channels;
    const uint64_t CS = (uint64_t)C * j->shape->num_samples_per_block;
    uint8_t *cursor = j->pool + j->per_slot * first;
    uint32_t f;
    for (f = first; f < first + count; f++) {
        if (!sc) { j->rets[f] = LNN_NG; continue; }
        struct emitted em;
        em.packed = j->packed; em.offsets = j->offsets ? j->offsets + (size_t)f * C : NULL; em.fetch = j->fetch; em.fetch_arg = j->fetch_arg; em.frame = f;
        j->rets[f] = pack_block(j->shape, j->ly, j->types[f], j->nsmp ? j->nsmp[f] : j->shape->num_samples_per_block,
                &j->pv, f, j->residual ? j->residual + f * CS : NULL, j->params + (size_t)f * C * LINNE_AMD_PARAM_WORDS,
                j->plan ? j->plan + (size_t)f * C * LINNE_AMD_RICE_PLAN_BYTES : NULL, j->offsets ? &em : NULL,
                cursor, j->per_slot, &j->sizes[f], sc);
        if (j->rets[f] == LNN_OK) cursor += j->sizes[f];
    }
    if (sc) { free(sc->u); free(sc->t); free(sc->fetched); free(sc); }
}
static void copy_range(void *arg, uint32_t first, uint32_t count)
{
    struct pack_share *j = arg;
    uint32_t t, f;
    for (t = first; t < first + count; t++) {
        uint64_t bytes = 0;
        for (f = j->share_first[t]; f < j->share_first[t + 1]; f++) bytes += j->sizes[f];
        memcpy(j->out + j->share_dst[t], j->pool + j->per_slot * j->share_first[t], bytes);
    }
}

/* The serialisation pool of a batch (worst-case bytes per frame x frames: hundreds of MB of address space, a few tens of MB
 * touched) is kept between calls: a fresh mapping per call costs thousands of page faults on 16 threads at once.  One pool per
 * process; a second concurrent caller simply gets its own temporary one. */
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
static uint8_t *g_pool = NULL; static uint64_t g_pool_size = 0; static int g_pool_busy = 0;
static uint8_t *pool_take(uint64_t bytes, int *cached)
{
    uint8_t *p = NULL;
    *cached = 0;
    pthread_mutex_lock(&g_pool_lock);
    if (!g_pool_busy) {
        if (g_pool_size < bytes) { free(g_pool); g_pool = malloc(bytes); g_pool_size = g_pool ? bytes : 0; }
        if (g_pool) { p = g_pool; g_pool_busy = 1; *cached = 1; }
    }
    pthread_mutex_unlock(&g_pool_lock);
    return p ? p : malloc(bytes);
}
static void pool_give(uint8_t *p, int cached)
{
    if (!p) return;
    if (!cached) { free(p); return; }
    pthread_mutex_lock(&g_pool_lock);
    g_pool_busy = 0;
    pthread_mutex_unlock(&g_pool_lock);
}

int LINNEAmd_PackFrames(const struct LINNEAmdShape *shape, const int32_t *pcm, const uint32_t *num_samples,
        uint32_t num_frames, const int32_t *residual, const int32_t *params, const double *stats,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state,
        uint32_t num_threads)
{
    return LINNEAmd_PackFramesPlanned(shape, pcm, num_samples, num_frames, residual, params, stats, NULL,
            blocks_out, blocks_capacity, block_sizes, parcor_state, num_threads);
}

static int pack_frames_core(const struct LINNEAmdShape *shape, const struct pcm_view *pv, const uint32_t *num_samples,
        uint32_t num_frames, const int32_t *residual, const int32_t *params, const double *stats, const uint8_t *rice_plan,
        const uint8_t *packed, const uint32_t *offsets, int (*fetch)(void *, uint32_t, int32_t *), void *fetch_arg,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state, uint32_t num_threads);

int LINNEAmd_PackFramesPlanned(const struct LINNEAmdShape *shape, const int32_t *pcm, const uint32_t *num_samples,
        uint32_t num_frames, const int32_t *residual, const int32_t *params, const double *stats, const uint8_t *rice_plan,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state,
        uint32_t num_threads)
{
    struct pcm_view pv; pv.frames = pcm; pv.planes = NULL; pv.first_sample = 0;
    if (!pcm || !residual) return LNN_INVALID_ARGUMENT;
    return pack_frames_core(shape, &pv, num_samples, num_frames, residual, params, stats, rice_plan, NULL, NULL, NULL, NULL,
            blocks_out, blocks_capacity, block_sizes, parcor_state, num_threads);
}

/* The host stage when the device wrote the Rice codes (LINNEAmd_RiceEmitDevice / staging slots with LINNE_AMD_SLOT_EMIT): block
 * types in stream order (quirk Q2), then per block the header, the parameter bits (linne_encoder.c:698-735), the channels' codes
 * appended at the running bit position, padding and CRC16 (:743-749, :848-855).  The frames' PCM is read in place from the
 * caller's planes (frame f of the batch starts at sample first_sample + f * num_samples_per_block of every plane): it is needed
 * for the SILENT test and for RAW blocks only.  fetch(arg, frame, dst[C][S]) must deliver a frame's residual; it is called for
 * the (rare) channel-frames without a device code. */
int LINNEAmd_PackFramesEmitted(const struct LINNEAmdShape *shape, const int32_t *const *planes, uint64_t first_sample,
        const uint32_t *num_samples, uint32_t num_frames, const int32_t *params, const double *stats, const uint8_t *rice_plan,
        const uint8_t *packed, const uint32_t *offsets, int (*fetch)(void *arg, uint32_t frame, int32_t *dst), void *fetch_arg,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state, uint32_t num_threads)
{
    struct pcm_view pv; pv.frames = NULL; pv.planes = planes; pv.first_sample = first_sample;
    if (!planes || !rice_plan || !packed || !offsets) return LNN_INVALID_ARGUMENT;
    return pack_frames_core(shape, &pv, num_samples, num_frames, NULL, params, stats, rice_plan, packed, offsets, fetch, fetch_arg,
            blocks_out, blocks_capacity, block_sizes, parcor_state, num_threads);
}

static int pack_frames_core(const struct LINNEAmdShape *shape, const struct pcm_view *pv, const uint32_t *num_samples,
        uint32_t num_frames, const int32_t *residual, const int32_t *params, const double *stats, const uint8_t *rice_plan,
        const uint8_t *packed, const uint32_t *offsets, int (*fetch)(void *, uint32_t, int32_t *), void *fetch_arg,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state, uint32_t num_threads)
{
    struct lnn_layers ly;
    uint8_t *types = NULL, *pool = NULL;
    uint64_t CS, per_slot;
    int pool_cached = 0;
    int *rets = NULL, ret = LNN_OK;
    double state = parcor_state ? *parcor_state : 0.0;
    uint32_t f, t, C;
    if (!shape || !params || !stats || !blocks_out || !block_sizes) return LNN_INVALID_ARGUMENT;
    if (lnn_shape_layers(shape, &ly) != 0) return LNN_INVALID_FORMAT;
    if (num_frames == 0) return LNN_OK;
    lnn_tables_init();
    C = shape->num_channels;
    CS = (uint64_t)C * shape->num_samples_per_block;
    types = malloc(num_frames);
    rets = malloc(sizeof(int) * num_frames);
    if (!types || !rets) { ret = LNN_NG; goto done; }
    /* sequential pass: block types (the only cross-frame dependency, quirk Q2) */
    for (f = 0; f < num_frames; f++) {
        const uint32_t n = num_samples ? num_samples[f] : shape->num_samples_per_block;
        const double *st = stats + (size_t)f * C * LINNE_AMD_STAT_WORDS;
        types[f] = (uint8_t)decide_block_type_view(shape, &ly, n, pv, f, st, &state);
        if (types[f] == LNN_BLOCK_COMPRESS) state = st[(size_t)(C - 1) * LINNE_AMD_STAT_WORDS + LINNE_AMD_ST_TAIL];
    }
    if (parcor_state) *parcor_state = state;
    /* parallel pass: serialise into per-worker regions, then copy the regions into place */
    per_slot = 64 + CS * 8;
    pool = pool_take(per_slot * (uint64_t)num_frames, &pool_cached);
    if (!pool) { ret = LNN_NG; goto done; }
    {
        struct pack_share sh;
        uint32_t nt, first = 0;
        uint64_t off = 0;
        if (num_threads == 0) num_threads = 1;
        if (num_threads > 64) num_threads = 64;
        nt = (num_threads < num_frames) ? num_threads : num_frames;
        sh.shape = shape; sh.ly = &ly; sh.pv = *pv; sh.residual = residual; sh.params = params; sh.nsmp = num_samples;
        sh.packed = packed; sh.offsets = offsets; sh.fetch = fetch; sh.fetch_arg = fetch_arg;
        sh.types = types; sh.plan = rice_plan; sh.pool = pool; sh.out = blocks_out; sh.per_slot = per_slot; sh.sizes = block_sizes; sh.rets = rets;
        lnn_parallel_for(num_frames, nt, pack_range, &sh);
        for (f = 0; f < num_frames; f++) if (rets[f] != LNN_OK) { ret = rets[f]; goto done; }
        /* the shares lnn_parallel_for handed out: count / nt frames each, the first count % nt one more */
        sh.nshare = nt;
        for (t = 0; t < nt; t++) {
            const uint32_t c = num_frames / nt + ((t < num_frames % nt) ? 1u : 0u);
            sh.share_first[t] = first; sh.share_dst[t] = off;
            for (f = first; f < first + c; f++) off += block_sizes[f];
            first += c;
        }
        sh.share_first[nt] = first;
        if (off > blocks_capacity) { ret = LNN_INSUFFICIENT_BUFFER; goto done; }
        lnn_parallel_for(nt, nt, copy_range, &sh);
    }
done:
    free(types); free(rets); pool_give(pool, pool_cached);
    return ret;
}

/* ---- block parsing (linne_decoder.c:564-668 without the synthesis) ------------------------------------ */
int lnn_parse_block(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, const uint8_t *data, uint64_t avail,
        int check_crc, uint32_t max_samples, uint32_t *type_out, uint32_t *n_out, uint32_t *consumed_out,
        int32_t *samples /* [C][S]: residual (COMPRESS) or PCM (RAW/SILENT) */, int32_t *params)
{
    return lnn_parse_block_head(shape, ly, data, avail, check_crc, max_samples, type_out, n_out, consumed_out, samples, params, NULL);
}

/* The same; with rice_bit_out != NULL a COMPRESS block is parsed up to its parameters only: *rice_bit_out = the bit offset (from the
 * block's first byte) at which the first channel's Rice code starts, `samples` is left alone and *consumed_out is the block's size
 * field + 6 (what a well-formed block consumes; the device's decoder reports the real figure, LINNEAmd_RiceDecodeDevice). */
int lnn_parse_block_head(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, const uint8_t *data, uint64_t avail,
        int check_crc, uint32_t max_samples, uint32_t *type_out, uint32_t *n_out, uint32_t *consumed_out,
        int32_t *samples, int32_t *params, uint64_t *rice_bit_out)
{
    const uint32_t C = shape->num_channels, bits = shape->bits_per_sample, S = shape->num_samples_per_block;
    uint32_t bsize, type, n, ch, l, i, s;
    if (avail < 11) return LNN_INSUFFICIENT_DATA;
    if (get_be16(data) != 0xFFFF) return LNN_INVALID_FORMAT;
    bsize = get_be32(data + 2);
    if ((uint64_t)bsize + 6 > avail) return LNN_INSUFFICIENT_DATA;
    if (bsize < 5) return LNN_INVALID_FORMAT;
    if (check_crc && lnn_crc16(data + 8, bsize - 2) != get_be16(data + 6)) return LNN_DETECT_DATA_CORRUPTION;
    type = data[8]; n = get_be16(data + 9);
    if (n > max_samples || n > S) return LNN_INSUFFICIENT_BUFFER;
    *type_out = type; *n_out = n;
    if (type == LNN_BLOCK_RAW) {
        const uint8_t *q = data + 11;
        if (bits != 8 && bits != 16 && bits != 24) return LNN_INVALID_FORMAT;
        if (avail - 11 < ((uint64_t)bits * n * C) / 8) return LNN_INSUFFICIENT_DATA;
        for (s = 0; s < n; s++)
            for (ch = 0; ch < C; ch++) {
                uint32_t u = 0;
                if (bits == 24) u = *q++;
                if (bits >= 16) u = (u << 8) | *q++;
                u = (u << 8) | *q++;
                samples[(size_t)ch * S + s] = unzz(u);
            }
        *consumed_out = 11 + (uint32_t)(q - (data + 11));
    } else if (type == LNN_BLOCK_SILENT) {
        for (ch = 0; ch < C; ch++) memset(samples + (size_t)ch * S, 0, sizeof(int32_t) * n);
        *consumed_out = 11;
    } else if (type == LNN_BLOCK_COMPRESS) {
        struct bitr r;
        if (n == 0) return LNN_INVALID_FORMAT;
        br_open(&r, data + 11, avail - 11);
        for (ch = 0; ch < C; ch++) {
            int32_t *rec = params + (size_t)ch * LINNE_AMD_PARAM_WORDS;
            memset(rec, 0, sizeof(int32_t) * LINNE_AMD_PARAM_WORDS);
            for (l = 0; l < 2; l++) {
                rec[LINNE_AMD_PRM_PREV + l] = unzz(br_get(&r, bits + 1));
                rec[LINNE_AMD_PRM_PCOEF + l] = (int32_t)br_get(&r, 4);
            }
        }
        for (ch = 0; ch < C; ch++) {
            int32_t *rec = params + (size_t)ch * LINNE_AMD_PARAM_WORDS;
            for (l = 0; l < ly->num_layers; l++) {
                rec[LINNE_AMD_PRM_UNITS + l] = (int32_t)(1u << br_get(&r, 3));
                rec[LINNE_AMD_PRM_RSHIFT + l] = (int32_t)br_get(&r, 4);
                for (i = 0; i < ly->size[l]; i++) rec[LINNE_AMD_PRM_COEF + ly->offset[l] + i] = unzz(huff_get(&r));
            }
        }
        if (rice_bit_out) { *rice_bit_out = 88u + r.consumed; *consumed_out = bsize + 6u; return LNN_OK; }
        for (ch = 0; ch < C; ch++) if (rice_decode(&r, samples + (size_t)ch * S, n) != 0) return LNN_INVALID_FORMAT;
        *consumed_out = 11 + (uint32_t)br_bytes(&r);
    } else return LNN_INVALID_FORMAT;
    return LNN_OK;
}

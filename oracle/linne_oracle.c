/*
 * linne_oracle.c -- TEST INFRASTRUCTURE ONLY (see linne_oracle.h).
 *
 * CPU restatement of the LINNE codec's per-frame prediction path plus the host entropy/bit-stream stage
 * needed to compare whole .lnn streams.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).  Floating-point work keeps the reference's association order:
 * build with -ffp-contract=off (the reference is ISO C90, i.e. no contraction).
 *
 * Two reference quirks are reproduced on purpose because they change output bytes:
 *   Q1  LPC_ApplyWindow(WELCH) never writes the middle sample of an odd-length window
 *       (libs/lpc/src/lpc.c:200-204), so the autocorrelation reads what an earlier window call left
 *       in the calculator's single buffer.
 *   Q2  LPCCalculator_EstimateCodeLength sums parcor_coef[1..order] (lpc.c:846-848) but the
 *       Levinson recursion only writes parcor_coef[0..order-1] (lpc.c:287,315): element [order] is
 *       whatever the previous, higher-order call left there.
 */
#include "linne_oracle.h"
#include "linne_oracle_freq.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------------------------------------------------------------------------------------
 * format constants (libs/linne_internal/include/linne_internal.h:8-35, include/linne.h:6-14)
 * ------------------------------------------------------------------------------------------- */
#define FORMAT_VERSION          1u
#define CODEC_VERSION           2u
#define HEADER_SIZE             30u
#define BLOCK_SYNC              0xFFFFu
#define PREEM_SHIFT             5
#define COEF_BITWIDTH           8u
#define LOG2_UNITS_BITWIDTH     3u
#define RSHIFT_BITWIDTH         4u
#define RAW_THRESHOLD           0.95f      /* LINNE_ESTIMATED_CODELENGTH_THRESHOLD (float literal) */
#define NUM_PRESETS             8u
#define RICE_LOG2_MAX_PARTS     10u
#define RICE_MAX_PARTS          (1u << RICE_LOG2_MAX_PARTS)
#define RICE_PARAM_BITS         5u
#define LPC_PI                  3.1415926535897932384626433832795029

/* presets: libs/linne_internal/src/linne_internal.c:16-41 */
static const uint32_t k_layers_a[] = { 2, 32 };
static const uint32_t k_layers_b[] = { 4, 64, 8 };
static const uint32_t k_layers_c[] = { 4, 128, 16 };
static const double k_regs_1[] = { 0.0 };
static const double k_regs_2[] = { 0.0, 1.0 / 512.0 };
static const double k_regs_4[] = { 0.0, 1.0 / 2048.0, 1.0 / 512.0, 1.0 / 128.0 };

struct Preset { uint32_t num_layers; const uint32_t *layers; uint32_t num_regs; const double *regs; };
static const struct Preset k_presets[NUM_PRESETS] = {
    { 2, k_layers_a, 1, k_regs_1 }, { 2, k_layers_a, 2, k_regs_2 },
    { 3, k_layers_b, 1, k_regs_1 }, { 3, k_layers_b, 2, k_regs_2 }, { 3, k_layers_b, 4, k_regs_4 },
    { 3, k_layers_c, 1, k_regs_1 }, { 3, k_layers_c, 2, k_regs_2 }, { 3, k_layers_c, 4, k_regs_4 },
};

/* zig-zag maps: libs/linne_internal/include/linne_utility.h:33-35 */
static uint32_t zigzag(int32_t v) { const uint32_t d = (uint32_t)v << 1; return (v < 0) ? ((0u - d) - 1u) : d; }
static int32_t unzigzag(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1u); }
/* ceil(log2(x)): linne_utility.h:55 (32 - nlz(x - 1)) */
static uint32_t log2ceil(uint32_t x) { uint32_t y = x - 1u; return (y == 0) ? 0u : 32u - (uint32_t)__builtin_clz(y); }
/* round half away from zero: lpc.c:49-52, linne_utility.c:58-61 */
static double round_away(double d) { return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5); }
/* log2 through natural log: lpc.c:55-60, linne_utility.c:64-69 */
static double log2_via_ln(double d) { return log(d) * 1.4426950408889634; }

/* ---------------------------------------------------------------------------------------------
 * CRC16-IBM (reflected 0xA001, init 0): libs/linne_internal/src/linne_utility.c:72-89
 * ------------------------------------------------------------------------------------------- */
uint16_t oracle_crc16(const uint8_t *data, uint64_t size)
{
    static uint16_t table[256];
    static int ready = 0;
    uint16_t crc = 0;
    if (!ready) {
        uint32_t i, b;
        for (i = 0; i < 256; i++) {
            uint16_t c = (uint16_t)i;
            for (b = 0; b < 8; b++) c = (uint16_t)((c & 1u) ? ((c >> 1) ^ 0xA001u) : (c >> 1));
            table[i] = c;
        }
        ready = 1;
    }
    while (size--) crc = (uint16_t)((crc >> 8) ^ table[(crc ^ *data++) & 0xFFu]);
    return crc;
}

/* ---------------------------------------------------------------------------------------------
 * MSB-first bit I/O (behaviour of libs/bit_stream/include/bit_stream.h:240-433, own structure)
 * ------------------------------------------------------------------------------------------- */
struct BitW { uint8_t *p; uint8_t *end; uint64_t acc; uint32_t n; int overflow; uint8_t *base; };
static void bw_open(struct BitW *w, uint8_t *mem, uint32_t size) { w->p = w->base = mem; w->end = mem + size; w->acc = 0; w->n = 0; w->overflow = 0; }
static void bw_put(struct BitW *w, uint32_t val, uint32_t nbits)
{
    if (nbits == 0) return;
    if (nbits < 32) val &= (1u << nbits) - 1u;
    w->acc = (w->acc << nbits) | val; w->n += nbits;
    while (w->n >= 8) {
        if (w->p >= w->end) { w->overflow = 1; w->n -= 8; continue; }
        *w->p++ = (uint8_t)(w->acc >> (w->n - 8)); w->n -= 8;
    }
}
/* zero run then a terminating 1 (bit_stream.h:285-302) */
static void bw_put_zero_run(struct BitW *w, uint32_t run) { while (run >= 31) { bw_put(w, 0, 31); run -= 31; } bw_put(w, 1, run + 1); }
static void bw_flush(struct BitW *w) { if (w->n) bw_put(w, 0, 8 - w->n); }
static uint32_t bw_tell(const struct BitW *w) { return (uint32_t)(w->p - w->base); }

struct BitR { const uint8_t *base; uint64_t size_bits; uint64_t pos; };
static void br_open(struct BitR *r, const uint8_t *mem, uint32_t size) { r->base = mem; r->size_bits = (uint64_t)size * 8u; r->pos = 0; }
static uint32_t br_bit(struct BitR *r)
{
    uint32_t b = 0;
    if (r->pos < r->size_bits) b = (r->base[r->pos >> 3] >> (7u - (uint32_t)(r->pos & 7u))) & 1u;
    r->pos++;
    return b;
}
static uint32_t br_get(struct BitR *r, uint32_t nbits) { uint32_t v = 0; while (nbits--) v = (v << 1) | br_bit(r); return v; }
static uint32_t br_zero_run(struct BitR *r) { uint32_t run = 0; while (r->pos < r->size_bits && br_bit(r) == 0) run++; return run; }
static uint32_t br_tell_bytes(const struct BitR *r) { return (uint32_t)((r->pos + 7u) >> 3); }

/* ---------------------------------------------------------------------------------------------
 * static Huffman code from the fixed frequency table:
 * libs/static_huffman/src/static_huffman.c:28-92 (tree), :95-131 (codes), :145-165 (decode)
 * ------------------------------------------------------------------------------------------- */
struct Huff { uint32_t root; uint32_t child0[512], child1[512]; uint32_t code[256]; uint8_t len[256]; };
static void huff_assign(struct Huff *h, uint32_t node, uint32_t code, uint8_t len)
{
    if (node < 256) { h->code[node] = code; h->len[node] = len; return; }
    huff_assign(h, h->child0[node], (code << 1) | 0u, (uint8_t)(len + 1));
    huff_assign(h, h->child1[node], (code << 1) | 1u, (uint8_t)(len + 1));
}
static void huff_build(struct Huff *h)
{
    uint32_t count[513];
    uint32_t free_node, node;
    memset(count, 0, sizeof(count));
    for (node = 0; node < 256; node++) count[node] = oracle_coef_freq[node] ? oracle_coef_freq[node] : 1u;
    count[512] = UINT32_MAX;                       /* sentinel */
    for (free_node = 256; ; free_node++) {
        uint32_t min1 = 512, min2 = 512;
        for (node = 0; node < free_node; node++) {
            if (count[node] > 0) {
                if (count[node] < count[min1]) { min2 = min1; min1 = node; }
                else if (count[node] < count[min2]) { min2 = node; }
            }
        }
        if (min2 == 512) break;
        count[free_node] = count[min1] + count[min2];
        count[min1] = count[min2] = 0;
        h->child0[free_node] = min1; h->child1[free_node] = min2;
    }
    h->root = free_node - 1;
    huff_assign(h, h->root, 0, 0);
}
static uint32_t huff_get(const struct Huff *h, struct BitR *r)
{
    uint32_t node = h->root;
    do { node = br_bit(r) ? h->child1[node] : h->child0[node]; } while (node >= 256);
    return node;
}
static struct Huff g_huff; static pthread_once_t g_huff_once = PTHREAD_ONCE_INIT;
static void huff_init_once(void) { huff_build(&g_huff); }
uint32_t oracle_huffman_code(uint32_t sym, uint32_t *code)
{
    pthread_once(&g_huff_once, huff_init_once);
    if (code) *code = g_huff.code[sym & 255u];
    return g_huff.len[sym & 255u];
}

/* ---------------------------------------------------------------------------------------------
 * partitioned recursive Rice coder: libs/linne_coder/src/linne_coder.c
 * ------------------------------------------------------------------------------------------- */
/* linne_coder.c:172-200 (only k1/k2 are used by the codec) */
static void rice_param(double mean, uint32_t *k1, uint32_t *k2)
{
    const double optx = 0.5127629514437670454896078808815218508243560791015625;
    const double rho = 1.0 / (1.0 + mean);
    const double t = floor(log2_via_ln(log(optx) / log(1.0 - rho)));
    *k2 = (uint32_t)((0 > t) ? 0 : t);
    *k1 = *k2 + 1;
}
/* linne_coder.c:203-214 (k1 = 32, i.e. a partition mean above 3.2e9, shifts by the type's width there and here: undefined; the
 * product takes the count modulo 32 -- see DESIGN 5) */
static uint32_t rice_len(uint32_t k1, uint32_t k2, uint32_t u)
{
    const uint32_t k1pow = 1u << k1;
    return (u < k1pow) ? (k1 + 1) : (k2 + 2 + ((u - k1pow) >> k2));
}
/* linne_coder.c:130-148 */
static void rice_put(struct BitW *w, uint32_t k1, uint32_t k2, uint32_t u)
{
    const uint32_t k1pow = 1u << k1;
    if (u < k1pow) { bw_put(w, 1, 1); bw_put(w, u, k1); }
    else { u -= k1pow; bw_put_zero_run(w, 1 + (u >> k2)); bw_put(w, u & ((1u << k2) - 1u), k2); }
}
/* linne_coder.c:151-169 */
static uint32_t rice_get(struct BitR *r, uint32_t k1, uint32_t k2)
{
    const uint32_t quot = br_zero_run(r);
    if (quot == 0) return br_get(r, k1);
    return br_get(r, k2) + (1u << k1) + ((quot - 1) << k2);
}
/* gamma code: linne_coder.c:86-127 */
static void gamma_put(struct BitW *w, uint32_t v)
{
    uint32_t nd;
    if (v == 0) { bw_put(w, 1, 1); return; }
    nd = log2ceil(v + 2);
    bw_put(w, 0, nd - 1); bw_put(w, v + 1, nd);
}
static uint32_t gamma_get(struct BitR *r)
{
    uint32_t nd = br_zero_run(r) + 1;
    if (nd == 1) return 0;
    return (uint32_t)((1ul << (nd - 1)) + br_get(r, nd - 1) - 1);
}
static uint32_t gamma_bits(uint32_t u) { return (u == 0) ? 1u : (2u * log2ceil(u + 2) - 1u); }

/* linne_coder.c:217-303 */
static void rice_encode_core(struct BitW *w, const int32_t *data, uint32_t n, double (*part_mean)[RICE_MAX_PARTS])
{
    uint32_t max_porder = 1, max_parts, porder, part, smpl, best_porder = 0, min_bits = UINT32_MAX;
    int32_t i;
    while ((n % (1u << max_porder)) == 0) max_porder++;
    max_porder = (max_porder - 1 < RICE_LOG2_MAX_PARTS) ? (max_porder - 1) : RICE_LOG2_MAX_PARTS;
    max_parts = 1u << max_porder;
    for (part = 0; part < max_parts; part++) {
        const uint32_t ns = n / max_parts;
        double sum = 0.0;
        for (smpl = 0; smpl < ns; smpl++) sum += zigzag(data[part * ns + smpl]);
        part_mean[max_porder][part] = sum / ns;
    }
    for (i = (int32_t)max_porder - 1; i >= 0; i--)
        for (part = 0; part < (1u << i); part++)
            part_mean[i][part] = (part_mean[i + 1][2 * part] + part_mean[i + 1][2 * part + 1]) / 2.0;
    for (porder = 0; porder <= max_porder; porder++) {
        const uint32_t ns = n >> porder;
        uint32_t k1, k2, prevk2 = 0, bits = 0;
        for (part = 0; part < (1u << porder); part++) {
            rice_param(part_mean[porder][part], &k1, &k2);
            for (smpl = 0; smpl < ns; smpl++) bits += rice_len(k1, k2, zigzag(data[part * ns + smpl]));
            if (part == 0) bits += RICE_PARAM_BITS;
            else bits += gamma_bits(zigzag((int32_t)k2 - (int32_t)prevk2));
            prevk2 = k2;
        }
        if (min_bits > bits) { min_bits = bits; best_porder = porder; }
    }
    {
        const uint32_t ns = n >> best_porder;
        uint32_t k1, k2, prevk2 = 0;
        bw_put(w, best_porder, RICE_LOG2_MAX_PARTS);
        for (part = 0; part < (1u << best_porder); part++) {
            rice_param(part_mean[best_porder][part], &k1, &k2);
            if (part == 0) bw_put(w, k2, RICE_PARAM_BITS);
            else gamma_put(w, zigzag((int32_t)k2 - (int32_t)prevk2));
            prevk2 = k2;
            for (smpl = 0; smpl < ns; smpl++) rice_put(w, k1, k2, zigzag(data[part * ns + smpl]));
        }
    }
}
/* linne_coder.c:306-327 */
static void rice_decode_core(struct BitR *r, int32_t *data, uint32_t n)
{
    uint32_t smpl, part, ns, best_porder, k1, k2 = 0;
    best_porder = br_get(r, RICE_LOG2_MAX_PARTS);
    ns = n >> best_porder;
    for (part = 0; part < (1u << best_porder); part++) {
        if (part == 0) k2 = br_get(r, RICE_PARAM_BITS);
        else k2 = (uint32_t)((int32_t)k2 + unzigzag(gamma_get(r)));
        k1 = k2 + 1;
        for (smpl = 0; smpl < ns; smpl++) data[part * ns + smpl] = unzigzag(rice_get(r, k1, k2));
    }
}
uint32_t oracle_rice_encode(const int32_t *data, uint32_t num_samples, uint8_t *out, uint32_t out_size)
{
    struct BitW w;
    double (*pm)[RICE_MAX_PARTS] = malloc(sizeof(double) * (RICE_LOG2_MAX_PARTS + 1) * RICE_MAX_PARTS);
    bw_open(&w, out, out_size);
    rice_encode_core(&w, data, num_samples, pm);
    bw_flush(&w);
    free(pm);
    return w.overflow ? 0 : bw_tell(&w);
}
uint32_t oracle_rice_decode(const uint8_t *in, uint32_t in_size, int32_t *data, uint32_t num_samples)
{
    struct BitR r;
    br_open(&r, in, in_size);
    rice_decode_core(&r, data, num_samples);
    return br_tell_bytes(&r);
}

/* ---------------------------------------------------------------------------------------------
 * LPC numerics: libs/lpc/src/lpc.c.  One calculator state per encoder (as the reference:
 * linne_network.c:462-472), including the shared window buffer (Q1) and parcor buffer (Q2).
 * ------------------------------------------------------------------------------------------- */
struct Lpc {
    uint32_t max_order, max_samples;
    double *a, *u, *v;          /* order + 2 each */
    double *auto_corr, *lpc_coef, *parcor;   /* order + 1 each */
    double *buffer;             /* window output, max_samples */
    double **r_mat;             /* (order + 1)^2, the auxiliary-function method's normal matrix (lpc.c:41) */
};
enum { WINDOW_SIN, WINDOW_WELCH };

/* lpc.c:176-212 */
static void lpc_window(int type, const double *in, uint32_t n, double *out)
{
    uint32_t s;
    if (type == WINDOW_SIN) {
        for (s = 0; s < n; s++) out[s] = in[s] * sin((LPC_PI * s) / (n - 1));
    } else {
        const double divisor = 4.0 * pow(n - 1, -2.0);
        for (s = 0; s < (n >> 1); s++) {
            const double weight = divisor * s * (n - 1 - s);
            out[s] = in[s] * weight;
            out[n - s - 1] = in[n - s - 1] * weight;
        }
        /* Q1: for odd n the element (n-1)/2 keeps its previous content */
    }
}
/* lpc.c:215-249; num_lags = order + 1.  Each lag is one chain over increasing i starting from +0.0. */
static void lpc_autocorr(const double *x, uint32_t n, double *r, uint32_t num_lags)
{
    uint32_t i, lag;
    for (lag = 0; lag < num_lags; lag++) r[lag] = 0.0;
    if (n < num_lags) {                                /* out of the reference's contract (lpc.c:221,234) */
        for (i = 0; i < n; i++) for (lag = 0; lag < n - i; lag++) r[lag] += x[i] * x[i + lag];
        return;
    }
    for (i = 0; i <= n - num_lags; i++) {
        const double t = x[i];
        for (lag = 0; lag < num_lags; lag++) r[lag] += t * x[i + lag];
    }
    for (; i < n; i++) {
        const double t = x[i];
        for (lag = 0; lag < n - i; lag++) r[lag] += t * x[i + lag];
    }
}
/* lpc.c:252-324 */
static void lpc_levinson(struct Lpc *c, const double *r, uint32_t order, double *coef, double *parcor)
{
    uint32_t k, i;
    double gamma, ek;
    double *a = c->a, *u = c->u, *v = c->v;
    if (fabs(r[0]) < FLT_EPSILON) {
        for (i = 0; i < order + 1; i++) coef[i] = parcor[i] = 0.0;
        return;
    }
    for (i = 0; i < order + 2; i++) a[i] = u[i] = v[i] = 0.0;
    a[0] = 1.0;
    ek = r[0];
    a[1] = -r[1] / r[0];
    parcor[0] = r[1] / ek;
    ek += r[1] * a[1];
    u[0] = 1.0; u[1] = 0.0;
    v[0] = 0.0; v[1] = 1.0;
    for (k = 1; k < order; k++) {
        gamma = 0.0;
        for (i = 0; i < k + 1; i++) gamma += a[i] * r[k + 1 - i];
        gamma /= -ek;
        ek *= (1.0 - gamma * gamma);
        for (i = 0; i < k; i++) u[i + 1] = v[k - i] = a[i + 1];
        u[0] = 1.0; u[k + 1] = 0.0;
        v[0] = 0.0; v[k + 1] = 1.0;
        for (i = 0; i < k + 2; i++) a[i] = u[i] + gamma * v[i];
        parcor[k] = -gamma;
    }
    memcpy(coef, &a[1], sizeof(double) * order);
}
/* lpc.c:327-366 */
static void lpc_calc(struct Lpc *c, const double *data, uint32_t n, uint32_t order, int window, double reg)
{
    uint32_t i;
    lpc_window(window, data, n, c->buffer);
    lpc_autocorr(c->buffer, n, c->auto_corr, order + 1);
    if (n < order) {
        for (i = 0; i < order + 1; i++) c->lpc_coef[i] = c->parcor[i] = 0.0;
        return;
    }
    c->auto_corr[0] *= (1.0 + reg);
    lpc_levinson(c, c->auto_corr, order, c->lpc_coef, c->parcor);
}
/* LPC_CholeskyDecomposition, lpc.c:402-448: solves A x = b in place (A symmetric positive definite, upper triangle in,
 * factor written below the diagonal); inv_diag[i] = pow(sum, -0.5) from libm.  Returns -1 for a non-positive pivot. */
static int lpc_cholesky(double **A, int32_t dim, double *x, const double *b, double *inv_diag)
{
    int32_t i, j, k;
    double sum;
    for (i = 0; i < dim; i++) {
        sum = A[i][i];
        for (k = i - 1; k >= 0; k--) sum -= A[i][k] * A[i][k];
        if (sum <= 0.0) return -1;
        inv_diag[i] = pow(sum, -0.5);
        for (j = i + 1; j < dim; j++) {
            sum = A[i][j];
            for (k = i - 1; k >= 0; k--) sum -= A[i][k] * A[j][k];
            A[j][i] = sum * inv_diag[i];
        }
    }
    for (i = 0; i < dim; i++) {
        sum = b[i];
        for (j = i - 1; j >= 0; j--) sum -= A[i][j] * x[j];
        x[i] = sum * inv_diag[i];
    }
    for (i = dim - 1; i >= 0; i--) {
        sum = x[i];
        for (j = i + 1; j < dim; j++) sum -= A[j][i] * x[j];
        x[i] = sum * inv_diag[i];
    }
    return 0;
}
/* LPCAF_CalculateCoefMatrixAndVector (forward residual form, the one compiled: lpc.c:451-509) */
#define LPCAF_RESIDUAL_EPSILON 1e-6
static double lpc_af_matrix(const double *data, uint32_t n, const double *a, double **R, double *rv, uint32_t order)
{
    uint32_t s, i, j;
    double obj = 0.0;
    for (i = 0; i < order; i++) { rv[i] = 0.0; for (j = 0; j < order; j++) R[i][j] = 0.0; }
    for (s = order; s < n; s++) {
        double residual = data[s], inv;
        for (i = 0; i < order; i++) residual += a[i] * data[s - i - 1];
        residual = fabs(residual);
        obj += residual;
        residual = (residual < LPCAF_RESIDUAL_EPSILON) ? LPCAF_RESIDUAL_EPSILON : residual;
        inv = 1.0 / residual;
        for (i = 0; i < order; i++) {
            rv[i] -= data[s] * data[s - i - 1] * inv;
            for (j = i; j < order; j++) R[i][j] += data[s - i - 1] * data[s - j - 1] * inv;
        }
    }
    for (i = 0; i < order; i++) for (j = i + 1; j < order; j++) R[j][i] = R[i][j];
    return obj / (n - order);
}
/* LPCCalculator_CalculateLPCCoefficientsAF -> LPC_CalculateCoefAF, lpc.c:578-661: Levinson-Durbin start, then up to max_iter
 * rounds of the auxiliary-function (IRLS for the L1 norm) update; the solve writes a_vec from r_vec = u_vec with v_vec as the
 * inverse diagonal, as the reference does (they are Levinson's work vectors otherwise) */
static void lpc_calc_af(struct Lpc *c, const double *data, uint32_t n, double *coef, uint32_t order, uint32_t max_iter, double reg)
{
    uint32_t i, itr;
    double obj, prev_obj;
    lpc_calc(c, data, n, order, WINDOW_WELCH, reg);
    memcpy(c->a, c->lpc_coef, sizeof(double) * order);
    if (fabs(c->auto_corr[0]) < FLT_EPSILON) {
        for (i = 0; i < order + 1; i++) c->lpc_coef[i] = 0.0;
        memmove(coef, c->lpc_coef, sizeof(double) * order);
        return;
    }
    prev_obj = FLT_MAX;
    for (itr = 0; itr < max_iter; itr++) {
        obj = lpc_af_matrix(data, n, c->a, c->r_mat, c->u, order);
        if (lpc_cholesky(c->r_mat, (int32_t)order, c->a, c->u, c->v) != 0) {
            for (i = 0; i < order; i++) c->lpc_coef[i] = 0.0;
            memmove(coef, c->lpc_coef, sizeof(double) * order);
            return;
        }
        if (fabs(prev_obj - obj) < 1e-8) break;
        prev_obj = obj;
    }
    memmove(c->lpc_coef, c->a, sizeof(double) * order);
    memmove(coef, c->lpc_coef, sizeof(double) * order);
}
static void lpc_calc_af0(struct Lpc *c, const double *data, uint32_t n, double *coef, uint32_t order, double reg)
{
    lpc_calc_af(c, data, n, coef, order, 0, reg);           /* LINNE_NUM_AF_METHOD_ITERATION_DETERMINEUNIT = 0 (linne_internal.h:26) */
}
/* lpc.c:810-865 (SIN window, regulariser 0) */
static double lpc_estimate_code_length(struct Lpc *c, const double *data, uint32_t n, uint32_t bits, uint32_t order)
{
    uint32_t ord;
    double p, ratio, len;
    lpc_calc(c, data, n, order, WINDOW_SIN, 0.0);
    p = c->auto_corr[0];
    p *= pow(2, (double)(2.0 * (bits - 1)));
    if (fabs(p) <= FLT_MIN) return 0.0;
    p = log2_via_ln(p) - log2_via_ln((double)n);
    ratio = 0.0;
    for (ord = 1; ord <= order; ord++) ratio += log2_via_ln(1.0 - c->parcor[ord] * c->parcor[ord]);   /* Q2 */
    len = 1.9426950408889634 + 0.5f * (p + ratio);
    if (len <= 0) return 1.0;
    return len;
}
/* lpc.c:981-1040 */
static void lpc_quantize(const double *dcoef, uint32_t order, uint32_t nbits, int32_t *icoef, uint32_t *rshift_out)
{
    uint32_t rshift;
    int32_t ord, ndigit, q;
    double max = 0.0, qerror;
    const int32_t qmax = 1 << (nbits - 1);
    for (ord = 0; ord < (int32_t)order; ord++) if (max < fabs(dcoef[ord])) max = fabs(dcoef[ord]);
    if (max <= pow(2.0, -(int32_t)(nbits - 1))) {
        *rshift_out = nbits;
        memset(icoef, 0, sizeof(int32_t) * order);
        return;
    }
    (void)frexp(max, &ndigit);
    nbits--;
    rshift = (uint32_t)((int32_t)nbits - ndigit);
    qerror = 0.0;
    for (ord = (int32_t)order - 1; ord >= 0; ord--) {
        qerror += dcoef[ord] * pow(2.0, rshift);
        q = (int32_t)round_away(qerror);
        if (q >= qmax) q = qmax - 1; else if (q < -qmax) q = -qmax;
        qerror -= q;
        icoef[ord] = q;
    }
    *rshift_out = rshift;
}

/* ---------------------------------------------------------------------------------------------
 * the layer cascade: libs/linne_network/src/linne_network.c
 * ------------------------------------------------------------------------------------------- */
struct Layer { double *din; double *params; uint32_t num_params, num_units; double *dout, *dparams, *momentum; };

/* linne_network.c:50-63 */
static double l1_loss(const double *d, uint32_t n)
{
    uint32_t s; double norm = 0.0f;
    for (s = 0; s < n; s++) norm += fabs(d[s]);
    return norm / n;
}
/* linne_network.c:165-210 */
static void layer_forward(struct Layer *L, double *data, uint32_t n)
{
    uint32_t unit, i, j;
    const uint32_t ns = n / L->num_units, np = L->num_params / L->num_units;
    memcpy(L->din, data, sizeof(double) * n);
    for (unit = 0; unit < L->num_units; unit++) {
        const double *h = &L->params[unit * np];
        const double *pin = &L->din[unit * ns];
        double *res = &data[unit * ns];
        double predict;
        i = 0;
        if (unit == 0) {
            for (i = 1; i < np; i++) {
                predict = 0.0f;
                for (j = 0; j < i; j++) predict += h[np - i + j] * pin[j];
                res[i] += predict;
            }
        }
        for (; i < ns; i++) {
            predict = 0.0f;
            for (j = 0; j < np; j++) predict += h[j] * pin[(int32_t)(i - np + j)];
            res[i] += predict;
        }
    }
}
/* TEST / MEASUREMENT TAP (tools/search_margins.py): when set, every unit-count search appends one record per trial --
 * { num_params, nunits, mean loss, largest L1 norm of a unit's coefficients, max |input| } -- so that the margins between the
 * trials (what a certificate of the argmin has to beat) can be read off without touching the arithmetic */
static __thread double *g_trial_tap = NULL;
static __thread uint32_t g_trial_tap_cap = 0, g_trial_tap_n = 0;
void oracle_set_trial_tap(double *buf, uint32_t cap_records) { g_trial_tap = buf; g_trial_tap_cap = cap_records; g_trial_tap_n = 0; }
uint32_t oracle_trial_tap_count(void) { return g_trial_tap_n; }
/* linne_network.c:268-347 */
static uint32_t layer_search_units(struct Layer *L, struct Lpc *c, const double *input, uint32_t n, uint32_t max_units, double reg)
{
    uint32_t unit, nunits, best = 0;
    double min_loss = FLT_MAX;
    for (nunits = 1; nunits <= max_units; nunits <<= 1) {
        const uint32_t np = L->num_params / nunits, ns = n / nunits;
        double mean_loss = 0.0f;
        if ((L->num_params % nunits) != 0 || (n % nunits) != 0) continue;
        for (unit = 0; unit < nunits; unit++) {
            uint32_t s, k;
            const double *pin = &input[unit * ns];
            double *h = &L->params[unit * np];
            double res;
            lpc_calc_af0(c, pin, ns, h, np, reg);
            for (k = 0; k < np / 2; k++) { double t = h[k]; h[k] = h[np - k - 1]; h[np - k - 1] = t; }
            s = 0;
            if (unit == 0) {
                for (s = 1; s < np; s++) {
                    res = pin[s];
                    for (k = 0; k < s; k++) res += h[np - s + k] * pin[k];
                    mean_loss += (res > 0) ? res : -res;
                }
            }
            for (; s < ns; s++) {
                res = pin[s];
                for (k = 0; k < np; k++) res += h[k] * pin[(int32_t)(s - np + k)];
                mean_loss += (res > 0) ? res : -res;
            }
        }
        mean_loss /= n;
        if (g_trial_tap && g_trial_tap_n < g_trial_tap_cap) {
            double *rec = g_trial_tap + 5 * (size_t)g_trial_tap_n++, hmax = 0.0, xmax = 0.0;
            uint32_t k;
            for (unit = 0; unit < nunits; unit++) { double a = 0.0; for (k = 0; k < np; k++) a += fabs(L->params[unit * np + k]); if (a > hmax) hmax = a; }
            for (k = 0; k < n; k++) if (fabs(input[k]) > xmax) xmax = fabs(input[k]);
            rec[0] = L->num_params; rec[1] = nunits; rec[2] = mean_loss; rec[3] = hmax; rec[4] = xmax;
        }
        if (mean_loss < min_loss) { min_loss = mean_loss; best = nunits; }
    }
    return best;
}
/* linne_network.c:350-376 */
static void layer_set_parameter(struct Layer *L, struct Lpc *c, const double *input, uint32_t n, uint32_t af_iters, double reg)
{
    uint32_t i, unit;
    const uint32_t np = L->num_params / L->num_units, ns = n / L->num_units;
    for (unit = 0; unit < L->num_units; unit++) {
        double *h = &L->params[unit * np];
        lpc_calc_af(c, &input[unit * ns], ns, h, np, af_iters, reg);
        for (i = 0; i < np / 2; i++) { double t = h[i]; h[i] = h[np - i - 1]; h[np - i - 1] = t; }
    }
}

/* linne_network.c:213-265: gradients of one layer; data holds the gradient w.r.t. the layer's output on entry and w.r.t. its
 * input on return */
static void layer_backward(struct Layer *L, double *data, uint32_t n)
{
    uint32_t unit, i, j;
    const uint32_t ns = n / L->num_units, np = L->num_params / L->num_units;
    memcpy(L->dout, data, sizeof(double) * n);
    for (unit = 0; unit < L->num_units; unit++) {
        const double *pin = &L->din[unit * ns], *pout = &L->dout[unit * ns], *h = &L->params[unit * np];
        double *pback = &data[unit * ns], *pd = &L->dparams[unit * np];
        for (i = 0; i < np; i++) {
            pd[i] = 0.0f;
            for (j = 0; j < (ns - np + i); j++) pd[i] += pin[j] * pout[np - i + j];
        }
        for (i = 0; i < (ns - np); i++) {
            double back = 0.0f;
            for (j = 0; j < np; j++) back += h[j] * pout[np + i - j];
            pback[i] += back / np;
        }
        for (; i < ns; i++) {
            double back = 0.0f;
            for (j = 0; j < np; j++) if ((np + i - j) < ns) back += h[j] * pout[np + i - j];
            pback[i] += back / np;
        }
    }
}
/* LINNENetworkTrainer_Train, linne_network.c:805-873 (momentum SGD on the L1 loss; `-l`), with LINNENetwork_CalculateGradient
 * (:558-579) and LINNEL1Norm_Backward (:66-75) */
static void network_train(struct OracleEncoder *e, const double *input, uint32_t n, uint32_t max_iter, double lr, double eps);

/* ---------------------------------------------------------------------------------------------
 * integer filters and channel utilities
 * ------------------------------------------------------------------------------------------- */
/* libs/linne_internal/src/linne_utility.c:120-132 */
static void ms_conversion(int32_t *c0, int32_t *c1, uint32_t n)
{
    uint32_t s;
    for (s = 0; s < n; s++) {
        c1[s] = (int32_t)((uint32_t)c1[s] - (uint32_t)c0[s]);
        c0[s] = (int32_t)((uint32_t)c0[s] + (uint32_t)(c1[s] >> 1));
    }
}
/* linne_utility.c:135-147 */
static void lr_conversion(int32_t *c0, int32_t *c1, uint32_t n)
{
    uint32_t s;
    for (s = 0; s < n; s++) {
        c0[s] = (int32_t)((uint32_t)c0[s] - (uint32_t)(c1[s] >> 1));
        c1[s] = (int32_t)((uint32_t)c1[s] + (uint32_t)c0[s]);
    }
}
/* linne_utility.c:158-193 */
static int32_t preem_coefficient(const int32_t *buf, uint32_t n)
{
    uint32_t s; int32_t coef;
    double corr[2] = { 0.0, 0.0 }, curr;
    curr = buf[0];
    for (s = 0; s + 1 < n; s++) {
        const double succ = buf[s + 1];
        corr[0] += curr * curr;
        corr[1] += curr * succ;
        curr = succ;
    }
    corr[1] /= corr[0];
    if ((corr[0] < 1e-6) || (corr[1] < 0.0)) {
        coef = 0;
    } else {
        coef = (int32_t)round_away(corr[1] * pow(2.0f, PREEM_SHIFT));
        if (coef >= (1 << (PREEM_SHIFT - 1))) coef = (1 << (PREEM_SHIFT - 1)) - 1;
    }
    return coef;
}
/* linne_utility.c:196-212 */
static void preemphasis(int32_t prev, int32_t coef, int32_t *buf, uint32_t n)
{
    uint32_t s;
    for (s = 0; s < n; s++) {
        const int32_t t = buf[s];
        buf[s] = (int32_t)((uint32_t)buf[s] - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> PREEM_SHIFT));
        prev = t;
    }
}
/* linne_utility.c:215-241 (two cascaded de-emphasis stages fused in one loop) */
static void deemphasis2(const int32_t prev[2], const int32_t coef[2], int32_t *b, uint32_t n)
{
    uint32_t s;
    const int32_t c0 = coef[0], c1 = coef[1];
#define MULSHR(x, c) ((int32_t)((uint32_t)(x) * (uint32_t)(c)) >> PREEM_SHIFT)
    b[0] += MULSHR(prev[1], c1);
    if (n > 1) b[1] += MULSHR(b[0], c1);
    b[0] += MULSHR(prev[0], c0);
    for (s = 2; s < n; s++) {
        b[s] += MULSHR(b[s - 1], c1);
        b[s - 1] += MULSHR(b[s - 2], c0);
    }
    if (n > 1) b[n - 1] += MULSHR(b[n - 2], c0);
#undef MULSHR
}
/* libs/linne_encoder/src/linne_lpc_predict.c:7-38 (int32 wrap-around arithmetic) */
static void lpc_predict(const int32_t *data, uint32_t n, const int32_t *coef, uint32_t order, int32_t *res, uint32_t rshift, uint32_t units)
{
    uint32_t u, s, k;
    const uint32_t half = 1u << ((rshift - 1u) & 31u);
    const uint32_t np = order / units, ns = n / units;
    memcpy(res, data, sizeof(int32_t) * n);
    if (ns < np) return;                           /* reference would run off the buffer */
    for (u = 0; u < units; u++) {
        const int32_t *in = &data[u * ns];
        int32_t *out = &res[u * ns];
        const int32_t *c = &coef[u * np];
        for (s = 0; s < ns - np; s++) {
            uint32_t pred = half;
            for (k = 0; k < np; k++) pred += (uint32_t)c[k] * (uint32_t)in[s + k];
            out[s + np] = (int32_t)((uint32_t)out[s + np] + (uint32_t)((int32_t)pred >> (rshift & 31u)));
        }
    }
}
/* libs/linne_decoder/src/linne_lpc_synthesize.c:8-83 (units are independent; the 2/4-way interleave of
 * the reference is an ILP device only) */
static void lpc_synthesize(int32_t *data, uint32_t n, const int32_t *coef, uint32_t order, uint32_t rshift, uint32_t units)
{
    uint32_t u, s, k;
    const uint32_t half = 1u << ((rshift - 1u) & 31u);
    const uint32_t np = order / units, ns = n / units;
    if (ns < np) return;
    for (u = 0; u < units; u++) {
        int32_t *d = &data[u * ns];
        const int32_t *c = &coef[u * np];
        for (s = 0; s < ns - np; s++) {
            uint32_t pred = half;
            for (k = 0; k < np; k++) pred += (uint32_t)c[k] * (uint32_t)d[s + k];
            d[s + np] = (int32_t)((uint32_t)d[s + np] - (uint32_t)((int32_t)pred >> (rshift & 31u)));
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * encoder: libs/linne_encoder/src/linne_encoder.c
 * ------------------------------------------------------------------------------------------- */
struct OracleEncoder {
    struct OracleEncodeParameter p;
    const struct Preset *preset;
    uint32_t block, max_params;
    struct Lpc lpc;
    struct Layer layer[ORACLE_MAX_LAYERS];
    double *data_buffer;            /* network data buffer (linne_network.c:30) */
    double *buffer_double;
    int32_t *buffer_int[ORACLE_MAX_CHANNELS];
    int32_t *residual[ORACLE_MAX_CHANNELS];
    double (*part_mean)[RICE_MAX_PARTS];
    uint32_t learning;              /* enable_learning (linne_encoder.c:459), 0 unless oracle_encoder_set_learning */
    uint32_t af_iters;              /* num_afmethod_iterations (linne_encoder.c:462), 0 unless oracle_encoder_set_af_iterations */
    void *arena;
};

static void *arena_take(uint8_t **p, size_t bytes) { void *r = *p; *p += (bytes + 15u) & ~(size_t)15u; return r; }

struct OracleEncoder *oracle_encoder_create(const struct OracleEncodeParameter *param)
{
    struct OracleEncoder *e;
    uint32_t l, ch, maxp = 0;
    size_t total;
    uint8_t *w;
    const struct Preset *ps;
    if (!param || param->preset >= NUM_PRESETS || param->num_channels == 0 || param->num_channels > ORACLE_MAX_CHANNELS
            || param->bits_per_sample == 0 || param->sampling_rate == 0 || param->num_samples_per_block == 0
            || param->ch_process_method > 1 || (param->ch_process_method == 1 && param->num_channels == 1)) return NULL;
    ps = &k_presets[param->preset];
    for (l = 0; l < ps->num_layers; l++) {
        if (param->num_samples_per_block <= ps->layers[l]) return NULL;      /* linne_encoder.c:176-181 */
        if (maxp < ps->layers[l]) maxp = ps->layers[l];
    }
    pthread_once(&g_huff_once, huff_init_once);
    e = calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->p = *param; e->preset = ps; e->block = param->num_samples_per_block; e->max_params = maxp;
    total = 64 + sizeof(double) * ((size_t)(maxp + 2) * 3 + (maxp + 1) * 3 + 64)
          + sizeof(double) * (size_t)e->block * (3 + 2 * ORACLE_MAX_LAYERS) + sizeof(double) * 3 * ORACLE_MAX_LAYERS * (maxp + 2) + 1024
          + sizeof(int32_t) * (size_t)e->block * 2 * param->num_channels + 64 * 64
          + sizeof(double) * (RICE_LOG2_MAX_PARTS + 1) * RICE_MAX_PARTS
          + (sizeof(double) * (maxp + 1) + sizeof(double *)) * (maxp + 1) + 64;
    /* the reference allocates its work area with malloc (linne_encoder.c:280); a fresh large malloc is
     * zero pages, which is what calloc gives here deterministically (matters for Q1/Q2 on the first block) */
    e->arena = calloc(1, total);
    if (!e->arena) { free(e); return NULL; }
    w = (uint8_t *)e->arena;
    e->lpc.max_order = maxp; e->lpc.max_samples = e->block;
    e->lpc.a = arena_take(&w, sizeof(double) * (maxp + 2));
    e->lpc.u = arena_take(&w, sizeof(double) * (maxp + 2));
    e->lpc.v = arena_take(&w, sizeof(double) * (maxp + 2));
    e->lpc.auto_corr = arena_take(&w, sizeof(double) * (maxp + 1));
    e->lpc.lpc_coef = arena_take(&w, sizeof(double) * (maxp + 1));
    e->lpc.parcor = arena_take(&w, sizeof(double) * (maxp + 1));
    e->lpc.buffer = arena_take(&w, sizeof(double) * e->block);
    e->lpc.r_mat = arena_take(&w, sizeof(double *) * (maxp + 1));
    for (l = 0; l < maxp + 1; l++) e->lpc.r_mat[l] = arena_take(&w, sizeof(double) * (maxp + 1));
    for (l = 0; l < ps->num_layers; l++) {
        e->layer[l].din = arena_take(&w, sizeof(double) * e->block);
        e->layer[l].params = arena_take(&w, sizeof(double) * ps->layers[l]);
        e->layer[l].dout = arena_take(&w, sizeof(double) * e->block);
        e->layer[l].dparams = arena_take(&w, sizeof(double) * ps->layers[l]);
        e->layer[l].momentum = arena_take(&w, sizeof(double) * ps->layers[l]);
        e->layer[l].num_params = ps->layers[l];
        e->layer[l].num_units = 1;
    }
    e->data_buffer = arena_take(&w, sizeof(double) * e->block);
    e->buffer_double = arena_take(&w, sizeof(double) * e->block);
    for (ch = 0; ch < param->num_channels; ch++) {
        e->buffer_int[ch] = arena_take(&w, sizeof(int32_t) * e->block);
        e->residual[ch] = arena_take(&w, sizeof(int32_t) * e->block);
    }
    e->part_mean = arena_take(&w, sizeof(double) * (RICE_LOG2_MAX_PARTS + 1) * RICE_MAX_PARTS);
    return e;
}
void oracle_encoder_destroy(struct OracleEncoder *e) { if (e) { free(e->arena); free(e); } }
void oracle_encoder_set_af_iterations(struct OracleEncoder *e, uint32_t n) { if (e) e->af_iters = n; }
void oracle_encoder_set_learning(struct OracleEncoder *e, uint32_t on) { if (e) e->learning = on; }

static void network_train(struct OracleEncoder *e, const double *input, uint32_t n, uint32_t max_iter, double lr, double eps)
{
    uint32_t itr, i, s_;
    int32_t l;
    const int32_t nl = (int32_t)e->preset->num_layers;
    const double alpha = 0.8f;                               /* trainer->momentum_alpha = 0.8f */
    double loss, prev_loss = FLT_MAX;
    for (l = 0; l < nl; l++) for (i = 0; i < e->layer[l].num_params; i++) e->layer[l].momentum[i] = 0.0f;
    for (itr = 0; itr < max_iter; itr++) {
        memcpy(e->data_buffer, input, sizeof(double) * n);
        for (l = 0; l < nl; l++) layer_forward(&e->layer[l], e->data_buffer, n);          /* LINNENetwork_CalculateLoss */
        loss = l1_loss(e->data_buffer, n);
        for (s_ = 0; s_ < n; s_++) { const double d = e->data_buffer[s_]; e->data_buffer[s_] = (double)((d > 0) - (d < 0)) / n; }
        for (l = nl - 1; l >= 0; l--) layer_backward(&e->layer[l], e->data_buffer, n);
        for (l = 0; l < nl; l++) {
            struct Layer *L = &e->layer[l];
            for (i = 0; i < L->num_params; i++) {
                L->momentum[i] = alpha * L->momentum[i] + lr * L->dparams[i];
                L->params[i] -= L->momentum[i];
            }
        }
        if (fabs(loss - prev_loss) < eps) break;
        prev_loss = loss;
    }
}

/* linne_network.c:582-602 */
static double network_search_set(struct OracleEncoder *e, const double *input, uint32_t n, uint32_t af_iters, double reg)
{
    uint32_t l;
    const uint32_t max_units = 1u << ((1u << LOG2_UNITS_BITWIDTH) - 1);
    memcpy(e->data_buffer, input, sizeof(double) * n);
    for (l = 0; l < e->preset->num_layers; l++) {
        struct Layer *L = &e->layer[l];
        L->num_units = layer_search_units(L, &e->lpc, e->data_buffer, n, (max_units < L->num_params) ? max_units : L->num_params, reg);
        layer_set_parameter(L, &e->lpc, e->data_buffer, n, af_iters, reg);
        layer_forward(L, e->data_buffer, n);
    }
    return l1_loss(e->data_buffer, n);
}
/* linne_network.c:605-630 */
static void network_set_units_and_parameters(struct OracleEncoder *e, const double *input, uint32_t n, struct OracleChannelTap *tap)
{
    uint32_t i, best_i = 0;
    double min_loss = FLT_MAX;
    for (i = 0; i < e->preset->num_regs; i++) {
        const double loss = network_search_set(e, input, n, 0, e->preset->regs[i]);
        if (tap) tap->pass_loss[i] = loss;
        if (loss < min_loss) { min_loss = loss; best_i = i; }
    }
    if (tap) tap->best_pass = best_i;
    (void)network_search_set(e, input, n, e->af_iters, e->preset->regs[best_i]);      /* the user's -a N applies to this pass only */
}

/* linne_encoder.c:480-529 */
static uint32_t decide_block_type(struct OracleEncoder *e, const int32_t *const *input, uint32_t n, struct OracleFrameTap *tap)
{
    uint32_t ch, s;
    double mean_length = 0.0;
    const uint32_t bits = e->p.bits_per_sample;
    for (ch = 0; ch < e->p.num_channels; ch++) {
        double len;
        for (s = 0; s < n; s++) e->buffer_double[s] = input[ch][s] * pow(2.0, -(int32_t)(bits - 1));
        len = lpc_estimate_code_length(&e->lpc, e->buffer_double, n, bits, e->layer[0].num_params);
        if (tap) {
            tap->ch[ch].est_r0 = e->lpc.auto_corr[0];
            memcpy(tap->ch[ch].est_parcor, e->lpc.parcor, sizeof(double) * (e->max_params + 1));
            tap->ch[ch].est_length = len;
        }
        mean_length += len;
    }
    mean_length /= e->p.num_channels;
    mean_length /= bits;
    if (mean_length >= RAW_THRESHOLD) return ORACLE_BLOCK_RAW;
    for (ch = 0; ch < e->p.num_channels; ch++)
        for (s = 0; s < n; s++) if (input[ch][s] != 0) return ORACLE_BLOCK_COMPRESS;
    return ORACLE_BLOCK_SILENT;
}

/* linne_encoder.c:613-696: MS, pre-emphasis, analysis, quantisation, FIR cascade */
static void compress_hotpath(struct OracleEncoder *e, const int32_t *const *input, uint32_t n,
        struct OracleFrameTap *tap, int32_t pre_prev[][ORACLE_NUM_PREEM], int32_t pre_coef[][ORACLE_NUM_PREEM],
        uint32_t units[][ORACLE_MAX_LAYERS], uint32_t rshifts[][ORACLE_MAX_LAYERS], int32_t (*icoef)[ORACLE_MAX_LAYERS][ORACLE_MAX_PARAMS])
{
    uint32_t ch, l, s, na;
    const uint32_t nch = e->p.num_channels, bits = e->p.bits_per_sample, nl = e->preset->num_layers;
    for (ch = 0; ch < nch; ch++) {
        memcpy(e->buffer_int[ch], input[ch], sizeof(int32_t) * n);
        if (n < e->block) memset(&e->buffer_int[ch][n], 0, sizeof(int32_t) * (e->block - n));
    }
    if (e->p.ch_process_method == 1) ms_conversion(e->buffer_int[0], e->buffer_int[1], n);
    for (ch = 0; ch < nch; ch++) {
        for (l = 0; l < ORACLE_NUM_PREEM; l++) {
            pre_prev[ch][l] = e->buffer_int[ch][0];
            pre_coef[ch][l] = preem_coefficient(e->buffer_int[ch], n);
            preemphasis(pre_prev[ch][l], pre_coef[ch][l], e->buffer_int[ch], n);
        }
    }
    /* linne_encoder.c:644-655 */
    na = ((n + 7u) / 8u) * 8u;
    if (na < e->max_params) na = e->max_params;
    if (na > e->block) na = e->block;
    if (tap) tap->num_analyze_samples = na;
    for (ch = 0; ch < nch; ch++) {
        struct OracleChannelTap *ct = tap ? &tap->ch[ch] : NULL;
        for (s = 0; s < na; s++) e->buffer_double[s] = e->buffer_int[ch][s] * pow(2.0, -(int32_t)(bits - 1));
        network_set_units_and_parameters(e, e->buffer_double, na, ct);
        if (e->learning) network_train(e, e->buffer_double, na, 2000, 0.1f, 1.0e-7);      /* linne_encoder.c:669-675, linne_internal.h:29-33 */
        for (l = 0; l < nl; l++) {
            units[ch][l] = e->layer[l].num_units;
            lpc_quantize(e->layer[l].params, e->layer[l].num_params, COEF_BITWIDTH, icoef[ch][l], &rshifts[ch][l]);
            if (ct) memcpy(ct->coef_double[l], e->layer[l].params, sizeof(double) * e->layer[l].num_params);
        }
        if (ct) ct->parcor_tail = e->lpc.parcor[e->layer[0].num_params];
    }
    for (ch = 0; ch < nch; ch++) {
        for (l = 0; l < nl; l++) {
            lpc_predict(e->buffer_int[ch], n, icoef[ch][l], e->layer[l].num_params, e->residual[ch], rshifts[ch][l], units[ch][l]);
            memcpy(e->buffer_int[ch], e->residual[ch], sizeof(int32_t) * n);
        }
    }
    if (tap) {
        for (ch = 0; ch < nch; ch++) {
            for (l = 0; l < ORACLE_NUM_PREEM; l++) { tap->ch[ch].preem_prev[l] = pre_prev[ch][l]; tap->ch[ch].preem_coef[l] = pre_coef[ch][l]; }
            for (l = 0; l < nl; l++) {
                tap->ch[ch].num_units[l] = units[ch][l]; tap->ch[ch].rshift[l] = rshifts[ch][l];
                memcpy(tap->ch[ch].coef[l], icoef[ch][l], sizeof(int32_t) * e->layer[l].num_params);
            }
        }
    }
}

/* linne_encoder.c:594-752 */
static int encode_compress(struct OracleEncoder *e, const int32_t *const *input, uint32_t n,
        uint8_t *data, uint32_t data_size, uint32_t *out_size, struct OracleFrameTap *tap)
{
    uint32_t ch, l, i;
    const uint32_t nch = e->p.num_channels, bits = e->p.bits_per_sample, nl = e->preset->num_layers;
    int32_t pre_prev[ORACLE_MAX_CHANNELS][ORACLE_NUM_PREEM], pre_coef[ORACLE_MAX_CHANNELS][ORACLE_NUM_PREEM];
    uint32_t units[ORACLE_MAX_CHANNELS][ORACLE_MAX_LAYERS], rshifts[ORACLE_MAX_CHANNELS][ORACLE_MAX_LAYERS];
    static __thread int32_t icoef[ORACLE_MAX_CHANNELS][ORACLE_MAX_LAYERS][ORACLE_MAX_PARAMS];
    struct BitW w;
    compress_hotpath(e, input, n, tap, pre_prev, pre_coef, units, rshifts, icoef);
    bw_open(&w, data, data_size);
    for (ch = 0; ch < nch; ch++)
        for (l = 0; l < ORACLE_NUM_PREEM; l++) {
            bw_put(&w, zigzag(pre_prev[ch][l]), bits + 1);
            bw_put(&w, (uint32_t)pre_coef[ch][l], PREEM_SHIFT - 1);
        }
    for (ch = 0; ch < nch; ch++)
        for (l = 0; l < nl; l++) {
            bw_put(&w, log2ceil(units[ch][l]), LOG2_UNITS_BITWIDTH);
            bw_put(&w, rshifts[ch][l], RSHIFT_BITWIDTH);
            for (i = 0; i < e->layer[l].num_params; i++) {
                const uint32_t sym = zigzag(icoef[ch][l][i]) & 255u;
                bw_put(&w, g_huff.code[sym], g_huff.len[sym]);
            }
        }
    for (ch = 0; ch < nch; ch++) rice_encode_core(&w, e->residual[ch], n, e->part_mean);
    bw_flush(&w);
    if (w.overflow) return ORACLE_INSUFFICIENT_BUFFER;
    *out_size = bw_tell(&w);
    return ORACLE_OK;
}

/* linne_encoder.c:532-591 */
static int encode_raw(struct OracleEncoder *e, const int32_t *const *input, uint32_t n, uint8_t *data, uint32_t data_size, uint32_t *out_size)
{
    uint32_t ch, s;
    const uint32_t bits = e->p.bits_per_sample, nch = e->p.num_channels;
    uint8_t *p = data;
    if (data_size < (bits * n * nch) / 8) return ORACLE_INSUFFICIENT_BUFFER;
    for (s = 0; s < n; s++)
        for (ch = 0; ch < nch; ch++) {
            const uint32_t u = zigzag(input[ch][s]);
            if (bits == 8) { *p++ = (uint8_t)u; }
            else if (bits == 16) { *p++ = (uint8_t)(u >> 8); *p++ = (uint8_t)u; }
            else if (bits == 24) { *p++ = (uint8_t)(u >> 16); *p++ = (uint8_t)(u >> 8); *p++ = (uint8_t)u; }
            else return ORACLE_INVALID_FORMAT;
        }
    *out_size = (uint32_t)(p - data);
    return ORACLE_OK;
}

int oracle_encode_block(struct OracleEncoder *e, const int32_t *const *input, uint32_t n,
        uint8_t *data, uint32_t data_size, uint32_t *output_size, struct OracleFrameTap *tap, int32_t *residual_out)
{
    uint32_t type, body = 0, ch;
    int ret;
    if (!e || !input || n == 0 || !data || data_size == 0 || !output_size) return ORACLE_INVALID_ARGUMENT;
    if (n > e->block) return ORACLE_INSUFFICIENT_BUFFER;
    if (data_size < 11) return ORACLE_INSUFFICIENT_BUFFER;
    if (tap) memset(tap, 0, sizeof(*tap));
    type = decide_block_type(e, input, n, tap);
    if (tap) { tap->block_type = type; tap->num_samples = n; }
    data[0] = 0xFF; data[1] = 0xFF;                        /* sync (linne_encoder.c:809) */
    data[8] = (uint8_t)type;
    data[9] = (uint8_t)(n >> 8); data[10] = (uint8_t)n;
    if (type == ORACLE_BLOCK_RAW) ret = encode_raw(e, input, n, data + 11, data_size - 11, &body);
    else if (type == ORACLE_BLOCK_COMPRESS) {
        ret = encode_compress(e, input, n, data + 11, data_size - 11, &body, tap);
        if (ret == ORACLE_OK && residual_out)
            for (ch = 0; ch < e->p.num_channels; ch++) memcpy(residual_out + (size_t)ch * e->block, e->residual[ch], sizeof(int32_t) * n);
    } else { body = 0; ret = ORACLE_OK; }
    if (ret != ORACLE_OK) return ret;
    {   /* linne_encoder.c:846-855 */
        const uint32_t bsize = body + 5;
        uint16_t crc;
        data[2] = (uint8_t)(bsize >> 24); data[3] = (uint8_t)(bsize >> 16); data[4] = (uint8_t)(bsize >> 8); data[5] = (uint8_t)bsize;
        crc = oracle_crc16(&data[8], body + 3);
        data[6] = (uint8_t)(crc >> 8); data[7] = (uint8_t)crc;
    }
    *output_size = 11 + body;
    return ORACLE_OK;
}

/* header: linne_encoder.c:53-138 */
static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static void put_be16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static uint32_t get_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t get_be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

int oracle_encode_whole(struct OracleEncoder *e, const int32_t *const *input, uint32_t num_samples,
        uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    uint32_t progress = 0, off = HEADER_SIZE, ch, wsize;
    const int32_t *ptr[ORACLE_MAX_CHANNELS];
    int ret;
    if (!e || !input || !data || !output_size) return ORACLE_INVALID_ARGUMENT;
    if (data_size < HEADER_SIZE) return ORACLE_INSUFFICIENT_BUFFER;
    if (num_samples == 0) return ORACLE_INVALID_FORMAT;
    memcpy(data, "IBRA", 4);
    put_be32(data + 4, FORMAT_VERSION); put_be32(data + 8, CODEC_VERSION);
    put_be16(data + 12, e->p.num_channels); put_be32(data + 14, num_samples); put_be32(data + 18, e->p.sampling_rate);
    put_be16(data + 22, e->p.bits_per_sample); put_be32(data + 24, e->p.num_samples_per_block);
    data[28] = (uint8_t)e->p.preset; data[29] = (uint8_t)e->p.ch_process_method;
    while (progress < num_samples) {
        const uint32_t n = (e->block < num_samples - progress) ? e->block : (num_samples - progress);
        for (ch = 0; ch < e->p.num_channels; ch++) ptr[ch] = &input[ch][progress];
        if ((ret = oracle_encode_block(e, ptr, n, data + off, data_size - off, &wsize, NULL, NULL)) != ORACLE_OK) return ret;
        off += wsize; progress += n;
    }
    *output_size = off;
    return ORACLE_OK;
}

int oracle_encode_frame_hotpath(struct OracleEncoder *e, const int32_t *input, uint32_t stride, uint32_t n,
        struct OracleFrameTap *tap, int32_t *residual)
{
    const int32_t *ptr[ORACLE_MAX_CHANNELS];
    uint32_t ch;
    int32_t pre_prev[ORACLE_MAX_CHANNELS][ORACLE_NUM_PREEM], pre_coef[ORACLE_MAX_CHANNELS][ORACLE_NUM_PREEM];
    uint32_t units[ORACLE_MAX_CHANNELS][ORACLE_MAX_LAYERS], rshifts[ORACLE_MAX_CHANNELS][ORACLE_MAX_LAYERS];
    static __thread int32_t icoef[ORACLE_MAX_CHANNELS][ORACLE_MAX_LAYERS][ORACLE_MAX_PARAMS];
    if (!e || !input || !tap || n == 0 || n > e->block) return ORACLE_INVALID_ARGUMENT;
    for (ch = 0; ch < e->p.num_channels; ch++) ptr[ch] = input + (size_t)ch * stride;
    memset(tap, 0, sizeof(*tap));
    tap->block_type = decide_block_type(e, ptr, n, tap);    /* runs first, as in EncodeBlock (Q1/Q2 state) */
    tap->num_samples = n;
    compress_hotpath(e, ptr, n, tap, pre_prev, pre_coef, units, rshifts, icoef);
    if (residual)
        for (ch = 0; ch < e->p.num_channels; ch++) memcpy(residual + (size_t)ch * stride, e->residual[ch], sizeof(int32_t) * n);
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * decoder: libs/linne_decoder/src/linne_decoder.c
 * ------------------------------------------------------------------------------------------- */
int oracle_decode_frame_hotpath(const struct OracleEncodeParameter *p, const struct OracleChannelTap *taps,
        int32_t *data, uint32_t stride, uint32_t n)
{
    uint32_t ch; int32_t l;
    const struct Preset *ps;
    if (!p || !taps || !data || p->preset >= NUM_PRESETS) return ORACLE_INVALID_ARGUMENT;
    ps = &k_presets[p->preset];
    for (ch = 0; ch < p->num_channels; ch++) {               /* linne_decoder.c:503-513 */
        int32_t *d = data + (size_t)ch * stride;
        for (l = (int32_t)ps->num_layers - 1; l >= 0; l--)
            lpc_synthesize(d, n, taps[ch].coef[l], ps->layers[l], taps[ch].rshift[l], taps[ch].num_units[l]);
        deemphasis2(taps[ch].preem_prev, taps[ch].preem_coef, d, n);
    }
    if (p->ch_process_method == 1) lr_conversion(data, data + stride, n);   /* :516-522 */
    return ORACLE_OK;
}

/* linne_decoder.c:430-526 */
static int decode_compress(const struct OracleEncodeParameter *p, const uint8_t *data, uint32_t size,
        int32_t *buffer, uint32_t stride, uint32_t n, uint32_t *consumed)
{
    static __thread struct OracleChannelTap taps[ORACLE_MAX_CHANNELS];
    const struct Preset *ps = &k_presets[p->preset];
    uint32_t ch, l, i;
    struct BitR r;
    br_open(&r, data, size);
    for (ch = 0; ch < p->num_channels; ch++)
        for (l = 0; l < ORACLE_NUM_PREEM; l++) {
            taps[ch].preem_prev[l] = unzigzag(br_get(&r, p->bits_per_sample + 1));
            taps[ch].preem_coef[l] = (int32_t)br_get(&r, PREEM_SHIFT - 1);
        }
    for (ch = 0; ch < p->num_channels; ch++)
        for (l = 0; l < ps->num_layers; l++) {
            taps[ch].num_units[l] = 1u << br_get(&r, LOG2_UNITS_BITWIDTH);
            taps[ch].rshift[l] = br_get(&r, RSHIFT_BITWIDTH);
            for (i = 0; i < ps->layers[l]; i++) taps[ch].coef[l][i] = unzigzag(huff_get(&g_huff, &r));
        }
    for (ch = 0; ch < p->num_channels; ch++) rice_decode_core(&r, buffer + (size_t)ch * stride, n);
    *consumed = br_tell_bytes(&r);
    return oracle_decode_frame_hotpath(p, taps, buffer, stride, n);
}

int oracle_decode_whole(const uint8_t *data, uint32_t data_size, int32_t *buffer, uint32_t buffer_channels,
        uint32_t stride, uint32_t *hdr, int check_crc)
{
    struct OracleEncodeParameter p;
    uint32_t total, progress = 0, off = HEADER_SIZE, ch, s;
    if (!data || !buffer) return ORACLE_INVALID_ARGUMENT;
    if (data_size < HEADER_SIZE) return ORACLE_INSUFFICIENT_DATA;
    if (memcmp(data, "IBRA", 4) != 0) return ORACLE_INVALID_FORMAT;
    pthread_once(&g_huff_once, huff_init_once);
    if (get_be32(data + 4) != FORMAT_VERSION || get_be32(data + 8) != CODEC_VERSION) return ORACLE_INVALID_FORMAT;
    p.num_channels = get_be16(data + 12); total = get_be32(data + 14); p.sampling_rate = get_be32(data + 18);
    p.bits_per_sample = get_be16(data + 22); p.num_samples_per_block = get_be32(data + 24);
    p.preset = data[28]; p.ch_process_method = data[29];
    if (hdr) { hdr[0] = FORMAT_VERSION; hdr[1] = CODEC_VERSION; hdr[2] = p.num_channels; hdr[3] = total; hdr[4] = p.sampling_rate;
               hdr[5] = p.bits_per_sample; hdr[6] = p.num_samples_per_block; hdr[7] = p.preset; hdr[8] = p.ch_process_method; }
    if (p.num_channels == 0 || total == 0 || p.sampling_rate == 0 || p.bits_per_sample == 0 || p.num_samples_per_block == 0
            || p.preset >= NUM_PRESETS || p.ch_process_method > 1 || (p.ch_process_method == 1 && p.num_channels == 1)) return ORACLE_INVALID_FORMAT;
    if (p.num_channels > ORACLE_MAX_CHANNELS || buffer_channels < p.num_channels || stride < total) return ORACLE_INSUFFICIENT_BUFFER;
    while (progress < total && off < data_size) {            /* linne_decoder.c:708-726 */
        const uint8_t *b = data + off;
        const uint32_t avail = data_size - off;
        uint32_t bsize, type, n, consumed = 0;
        int ret = ORACLE_OK;
        if (get_be16(b) != BLOCK_SYNC) return ORACLE_INVALID_FORMAT;
        bsize = get_be32(b + 2);
        if (bsize + 6 > avail) return ORACLE_INSUFFICIENT_DATA;
        if (check_crc && oracle_crc16(b + 8, bsize - 2) != get_be16(b + 6)) return ORACLE_DETECT_DATA_CORRUPTION;
        type = b[8]; n = get_be16(b + 9);
        if (n > stride - progress) return ORACLE_INSUFFICIENT_BUFFER;
        if (type == ORACLE_BLOCK_RAW) {                      /* :357-427 */
            const uint8_t *q = b + 11;
            const uint32_t bits = p.bits_per_sample;
            if (avail - 11 < (bits * n * p.num_channels) / 8) return ORACLE_INSUFFICIENT_DATA;
            for (s = 0; s < n; s++)
                for (ch = 0; ch < p.num_channels; ch++) {
                    uint32_t u = 0;
                    if (bits == 8) { u = q[0]; q += 1; } else if (bits == 16) { u = get_be16(q); q += 2; }
                    else if (bits == 24) { u = ((uint32_t)q[0] << 16) | ((uint32_t)q[1] << 8) | q[2]; q += 3; } else return ORACLE_INVALID_FORMAT;
                    buffer[(size_t)ch * stride + progress + s] = unzigzag(u);
                }
            consumed = (uint32_t)(q - (b + 11));
        } else if (type == ORACLE_BLOCK_COMPRESS) {
            ret = decode_compress(&p, b + 11, avail - 11, buffer + progress, stride, n, &consumed);
        } else if (type == ORACLE_BLOCK_SILENT) {
            for (ch = 0; ch < p.num_channels; ch++) memset(buffer + (size_t)ch * stride + progress, 0, sizeof(int32_t) * n);
        } else return ORACLE_INVALID_FORMAT;
        if (ret != ORACLE_OK) return ret;
        off += 11 + consumed; progress += n;
    }
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * cpu_baseline helper ("port"): one handle per thread over disjoint frames
 * ------------------------------------------------------------------------------------------- */
struct BenchJob { const struct OracleEncodeParameter *p; const int32_t *frames; uint32_t first, count; uint64_t bytes; };
static void *bench_worker(void *arg)
{
    struct BenchJob *j = arg;
    struct OracleEncoder *e = oracle_encoder_create(j->p);
    const uint32_t block = j->p->num_samples_per_block, nch = j->p->num_channels;
    const uint32_t cap = block * nch * 4 + 4096;
    uint8_t *out = malloc(cap);
    uint32_t f, ch, sz;
    const int32_t *ptr[ORACLE_MAX_CHANNELS];
    j->bytes = 0;
    for (f = j->first; f < j->first + j->count; f++) {
        for (ch = 0; ch < nch; ch++) ptr[ch] = j->frames + ((size_t)f * nch + ch) * block;
        if (oracle_encode_block(e, ptr, block, out, cap, &sz, NULL, NULL) == ORACLE_OK) j->bytes += sz;
    }
    free(out);
    oracle_encoder_destroy(e);
    return NULL;
}
double oracle_bench_encode(const struct OracleEncodeParameter *param, const int32_t *frames, uint32_t num_frames,
        uint32_t num_threads, uint64_t *total_bytes)
{
    pthread_t th[256];
    struct BenchJob jobs[256];
    struct timespec t0, t1;
    uint32_t t, first = 0;
    if (num_threads == 0) num_threads = 1;
    if (num_threads > 256) num_threads = 256;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (t = 0; t < num_threads; t++) {
        const uint32_t cnt = num_frames / num_threads + ((t < num_frames % num_threads) ? 1u : 0u);
        jobs[t].p = param; jobs[t].frames = frames; jobs[t].first = first; jobs[t].count = cnt; first += cnt;
        pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    }
    if (total_bytes) *total_bytes = 0;
    for (t = 0; t < num_threads; t++) { pthread_join(th[t], NULL); if (total_bytes) *total_bytes += jobs[t].bytes; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Sanitizer fuzz of the host entropy stage (CPU only; GPU sanitizers are not available on the pool):
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Ilinne_amd/csrc -Iinclude \
 *       tools/entropy_fuzz.c -lm -lpthread -o /tmp/entropy_fuzz && /tmp/entropy_fuzz [seconds] [seed]
 * 1. random COMPRESS / RAW / SILENT blocks of random shapes are packed and parsed back: the round trip must be exact;
 * 2. each valid block is then damaged (byte flips, truncation, length-field edits) and parsed with the CRC check off --
 *    any return code is fine, an out-of-bounds access or undefined shift is not (the sanitizers abort).
 * The output buffers are malloc'ed at their exact sizes so that AddressSanitizer sees a one-element overrun. */
#include "../linne_amd/csrc/lnn_entropy.c"
#include <stdio.h>
#include <time.h>

/* the layer shapes of the presets the library knows (linne_amd/csrc/lnn_api.c owns the real table) */
int lnn_preset_info(uint32_t p, uint32_t *nl, uint32_t *size, uint32_t *nr, double *regs)
{
    static const uint32_t tab[4][3] = { { 2, 8, 0 }, { 4, 32, 8 }, { 4, 128, 16 }, { 2, 4, 2 } };
    const uint32_t *t = tab[p & 3u];
    uint32_t l;
    *nl = t[2] ? 3u : 2u;
    for (l = 0; l < *nl; l++) size[l] = t[l];
    *nr = 1; regs[0] = 0.0;
    return 0;
}

static uint64_t g_s = 88172645463325252ull;
static uint32_t rnd(void) { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (uint32_t)(g_s >> 11); }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int32_t draw(uint32_t kind, uint32_t scale, uint32_t s)
{
    switch (kind) {
    case 0: return 0;
    case 1: { const double u = (rnd() + 1.0) / 4294967297.0; const int32_t v = (int32_t)(-(double)scale * log(u)); return (rnd() & 1u) ? v : -v; }
    case 2: return (int32_t)rnd();                                  /* full range, INT32_MIN included */
    case 3: return (rnd() % 97u == 0) ? (int32_t)(rnd() | 0x40000000u) : (int32_t)(rnd() % 3u) - 1;   /* spikes */
    case 4: return (s & 1u) ? INT32_MIN : INT32_MAX;
    default: return (int32_t)(rnd() % (2u * scale + 1u)) - (int32_t)scale;
    }
}

int main(int argc, char **argv)
{
    const double budget = argc > 1 ? atof(argv[1]) : 10.0;
    const double t0 = now();
    unsigned long long blocks = 0, parses = 0, accepted = 0;
    struct rice_scratch *sc = calloc(1, sizeof(*sc));
    if (argc > 2) g_s ^= strtoull(argv[2], NULL, 0) * 0x9E3779B97F4A7C15ull;
    lnn_tables_init();
    while (now() - t0 < budget) {
        static const uint32_t sizes[] = { 1, 2, 3, 7, 16, 100, 128, 129, 255, 256, 1000, 1024, 2048, 4095, 4096, 10240 };
        static const uint32_t bitses[] = { 8, 16, 24 };
        struct LINNEAmdShape sh;
        struct lnn_layers ly;
        uint32_t n, ch, l, i, s, type, size = 0, kind, scale, m;
        int32_t *pcm, *res, *prm, *out, *oprm;
        uint8_t *blk, *dmg;
        uint64_t cap;
        int ret;
        sh.num_channels = 1 + rnd() % 8; sh.bits_per_sample = bitses[rnd() % 3]; sh.preset = rnd() % 4; sh.ch_process_method = 0;
        n = sizes[rnd() % (sizeof(sizes) / sizeof(sizes[0]))];
        sh.num_samples_per_block = n + rnd() % 3;
        if (lnn_shape_layers(&sh, &ly) != 0) return 2;
        const uint32_t C = sh.num_channels, S = sh.num_samples_per_block;
        pcm = malloc(sizeof(int32_t) * C * S); res = malloc(sizeof(int32_t) * C * S);
        prm = calloc((size_t)C * LINNE_AMD_PARAM_WORDS, sizeof(int32_t));
        out = malloc(sizeof(int32_t) * C * S); oprm = malloc(sizeof(int32_t) * C * LINNE_AMD_PARAM_WORDS);
        kind = rnd() % 6; scale = 1u << (rnd() % 20);
        for (ch = 0; ch < C; ch++)
            for (s = 0; s < S; s++) {
                res[(size_t)ch * S + s] = draw(kind, scale, s);
                pcm[(size_t)ch * S + s] = (int32_t)(rnd() % (1u << sh.bits_per_sample)) - (int32_t)(1u << (sh.bits_per_sample - 1));
            }
        for (ch = 0; ch < C; ch++) {
            int32_t *rec = prm + (size_t)ch * LINNE_AMD_PARAM_WORDS;
            for (l = 0; l < 2; l++) {
                rec[LINNE_AMD_PRM_PREV + l] = (int32_t)(rnd() % (1u << sh.bits_per_sample)) - (int32_t)(1u << (sh.bits_per_sample - 1));
                rec[LINNE_AMD_PRM_PCOEF + l] = (int32_t)(rnd() % 16);
            }
            for (l = 0; l < ly.num_layers; l++) {
                rec[LINNE_AMD_PRM_UNITS + l] = (int32_t)(1u << (rnd() % 8));
                rec[LINNE_AMD_PRM_RSHIFT + l] = (int32_t)(rnd() % 16);
                for (i = 0; i < ly.size[l]; i++) rec[LINNE_AMD_PRM_COEF + ly.offset[l] + i] = (int32_t)(rnd() % 256) - 128;
            }
        }
        type = (rnd() % 8 == 0) ? LNN_BLOCK_RAW : (rnd() % 16 == 0) ? LNN_BLOCK_SILENT : LNN_BLOCK_COMPRESS;
        cap = 64 + (uint64_t)C * (64 + 2 * ly.total) + (uint64_t)C * n * 9;       /* generous: 33+ bits per sample worst case */
        blk = malloc(cap);
        { struct pcm_view pv; pv.frames = pcm; pv.planes = NULL; pv.first_sample = 0;
          ret = pack_block(&sh, &ly, type, n, &pv, 0, res, prm, NULL, NULL, blk, cap, &size, sc); }
        if (ret != LNN_OK) { fprintf(stderr, "pack_block failed: %d (type %u n %u C %u kind %u)\n", ret, type, n, C, kind); return 1; }
        blocks++;
        {   /* exact-size copy, round trip */
            uint32_t t2 = 99, n2 = 0, used = 0;
            dmg = malloc(size); memcpy(dmg, blk, size);
            ret = lnn_parse_block(&sh, &ly, dmg, size, 1, S, &t2, &n2, &used, out, oprm);
            if (ret != LNN_OK || t2 != type || n2 != n || used != size) { fprintf(stderr, "round trip: ret %d type %u/%u n %u/%u used %u/%u\n", ret, t2, type, n2, n, used, size); return 1; }
            for (ch = 0; ch < C; ch++) {
                const int32_t *want = (type == LNN_BLOCK_COMPRESS) ? res : pcm;
                for (s = 0; s < n; s++) {
                    const int32_t w = (type == LNN_BLOCK_SILENT) ? 0 : want[(size_t)ch * S + s];
                    if (out[(size_t)ch * S + s] != w) { fprintf(stderr, "round trip: sample mismatch ch %u s %u (type %u kind %u)\n", ch, s, type, kind); return 1; }
                }
                if (type == LNN_BLOCK_COMPRESS && memcmp(oprm + (size_t)ch * LINNE_AMD_PARAM_WORDS, prm + (size_t)ch * LINNE_AMD_PARAM_WORDS, sizeof(int32_t) * (LINNE_AMD_PRM_COEF + ly.total)) != 0) {
                    fprintf(stderr, "round trip: parameter mismatch ch %u\n", ch); return 1;
                }
            }
            free(dmg);
        }
        for (m = 0; m < 24; m++) {  /* damaged copies, each in a buffer of exactly the bytes the parser is told about */
            uint32_t t2 = 0, n2 = 0, used = 0, avail = size, flips = 1 + rnd() % 4;
            const uint32_t mode = rnd() % 4;
            if (mode == 1 && size > 1) avail = 1 + rnd() % (size - 1);                 /* truncated */
            dmg = malloc(avail ? avail : 1); memcpy(dmg, blk, avail);
            if (mode == 0 || mode == 3) while (flips--) dmg[rnd() % avail] ^= (uint8_t)(1u << (rnd() % 8));
            if (mode == 2 && avail >= 11) {                                          /* header edits: size, type, sample count */
                const uint32_t w = rnd() % 3;
                if (w == 0) put_be32(dmg + 2, rnd() % (2 * size + 2)); else if (w == 1) dmg[8] = (uint8_t)rnd(); else put_be16(dmg + 9, rnd() & 0xFFFFu);
            }
            if (mode == 3 && avail > 16) for (i = 11 + rnd() % (avail - 11); i < avail; i++) dmg[i] = (uint8_t)rnd();   /* random tail */
            ret = lnn_parse_block(&sh, &ly, dmg, avail, 0, S, &t2, &n2, &used, out, oprm);
            parses++; if (ret == LNN_OK) accepted++;
            free(dmg);
        }
        free(blk); free(pcm); free(res); free(prm); free(out); free(oprm);
    }
    free(sc->u); free(sc->t); free(sc);
    printf("entropy_fuzz: %llu blocks round-tripped, %llu damaged parses (%llu accepted), %.1f s, no sanitizer report\n", blocks, parses, accepted, now() - t0);
    return 0;
}

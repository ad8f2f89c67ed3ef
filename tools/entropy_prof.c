/* phase timing of the host entropy stage on synthetic Laplacian residuals (many distinct frames, so that branch history
 * does not learn one): gcc -O3 -Ilinne_amd/csrc -Iinclude tools/entropy_prof.c -lm -lpthread */
#define LNN_PROF 1
#include "../linne_amd/csrc/lnn_entropy.c"
#include <stdio.h>
#include <time.h>
int lnn_preset_info(uint32_t p, uint32_t *a, uint32_t *b, uint32_t *c, double *d) { (void)p; (void)a; (void)b; (void)c; (void)d; return -1; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void)
{
    enum { N = 10240, NF = 64, REP = 20 };
    static int32_t x[NF][N], y[N]; static uint8_t buf[NF][N * 4]; static uint64_t bytes[NF];
    struct rice_scratch *sc = calloc(1, sizeof(*sc));
    struct bitw w; struct bitr r; uint32_t i, f, rep; double t0, t1, t2, t3; int bad = 0; uint32_t crc = 0;
    lnn_tables_init();
    srand(1);
    for (f = 0; f < NF; f++)
        for (i = 0; i < N; i++) { double u = (rand() + 1.0) / (RAND_MAX + 2.0); double v = -120.0 * (1.0 + ((i + 97 * f) % N) / 2048) * log(u); x[f][i] = (rand() & 1) ? (int32_t)v : -(int32_t)v; }
    t0 = now();
    for (rep = 0; rep < REP; rep++) for (f = 0; f < NF; f++) { bw_open(&w, buf[f], sizeof(buf[f])); rice_encode(&w, x[f], N, sc); bw_flush(&w); bytes[f] = bw_bytes(&w); }
    t1 = now();
    for (rep = 0; rep < REP; rep++) for (f = 0; f < NF; f++) { br_open(&r, buf[f], bytes[f]); rice_decode(&r, y, N); if (rep == 0 && memcmp(x[f], y, sizeof(y))) bad = 1; }
    t2 = now();
    for (rep = 0; rep < REP; rep++) for (f = 0; f < NF; f++) crc += lnn_crc16(buf[f], bytes[f]);
    t3 = now();
    printf("per %d samples: rice encode %.1f us, rice decode %.1f us, crc16 %.1f us (%llu bytes) roundtrip %s [%u]\n", N,
            (t1 - t0) / (REP * NF) * 1e6, (t2 - t1) / (REP * NF) * 1e6, (t3 - t2) / (REP * NF) * 1e6, (unsigned long long)bytes[0], bad ? "BAD" : "ok", crc);
    printf("encode cycles per call: zigzag %.0f, means %.0f, search %.0f, emit %.0f; k2 %.0f, prefix %.0f\n", (double)g_prof[0] / (REP * NF), (double)g_prof[1] / (REP * NF), (double)g_prof[2] / (REP * NF), (double)g_prof[3] / (REP * NF), (double)g_prof[4] / (REP * NF), (double)g_prof[5] / (REP * NF));
    return 0;
}

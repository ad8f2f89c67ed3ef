/* phase timing of the host Rice coder on a synthetic Laplacian residual: gcc -O3 -Ilinne_amd/csrc -Iinclude tools/entropy_prof.c -lm -lpthread */
#include "../linne_amd/csrc/lnn_entropy.c"
#include <stdio.h>
int lnn_preset_info(uint32_t p, uint32_t *a, uint32_t *b, uint32_t *c, double *d) { (void)p; (void)a; (void)b; (void)c; (void)d; return -1; }
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void)
{
    enum { N = 10240, REP = 2000 };
    static int32_t x[N], y[N]; static uint8_t buf[N * 8];
    struct rice_scratch *sc = calloc(1, sizeof(*sc));
    struct bitw w; struct bitr r; uint32_t i, rep; double t0, t1, t2; uint64_t bytes = 0;
    lnn_tables_init();
    srand(1);
    for (i = 0; i < N; i++) { double u = (rand() + 1.0) / (RAND_MAX + 2.0); double v = -40.0 * (1.0 + (i / 2048)) * log(u); x[i] = (rand() & 1) ? (int32_t)v : -(int32_t)v; }
    t0 = now();
    for (rep = 0; rep < REP; rep++) { bw_open(&w, buf, sizeof(buf)); rice_encode(&w, x, N, sc); bw_flush(&w); bytes = bw_bytes(&w); }
    t1 = now();
    for (rep = 0; rep < REP; rep++) { br_open(&r, buf, bytes); rice_decode(&r, y, N); }
    t2 = now();
    printf("encode %.1f us, decode %.1f us per %d samples (%llu bytes) roundtrip %s\n", (t1 - t0) / REP * 1e6, (t2 - t1) / REP * 1e6, N, (unsigned long long)bytes, memcmp(x, y, sizeof(x)) ? "BAD" : "ok");
    return 0;
}

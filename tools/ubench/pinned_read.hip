// CPU read bandwidth of hipHostMalloc memory (default / non-coherent) vs malloc memory, single thread and 16 threads
// hipcc -O2 pinned_read.hip -o pinned_read -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <pthread.h>
#include <time.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
struct Job { const int32_t *p; size_t n; long long sum; };
static void *worker(void *a) { Job *j = (Job *)a; long long s = 0; for (size_t i = 0; i < j->n; i++) s += j->p[i]; j->sum = s; return nullptr; }
static void run(const char *name, int32_t *buf, size_t n)
{
    for (int threads = 1; threads <= 16; threads *= 16) {
        pthread_t th[16]; Job jobs[16];
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            for (int t = 0; t < threads; t++) { jobs[t].p = buf + (n / threads) * t; jobs[t].n = n / threads; pthread_create(&th[t], nullptr, worker, &jobs[t]); }
            for (int t = 0; t < threads; t++) pthread_join(th[t], nullptr);
            const double dt = now() - t0; if (dt < best) best = dt;
        }
        printf("%-28s %2d threads: %.1f GB/s\n", name, threads, n * 4.0 / best / 1e9);
    }
}
int main()
{
    const size_t n = 64u << 20;   // 256 MB
    int32_t *a = (int32_t *)malloc(n * 4); memset(a, 1, n * 4);
    int32_t *b = nullptr, *c = nullptr;
    hipHostMalloc((void **)&b, n * 4, hipHostMallocDefault); memset(b, 1, n * 4);
    hipHostMalloc((void **)&c, n * 4, hipHostMallocNonCoherent); memset(c, 1, n * 4);
    run("malloc", a, n); run("hipHostMalloc default", b, n); run("hipHostMalloc non-coherent", c, n);
    return 0;
}

// micro-benchmark: v_mfma_f64_16x16x4_f64 on gfx950 -- its rate, and whether it runs BESIDE FP64 vector work
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_f64.hip -o mfma_f64
//   MODE 0: vector only   (16 independent chains per lane of unfused multiply + add, as dp_rate.hip)
//   MODE 1: matrix only   (8 independent accumulator tiles per wave)
//   MODE 2: both in every wave, interleaved (16 multiply+add pairs and MF matrix instructions per trip)
//   MODE 3: half of the block's waves run the vector loop, the other half the matrix loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE, int MF>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters)
{
    double acc[16];
    d4 m[8];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-9 + i;
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = (d4){ 0.0, 0.0, 0.0, 0.0 };
    const bool vec = (MODE == 0) || (MODE == 2) || (MODE == 3 && (threadIdx.x >> 6) % 2 == 0);
    const bool mat = (MODE == 1) || (MODE == 2) || (MODE == 3 && (threadIdx.x >> 6) % 2 == 1);
    double av = a + threadIdx.x * 1e-12, bv = b;
    for (int it = 0; it < iters; it++) {
        if (vec) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = acc[i] + a * acc[(i + 1) & 15];
        }
        if (mat) {
#pragma unroll
            for (int i = 0; i < MF; i++) m[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, m[i & 7], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += m[i].x + m[i].y + m[i].z + m[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int MF> void run(const char *name, int wpb_blocks)
{
    const int blocks = 256 * wpb_blocks, iters = 2000;
    double *d; hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, MF>), dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, MF>), dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double vfrac = (MODE == 0 || MODE == 2) ? 1.0 : (MODE == 3 ? 0.5 : 0.0), mfrac = (MODE == 1 || MODE == 2) ? 1.0 : (MODE == 3 ? 0.5 : 0.0);
    const double vpairs = waves * vfrac * iters * 16 * 64, mflop = waves * mfrac * iters * MF * 2048.0;
    printf("%-28s blocks/CU %d: %8.3f ms | vector %.2f T pairs/s = %.1f TFLOP/s | matrix %.1f TFLOP/s (%.1f clk/SIMD per MFMA at 2.4 GHz if alone)\n", name, wpb_blocks, ms,
           vpairs / ms / 1e9, 2 * vpairs / ms / 1e9, mflop / ms / 1e9, mfrac > 0 ? (ms * 1e-3 * 2.4e9) / (waves * mfrac * iters * MF / 1024.0) : 0.0);
    hipFree(d);
}
int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0>("vector only", w);
        run<1, 8>("matrix only (8 per trip)", w);
        run<2, 2>("both, 2 MFMA per 16 pairs", w);
        run<2, 4>("both, 4 MFMA per 16 pairs", w);
        run<2, 8>("both, 8 MFMA per 16 pairs", w);
        run<3, 4>("half vector / half matrix", w);
    }
    return 0;
}

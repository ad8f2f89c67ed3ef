// micro-benchmark: issue rate of unfused FP64 multiply + add (and FMA) on gfx950, all CUs busy
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dp_rate.hip -o dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters)
{
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) acc[i] = acc[i] + a * acc[(i + 1) & 15];          // mul + add, unfused
            else if (MODE == 1) acc[i] = __builtin_fma(a, acc[(i + 1) & 15], acc[i]);
            else if (MODE == 2) acc[i] = acc[i] + b;                          // add only
            else acc[i] = acc[i] * a;                                         // mul only
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, double flop_per_op, int wpb_blocks)
{
    const int blocks = 256 * wpb_blocks, iters = 4000;
    double *d; hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16;
    printf("%-14s blocks/CU %d: %.3f ms, %.2f T lane-ops/s (%s), %.2f TFLOP/s\n", name, wpb_blocks, ms, ops / ms / 1e9, MODE == 0 ? "mul+add pairs" : "instr", ops * flop_per_op / ms / 1e9);
    hipFree(d);
}
int main()
{
    for (int w = 1; w <= 4; w *= 2) { run<0>("mul+add", 2, w); run<1>("fma", 2, w); run<2>("add", 1, w); run<3>("mul", 1, w); }
    return 0;
}

// micro-benchmark: do fp32 matrix instructions (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32) run BESIDE unfused FP64 vector
// work on gfx950?  (FP64 MFMA does not: tools/ubench/mfma_f64.hip, profiles/r02_mfma_f64.txt.)  The question behind it: could the
// order-free trials of the long layer's unit-count search (127 of k_search_long's 383 FP64 instructions per sample) leave the FP64
// vector unit as fp32 Toeplitz products under a wider certificate?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_f32.hip -o mfma_f32
//   MODE 0: vector only   (16 independent chains per lane of unfused FP64 multiply + add, as dp_rate.hip)
//   MODE 1: matrix only   (8 independent accumulator tiles per wave)
//   MODE 2: both in every wave, interleaved (16 multiply+add pairs and MF matrix instructions per trip)
//   MODE 3: half of the block's waves run the vector loop, the other half the matrix loop
//   BIG = 0: 16x16x4 (512 MAC per instruction), BIG = 1: 32x32x2 (2048 MAC per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int MODE, int MF, int BIG>
__global__ __launch_bounds__(256) void k(double *out, double a, float b, int iters)
{
    double acc[16];
    f4 m[8];
    f16 M[4];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-9 + i;
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = (f4){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) M[i][j] = 0.f;
    const bool vec = (MODE == 0) || (MODE == 2) || (MODE == 3 && (threadIdx.x >> 6) % 2 == 0);
    const bool mat = (MODE == 1) || (MODE == 2) || (MODE == 3 && (threadIdx.x >> 6) % 2 == 1);
    float av = (float)a + threadIdx.x * 1e-6f, bv = b;
    for (int it = 0; it < iters; it++) {
        if (vec) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = acc[i] + a * acc[(i + 1) & 15];
        }
        if (mat) {
            if (BIG) {
#pragma unroll
                for (int i = 0; i < MF; i++) M[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, M[i & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < MF; i++) m[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, m[i & 7], 0, 0, 0);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += m[i].x + m[i].y + m[i].z + m[i].w;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) s += M[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int MF, int BIG> void run(const char *name, int wpb_blocks)
{
    const int blocks = 256 * wpb_blocks, iters = 2000;
    double *d; hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, MF, BIG>), dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, MF, BIG>), dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double vfrac = (MODE == 0 || MODE == 2) ? 1.0 : (MODE == 3 ? 0.5 : 0.0), mfrac = (MODE == 1 || MODE == 2) ? 1.0 : (MODE == 3 ? 0.5 : 0.0);
    const double vpairs = waves * vfrac * iters * 16 * 64, mflop = waves * mfrac * iters * MF * (BIG ? 4096.0 : 1024.0);
    printf("%-34s blocks/CU %d: %8.3f ms | FP64 vector %.2f T pairs/s = %.1f TFLOP/s | fp32 matrix %.1f TFLOP/s\n", name, wpb_blocks, ms,
           vpairs / ms / 1e9, 2 * vpairs / ms / 1e9, mflop / ms / 1e9);
    hipFree(d);
}
int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0, 0>("vector only", w);
        run<1, 8, 0>("16x16x4 only (8 per trip)", w);
        run<1, 4, 1>("32x32x2 only (4 per trip)", w);
        run<2, 2, 0>("both, 2 x 16x16x4 per 16 pairs", w);
        run<2, 4, 0>("both, 4 x 16x16x4 per 16 pairs", w);
        run<2, 8, 0>("both, 8 x 16x16x4 per 16 pairs", w);
        run<2, 2, 1>("both, 2 x 32x32x2 per 16 pairs", w);
        run<2, 4, 1>("both, 4 x 32x32x2 per 16 pairs", w);
        run<3, 8, 0>("half vector / half 16x16x4", w);
        run<3, 4, 1>("half vector / half 32x32x2", w);
    }
    return 0;
}

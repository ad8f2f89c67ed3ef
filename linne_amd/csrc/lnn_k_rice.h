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
struct RicePlanArgs {
    const int32_t *resid; const uint32_t *nsmp; uint8_t *plan; uint32_t C, S, nsteps;
    double steps[32];
};
__device__ __forceinline__ uint32_t rp_zz(int32_t v) { const uint32_t d = (uint32_t)v << 1; return (v < 0) ? ((0u - d) - 1u) : d; }
__device__ __forceinline__ uint32_t rp_gamma_len(uint32_t u) { return u ? (2u * (32u - (uint32_t)__clz((int)(u + 1u))) - 1u) : 1u; }   /* 2*ceil_log2(u+2)-1 */

__global__ __launch_bounds__(RICE_THREADS) void k_rice_plan(RicePlanArgs a)
{
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
    for (uint32_t p = tid; p < parts; p += RICE_THREADS) {
        const int32_t *q = x + (size_t)p * nsf;
        uint64_t sum = 0;
        for (uint32_t j = 0; j < nsf; j++) sum += rp_zz(q[j]);
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
    /* per entry: the samples' fixed part and the parameter's own code */
    for (uint32_t e = tid; e < nent; e += RICE_THREADS) {
        const uint32_t o = 31u - (uint32_t)__clz((int)(e + 1u)), p = e - ((1u << o) - 1u);
        const uint32_t k = kk[e];
        uint32_t bits = (n >> o) * (k + 2u);
        bits += p ? rp_gamma_len(rp_zz((int32_t)k - (int32_t)kk[e - 1])) : 5u;
        atomicAdd(&tot[o], bits);
    }
    /* per finest partition: the excess of its samples under the parameter of each order's enclosing partition */
    for (uint32_t p = tid; p < parts; p += RICE_THREADS) {
        const int32_t *q = x + (size_t)p * nsf;
        uint32_t kc[11], acc[11];
#pragma unroll
        for (uint32_t o = 0; o < 11; o++) { acc[o] = 0; kc[o] = (o <= max_order) ? kk[((1u << o) - 1u) + (p >> (max_order - o))] : 0u; }
        for (uint32_t j = 0; j < nsf; j++) {
            const uint32_t v = rp_zz(q[j]);
#pragma unroll
            for (uint32_t o = 0; o < 11; o++) { const uint32_t k1pow = 1u << ((kc[o] + 1u) & 31u); acc[o] += ((v > k1pow) ? (v - k1pow) : 0u) >> kc[o]; }
        }
#pragma unroll
        for (uint32_t o = 0; o < 11; o++) if (o <= max_order) atomicAdd(&tot[o], acc[o]);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0, min_bits = 0xFFFFFFFFu;
        for (uint32_t o = 0; o <= max_order; o++) if (min_bits > tot[o]) { min_bits = tot[o]; best = o; }
        best_s = best;
        rec[0] = (uint8_t)best; rec[1] = (uint8_t)flag;
    }
    __syncthreads();
    const uint32_t best = best_s;
    for (uint32_t p = tid; p < (1u << best); p += RICE_THREADS) rec[LINNE_AMD_RICE_PLAN_K2 + p] = kk[((1u << best) - 1u) + p];
}


#endif

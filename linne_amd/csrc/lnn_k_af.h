/* lnn_k_af.h -- the auxiliary-function refinement of the coefficients (`-a N`).
 * Part of the single translation unit lnn_device.hip (included there); not a stand-alone header.
 *
 * Reference: LPC_CalculateCoefAF, libs/lpc/src/lpc.c:578-633 -- an IRLS iteration for the L1 norm of the residual, started
 * from the Levinson-Durbin coefficients -- with LPCAF_CalculateCoefMatrixAndVector (:451-509, the forward-residual form that
 * is compiled) and LPC_CholeskyDecomposition (:402-448); reached only from the FINAL pass of LINNENetwork_SetUnitsAndParameters
 * (linne_network.c:605-630 -> :350-376) with the user's iteration count, per layer, for every unit of the chosen unit count.
 *
 * One problem = one unit of one channel-frame: order np = P / u, n = na / u samples of the layer's (unwindowed) input.  Per
 * iteration:
 *   k_af_resid   residual of every sample (chain over the taps, unfused), |.|, the clamp at 1e-6 and the IEEE reciprocal;
 *   k_af_obj     the objective: ONE ordered chain over the unit's samples, as the reference adds it (decides convergence);
 *   k_af_matrix  r_mat[i][j] (j >= i) and r_vec[i]: each entry its own ordered chain over the samples, products associated as
 *                the reference writes them ((x_i * x_j) * inv) -- one thread per entry, 8384 independent chains for np = 128;
 *   k_af_pivot / k_af_column   the Cholesky factorisation, one pivot per round: the device forms the pivot sums (chains in the
 *                reference's descending order), the HOST takes pow(sum, -0.5) with its libm -- glibc's pow is not correctly
 *                rounded, so no device routine can promise its bits -- and the device finishes the column;
 *   k_af_solve   forward / backward substitution (chains in the reference's order), the convergence test, the new coefficients.
 * Problems of a layer are listed compactly (k_af_init) so that a pivot round trip moves 8 bytes per live problem.
 */
#ifndef LNN_K_AF_H_INCLUDED
#define LNN_K_AF_H_INCLUDED

#define AF_EPS 1e-6                     /* LPCAF_RESIDUAL_EPSILON, lpc.c:20 */

/* input sample s of the layer (layer 0 reads the pre-emphasised int32 channel, linne_encoder.c:661-663) */
__device__ __forceinline__ double af_x(const Plan &p, uint32_t layer, uint32_t cur, uint32_t job, uint32_t s)
{
    return (layer == 0) ? ((double)p.xint[(size_t)(job / p.R) * p.S + s] * p.scale) : p.sig[((size_t)job * 2 + cur) * p.S + s];
}

/* winner of the R search passes of every channel-frame (linne_network.c:618-626): its index, loss and regulariser */
__global__ void k_af_best(Plan p, uint32_t *best, double *loss, double *reg)
{
    const uint32_t cf = blockIdx.x * blockDim.x + threadIdx.x;
    if (cf >= p.J / p.R) return;
    double min_loss = (double)FLT_MAX; uint32_t b = 0;
    for (uint32_t r = 0; r < p.R; r++) { const double l = p.jloss[(size_t)cf * p.R + r]; if (l < min_loss) { min_loss = l; b = r; } }
    best[cf] = b; loss[cf] = p.jloss[(size_t)cf * p.R + b]; reg[cf] = p.regs[b];
}

/* per job: the chosen units' coefficients in LPC order, the problems' states, and the compact problem list */
__global__ void k_af_init(Plan p, uint32_t layer)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= p.J) return;
    const DevClass &c = job_class(p, job);
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    const uint32_t t = 31u - (uint32_t)__clz((int)u);
    const double reg = p.job_reg ? p.job_reg[job] : p.regs[job % p.R];
    const double *h = p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP;
    double *a = p.af_a + (size_t)job * LNN_MAXP;
    const uint32_t ajob = (layer == 0) ? job - job % p.R : job;
    for (uint32_t un = 0; un < u; un++) {
        for (uint32_t i = 0; i < np; i++) a[un * np + i] = h[un * np + (np - 1 - i)];     /* undo linne_network.c:368-373 */
        const double r0 = p.acorr[((size_t)ajob * LNN_MAXT + t) * LNN_ACW + (size_t)un * (np + 1)];
        /* lpc.c:349-355 (n < order: zeros, lags left unscaled), :358 (ridge), :594-601 (small lag 0: zeros) */
        const bool zero = (n < np) ? (fabs(r0) < (double)FLT_EPSILON) : (fabs(r0 * (1.0 + reg)) < (double)FLT_EPSILON);
        p.af_state[(size_t)job * LNN_MAXU + un] = zero ? 2u : 0u;
        p.af_prev[(size_t)job * LNN_MAXU + un] = (double)FLT_MAX;
        if (!zero) p.af_prob[atomicAdd(p.af_nprob, 1u)] = job * LNN_MAXU + un;
    }
}

#define AFR_THREADS 256
/* residual, clamp, reciprocal (lpc.c:480-489); |residual| goes to the layer's OUTPUT buffer (free until the forward pass) */
__global__ __launch_bounds__(AFR_THREADS) void k_af_resid(Plan p, uint32_t layer, uint32_t cur)
{
    __shared__ double sa[LNN_MAXP];
    const uint32_t job = blockIdx.x, tid = threadIdx.x;
    const DevClass &c = job_class(p, job);
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    for (uint32_t i = tid; i < P; i += AFR_THREADS) sa[i] = p.af_a[(size_t)job * LNN_MAXP + i];
    __syncthreads();
    double *absr = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S;
    double *inv = p.af_inv + (size_t)job * p.S;
    for (uint32_t s = blockIdx.y * AFR_THREADS * 4u + tid; s < c.na && s < (blockIdx.y + 1u) * AFR_THREADS * 4u; s += AFR_THREADS) {
        const uint32_t un = s / n, loc = s - un * n;
        if (loc < np || p.af_state[(size_t)job * LNN_MAXU + un] != 0u) { absr[s] = 0.0; inv[s] = 0.0; continue; }
        const double *a = sa + un * np;
        double r = af_x(p, layer, cur, job, s);
        for (uint32_t i = 0; i < np; i++) r += a[i] * af_x(p, layer, cur, job, s - i - 1);
        r = fabs(r);
        absr[s] = r;
        inv[s] = 1.0 / ((r < AF_EPS) ? AF_EPS : r);
    }
}

/* the objective of every live problem: one ordered chain (lpc.c:486, :503).  A wave per problem: 256 magnitudes at a time go to
 * LDS with coalesced loads, and the chain adds them up from there (every lane runs the same chain). */
__global__ __launch_bounds__(64) void k_af_obj(Plan p, uint32_t layer, uint32_t cur)
{
    __shared__ __attribute__((aligned(16))) double mag[256];
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    if (k >= *p.af_nprob) return;
    const uint32_t job = p.af_prob[k] / LNN_MAXU, un = p.af_prob[k] % LNN_MAXU;
    if (p.af_state[(size_t)job * LNN_MAXU + un] != 0u) return;
    const DevClass &c = job_class(p, job);
    const uint32_t u = p.lunits[(size_t)job * LNN_MAXL + layer], np = p.P[layer] / u, n = c.na / u;
    const double *absr = p.sig + ((size_t)job * 2 + (cur ^ 1u)) * p.S + (size_t)un * n;
    double obj = 0.0;
    for (uint32_t s0 = np; s0 < n; s0 += 256u) {
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++) { const uint32_t s = s0 + r * 64u + lane; mag[r * 64u + lane] = (s < n) ? absr[s] : 0.0; }
        __builtin_amdgcn_wave_barrier();
        const uint32_t cnt = (n - s0 < 256u) ? (n - s0) : 256u;
        if (cnt == 256u) {
#pragma unroll 8
            for (uint32_t q = 0; q < 256u; q += 2u) { const lnn_d2 v = *(const lnn_d2 *)(mag + q); obj += v.x; obj += v.y; }
        } else for (uint32_t q = 0; q < cnt; q++) obj += mag[q];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) p.af_obj[(size_t)job * LNN_MAXU + un] = obj / (double)(n - np);
}

#define AFM_THREADS 256
#define AFM_T 1024                      /* chain steps per staged tile */
/* blocks per job: a block works on ONE unit (its samples are staged in LDS for all its threads), 256 entries of it */
__host__ __device__ inline uint32_t afm_entries(uint32_t np) { return np * (np + 1u) / 2u + np; }
__host__ __device__ inline uint32_t afm_blocks_per_unit(uint32_t np) { return (afm_entries(np) + AFM_THREADS - 1u) / AFM_THREADS; }
/* r_mat (upper triangle) and r_vec of every live problem: grid = (jobs, max over unit counts of units x blocks per unit); entry r
 * of a unit: the first np entries are r_vec, the others the pairs (i, j >= i) in row order.  The unit's samples and reciprocals
 * are staged in LDS AFM_T steps at a time (coalesced loads); a wave's lanes read x[s-i-1] at one or two addresses (same row) and
 * x[s-j-1] at consecutive ones. */
__global__ __launch_bounds__(AFM_THREADS) void k_af_matrix(Plan p, uint32_t layer, uint32_t cur)
{
    __shared__ double xs[AFM_T + LNN_MAXP];      /* xs[k] = x[base + s0 - np + k] */
    __shared__ double iv[AFM_T];                 /* iv[k] = inv[s0 + k] */
    const uint32_t job = blockIdx.x, tid = threadIdx.x;
    const DevClass &c = job_class(p, job);
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    const uint32_t M = afm_entries(np), bpu = afm_blocks_per_unit(np);
    const uint32_t un = blockIdx.y / bpu, r = (blockIdx.y - un * bpu) * AFM_THREADS + tid;
    if (un >= u || p.af_state[(size_t)job * LNN_MAXU + un] != 0u) return;
    const double *inv = p.af_inv + (size_t)job * p.S + (size_t)un * n;
    const uint32_t base = un * n;
    const bool valid = r < M, vec = r < np;
    uint32_t i = r, j = 0;
    if (valid && !vec) {
        uint32_t q = r - np; i = 0;
        while (q >= np - i) { q -= np - i; i++; }
        j = i + q;
    }
    /* r_vec[i] -= (x[s] * x[s-i-1]) * inv;   r_mat[i][j] += (x[s-i-1] * x[s-j-1]) * inv */
    const double *xa = xs + (vec ? np : (np - i - 1u)), *xb = xs + (vec ? (np - i - 1u) : (np - j - 1u));
    double acc = 0.0;
    for (uint32_t s0 = np; s0 < n; s0 += AFM_T) {
        __syncthreads();
        for (uint32_t k = tid; k < AFM_T + np; k += AFM_THREADS) { const uint32_t pos = s0 - np + k; xs[k] = (pos < n) ? af_x(p, layer, cur, job, base + pos) : 0.0; }
        for (uint32_t k = tid; k < AFM_T; k += AFM_THREADS) iv[k] = (s0 + k < n) ? inv[s0 + k] : 0.0;
        __syncthreads();
        const uint32_t cnt = (n - s0 < AFM_T) ? (n - s0) : AFM_T;
        if (!valid) continue;
        if (vec) { for (uint32_t k = 0; k < cnt; k++) acc -= xa[k] * xb[k] * iv[k]; }
        else if (cnt == AFM_T) {
#pragma unroll 8
            for (uint32_t k = 0; k < AFM_T; k++) acc += xa[k] * xb[k] * iv[k];
        } else for (uint32_t k = 0; k < cnt; k++) acc += xa[k] * xb[k] * iv[k];
    }
    if (!valid) return;
    if (vec) p.af_rv[(size_t)job * LNN_MAXP + un * np + i] = acc;
    else p.af_R[(size_t)job * LNN_MAXP * LNN_MAXP + (size_t)un * np * np + (size_t)i * np + j] = acc;
}

/* Cholesky step i, first half: the pivot sum of every live problem of order > i (lpc.c:416-420) */
__global__ void k_af_pivot(Plan p, uint32_t layer, uint32_t i)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *p.af_nprob) return;
    const uint32_t job = p.af_prob[k] / LNN_MAXU, un = p.af_prob[k] % LNN_MAXU;
    const uint32_t u = p.lunits[(size_t)job * LNN_MAXL + layer], np = p.P[layer] / u;
    if (p.af_state[(size_t)job * LNN_MAXU + un] != 0u || i >= np) { p.af_pivot[k] = 1.0; return; }
    const double *A = p.af_R + (size_t)job * LNN_MAXP * LNN_MAXP + (size_t)un * np * np;
    double sum = A[(size_t)i * np + i];
    for (int32_t kk = (int32_t)i - 1; kk >= 0; kk--) sum -= A[(size_t)i * np + kk] * A[(size_t)i * np + kk];
    p.af_pivot[k] = sum;
}

/* second half: af_pivot now holds pow(sum, -0.5) from the host's libm, or a negative number for a non-positive pivot
 * (singular: the coefficients become zeros, lpc.c:612-618); column i below the diagonal (lpc.c:422-428) */
__global__ void k_af_column(Plan p, uint32_t layer, uint32_t i)
{
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    const uint32_t job = p.af_prob[k] / LNN_MAXU, un = p.af_prob[k] % LNN_MAXU;
    const uint32_t u = p.lunits[(size_t)job * LNN_MAXL + layer], np = p.P[layer] / u;
    if (p.af_state[(size_t)job * LNN_MAXU + un] != 0u || i >= np) return;
    const double invd = p.af_pivot[k];
    if (invd < 0.0) { __syncthreads(); if (tid == 0) p.af_state[(size_t)job * LNN_MAXU + un] = 3u; return; }
    double *A = p.af_R + (size_t)job * LNN_MAXP * LNN_MAXP + (size_t)un * np * np;
    if (tid == 0) p.af_invd[(size_t)job * LNN_MAXP + un * np + i] = invd;
    for (uint32_t j = i + 1 + tid; j < np; j += blockDim.x) {
        double sum = A[(size_t)i * np + j];
        for (int32_t kk = (int32_t)i - 1; kk >= 0; kk--) sum -= A[(size_t)i * np + kk] * A[(size_t)j * np + kk];
        A[(size_t)j * np + i] = sum * invd;
    }
}

/* substitutions (lpc.c:431-445), convergence (lpc.c:620-624) */
__global__ void k_af_solve(Plan p, uint32_t layer)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *p.af_nprob) return;
    const uint32_t job = p.af_prob[k] / LNN_MAXU, un = p.af_prob[k] % LNN_MAXU;
    const size_t pi = (size_t)job * LNN_MAXU + un;
    if (p.af_state[pi] != 0u) return;
    const uint32_t u = p.lunits[(size_t)job * LNN_MAXL + layer], np = p.P[layer] / u;
    const double *A = p.af_R + (size_t)job * LNN_MAXP * LNN_MAXP + (size_t)un * np * np;
    const double *b = p.af_rv + (size_t)job * LNN_MAXP + un * np, *invd = p.af_invd + (size_t)job * LNN_MAXP + un * np;
    double *x = p.af_a + (size_t)job * LNN_MAXP + un * np;
    for (int32_t i = 0; i < (int32_t)np; i++) {
        double sum = b[i];
        for (int32_t j = i - 1; j >= 0; j--) sum -= A[(size_t)i * np + j] * x[j];
        x[i] = sum * invd[i];
    }
    for (int32_t i = (int32_t)np - 1; i >= 0; i--) {
        double sum = x[i];
        for (int32_t j = i + 1; j < (int32_t)np; j++) sum -= A[(size_t)j * np + i] * x[j];
        x[i] = sum * invd[i];
    }
    const double obj = p.af_obj[pi];
    if (fabs(p.af_prev[pi] - obj) < 1e-8) p.af_state[pi] = 1u;
    p.af_prev[pi] = obj;
}

/* the refined coefficients back into the layer's parameters, filter order (linne_network.c:368-373); zeros for a singular problem */
__global__ void k_af_finish(Plan p, uint32_t layer)
{
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= p.J) return;
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u;
    double *h = p.lparams + ((size_t)job * LNN_MAXL + layer) * LNN_MAXP;
    const double *a = p.af_a + (size_t)job * LNN_MAXP;
    for (uint32_t un = 0; un < u; un++) {
        const uint32_t st = p.af_state[(size_t)job * LNN_MAXU + un];
        if (st == 2u) continue;                              /* zero problem: the Levinson zeros stand */
        for (uint32_t i = 0; i < np; i++) h[un * np + (np - 1 - i)] = (st == 3u) ? 0.0 : a[un * np + i];
    }
}

#endif

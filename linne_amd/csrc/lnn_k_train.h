/* lnn_k_train.h -- the network trainer (`-l`).
 * Part of the single translation unit lnn_device.hip (included there); not a stand-alone header.
 *
 * Reference: LINNENetworkTrainer_Train, libs/linne_network/src/linne_network.c:805-873 -- up to 2000 steps of momentum SGD
 * (alpha = 0.8f, learning rate 0.1f) on the L1 norm of the cascade's output, over all layers' coefficients at once, started from
 * what the analysis found (linne_encoder.c:669-675); with LINNENetwork_CalculateGradient (:558-579), LINNENetworkLayer_Forward
 * (:165-210), LINNEL1Norm_Loss / _Backward (:50-75), LINNENetworkLayer_Backward (:213-265).  A channel-frame stops when its loss
 * moves by less than 1e-7 between two steps.
 *
 * One step = a handful of launches over all channel-frames that are still training; inside a step every reference sum is one
 * ordered chain owned by one lane: a sample's prediction (taps in order), the loss (samples in order), a coefficient's gradient
 * (10 240 products in order -- 148 chains per channel-frame), a sample's back-propagated signal (taps in order).  Buffers per
 * channel-frame: the inputs of layers 1 .. L-1, and one gradient-signal buffer per layer (the reference's in-place update reads the
 * copy it made in layer->dout: reading one buffer and writing the next is the same thing without the copy, and it leaves every
 * layer's signal in place for the parameter gradients, which are taken in one launch at the end of the backward pass).
 * The operands of every chain are staged in LDS by the whole block (coalesced loads) and read from there.
 */
#ifndef LNN_K_TRAIN_H_INCLUDED
#define LNN_K_TRAIN_H_INCLUDED

struct TrainArgs {
    Plan p;                             /* class tables, xint, scale, lparams / lunits of the jobs */
    const uint32_t *best;               /* [CF] the winning regulariser pass: channel-frame cf trains the parameters of job cf * R + best[cf]; NULL: job = cf (R = 1) */
    double *buf;                        /* [CF][TR_NBUF][S]: inputs of layers 1 and 2, the gradient signal in front of layers 2 / 1 / 0 (the first also holds the cascade's output) */
    double *dparams, *momentum;         /* [CF][MAXL][MAXP] */
    double *loss, *prev;                /* [CF] */
    uint32_t *active;                   /* [CF] 1 while the channel-frame trains */
    uint32_t *nactive;                  /* count of channel-frames that go on after this step */
    uint32_t CF;
};
#define TR_IN1 0
#define TR_IN2 1
#define TR_G0  2                        /* gradient signal at the cascade's output (= behind the last layer) */
#define TR_NBUF (2 + LNN_MAXL)
#define TR_THREADS 256
#define TR_TILE (TR_THREADS * 4)        /* output samples per block of the forward / backward kernels */
#define TR_GT 512                       /* chain steps per staged tile of k_tr_gradp */

__device__ __forceinline__ uint32_t tr_job(const TrainArgs &a, uint32_t cf) { return a.best ? cf * a.p.R + a.best[cf] : cf; }
/* input sample s of layer `layer` of channel-frame cf */
__device__ __forceinline__ double tr_in(const TrainArgs &a, uint32_t layer, uint32_t cf, uint32_t s)
{
    const Plan &p = a.p;
    return (layer == 0) ? ((double)p.xint[(size_t)cf * p.S + s] * p.scale) : a.buf[((size_t)cf * TR_NBUF + (layer == 1 ? TR_IN1 : TR_IN2)) * p.S + s];
}
__device__ __forceinline__ double *tr_out(const TrainArgs &a, uint32_t layer, uint32_t cf)      /* where layer `layer` writes its output */
{
    const Plan &p = a.p;
    const uint32_t which = (layer + 1 == p.L) ? TR_G0 : (layer == 0 ? TR_IN1 : TR_IN2);
    return a.buf + ((size_t)cf * TR_NBUF + which) * p.S;
}
/* the gradient signal behind layer `layer` (what its Backward reads): G0 for the last layer, then one buffer further per layer */
__device__ __forceinline__ double *tr_grad(const TrainArgs &a, uint32_t layer, uint32_t cf)
{
    return a.buf + ((size_t)cf * TR_NBUF + TR_G0 + (a.p.L - 1u - layer)) * a.p.S;
}

__global__ void k_tr_init(TrainArgs a)
{
    const uint32_t cf = blockIdx.x * blockDim.x + threadIdx.x;
    if (cf >= a.CF) return;
    for (uint32_t i = 0; i < LNN_MAXL * LNN_MAXP; i++) a.momentum[(size_t)cf * LNN_MAXL * LNN_MAXP + i] = 0.0;
    a.prev[cf] = (double)FLT_MAX; a.active[cf] = 1u;
}

/* LINNENetworkLayer_Forward (linne_network.c:165-210): out[s] = in[s] + predict, predict = 0.0 + h[0] in[s-np] + ... in tap order;
 * zeros stand in front of the frame's first sample (unit 0's ramp skips those taps: adding +-0.0 first changes nothing), the
 * first sample itself is copied.  One block = TR_TILE outputs of one channel-frame; their inputs and the up to 128 samples in
 * front of them are staged in LDS once (coalesced), the taps read from there. */
__global__ __launch_bounds__(TR_THREADS) void k_tr_forward(TrainArgs a, uint32_t layer)
{
    __shared__ double sh[LNN_MAXP];
    __shared__ double xin[LNN_MAXP + TR_TILE];          /* xin[k] = in[s0 - LNN_MAXP + k] */
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    if (!a.active[cf]) return;
    const uint32_t job = tr_job(a, cf);
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t s0 = blockIdx.y * TR_TILE;
    if (s0 >= c.na) return;
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    for (uint32_t i = tid; i < P; i += TR_THREADS) sh[i] = p.lparams[((size_t)job * LNN_MAXL + layer) * LNN_MAXP + i];
    for (uint32_t k = tid; k < LNN_MAXP + TR_TILE; k += TR_THREADS) {
        const int64_t g = (int64_t)s0 - LNN_MAXP + k;
        xin[k] = (g >= 0 && g < (int64_t)c.na) ? tr_in(a, layer, cf, (uint32_t)g) : 0.0;
    }
    __syncthreads();
    double *out = tr_out(a, layer, cf);
    for (uint32_t r = 0; r < 4u; r++) {
        const uint32_t k = r * TR_THREADS + tid, s = s0 + k;
        if (s >= c.na) break;
        const double x = xin[LNN_MAXP + k];
        if (s == 0) { out[0] = x; continue; }
        const double *h = sh + (s / n) * np, *w = xin + LNN_MAXP + k - np;
        double predict = 0.0;
        for (uint32_t j = 0; j < np; j++) predict += h[j] * w[j];
        out[s] = x + predict;
    }
}

/* LINNEL1Norm_Loss (:50-63) -- one ordered chain per channel-frame -- and LINNEL1Norm_Backward (:66-75: sign / n, in place) in one
 * pass: a wave per channel-frame loads 256 outputs at a time (coalesced), leaves their signs / n in their place and their
 * magnitudes in LDS, from where the chain adds them up in order (every lane runs the same chain: nothing to exchange). */
__global__ __launch_bounds__(64) void k_tr_loss(TrainArgs a)
{
    __shared__ __attribute__((aligned(16))) double mag[256];
    const uint32_t cf = blockIdx.x, lane = threadIdx.x;
    if (!a.active[cf]) return;
    const Plan &p = a.p;
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t na = c.na;
    double *o = a.buf + ((size_t)cf * TR_NBUF + TR_G0) * p.S;
    const double inv = 1.0 / (double)na;     /* (sign / n: +-1.0 / n and 0.0 / n are what the reference's division gives) */
    double norm = 0.0;
    for (uint32_t s0 = 0; s0 < na; s0 += 256u) {
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++) {
            const uint32_t s = s0 + r * 64u + lane;
            double d = 0.0;
            if (s < na) { d = o[s]; o[s] = (d > 0.0) ? inv : ((d < 0.0) ? -inv : 0.0); }
            mag[r * 64u + lane] = fabs(d);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      /* lgkmcnt(0): the tile is in LDS */
        const uint32_t cnt = (na - s0 < 256u) ? (na - s0) : 256u;
        if (cnt == 256u) {
#pragma unroll 8
            for (uint32_t k = 0; k < 256u; k += 2u) { const lnn_d2 v = *(const lnn_d2 *)(mag + k); norm += v.x; norm += v.y; }
        } else for (uint32_t k = 0; k < cnt; k++) norm += mag[k];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) a.loss[cf] = norm / (double)na;
}

/* LINNENetworkLayer_Backward, parameter gradients (:237-243): dparams[un np + i] = sum_j in[un n + j] * dout[un n + np - i + j],
 * j = 0 .. n - np + i - 1, one ordered chain per coefficient.  One block per (channel-frame, layer), lanes = taps of one unit, the
 * units one after the other (the chains of all units together are as long as the frame, whatever the unit count); the operands
 * of TR_GT steps are staged in LDS: in[] is read by all lanes at the same address, dout[] at consecutive ones. */
__global__ __launch_bounds__(128) void k_tr_gradp(TrainArgs a)
{
    __shared__ double sin_[TR_GT];
    __shared__ double sp[TR_GT + LNN_MAXP + 1];
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, layer = blockIdx.y, tid = threadIdx.x;
    if (layer >= p.L || !a.active[cf]) return;
    const uint32_t job = tr_job(a, cf);
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    const double *dout = tr_grad(a, layer, cf);
    double *dp = a.dparams + ((size_t)cf * LNN_MAXL + layer) * LNN_MAXP;
    for (uint32_t un = 0; un < u; un++) {
        const uint32_t base = un * n;
        const uint32_t len = (tid < np) ? (n - np + tid) : 0u;     /* this lane's chain */
        const double *w = sp + ((tid < np) ? (np - tid) : 0u);      /* w[jj] = dout[base + j0 + np - i + jj] */
        double acc = 0.0;
        for (uint32_t j0 = 0; j0 < n; j0 += TR_GT) {
            __syncthreads();
            for (uint32_t k = tid; k < TR_GT; k += 128u) sin_[k] = (j0 + k < n) ? tr_in(a, layer, cf, base + j0 + k) : 0.0;
            for (uint32_t k = tid; k < TR_GT + np + 1u; k += 128u) sp[k] = (j0 + k < n) ? dout[base + j0 + k] : 0.0;
            __syncthreads();
            if (tid < np) {
                if (j0 + TR_GT <= len) {
#pragma unroll 8
                    for (uint32_t jj = 0; jj < TR_GT; jj++) acc += sin_[jj] * w[jj];
                } else if (j0 < len) {
                    const uint32_t m = len - j0;
                    for (uint32_t jj = 0; jj < m; jj++) acc += sin_[jj] * w[jj];
                }
            }
        }
        if (tid < np) dp[un * np + tid] = acc;
    }
}

/* LINNENetworkLayer_Backward, the signal (:246-263): gdst[i] = gsrc[i] + (sum_j h[j] gsrc[np + i - j]) / np, terms beyond the unit
 * dropped.  Reads the signal behind layer `layer`, writes the one in front of it. */
__global__ __launch_bounds__(TR_THREADS) void k_tr_back(TrainArgs a, uint32_t layer)
{
    __shared__ double sh[LNN_MAXP];
    __shared__ double g[TR_TILE + LNN_MAXP + 1];        /* g[k] = gsrc[s0 + k] */
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    if (!a.active[cf]) return;
    const uint32_t job = tr_job(a, cf);
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t s0 = blockIdx.y * TR_TILE;
    if (s0 >= c.na) return;
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    for (uint32_t i = tid; i < P; i += TR_THREADS) sh[i] = p.lparams[((size_t)job * LNN_MAXL + layer) * LNN_MAXP + i];
    const double *src = tr_grad(a, layer, cf);
    double *dst = tr_grad(a, layer - 1u, cf);
    for (uint32_t k = tid; k < TR_TILE + LNN_MAXP + 1u; k += TR_THREADS) g[k] = (s0 + k < c.na) ? src[s0 + k] : 0.0;
    __syncthreads();
    for (uint32_t r = 0; r < 4u; r++) {
        const uint32_t k = r * TR_THREADS + tid, s = s0 + k;
        if (s >= c.na) break;
        const uint32_t un = s / n, i = s - un * n;
        const double *h = sh + un * np, *w = g + k + np;     /* w[-j] = gsrc[s + np - j] */
        double back = 0.0;
        for (uint32_t j = 0; j < np; j++) if (np + i - j < n) back += h[j] * w[-(int32_t)j];
        dst[s] = g[k] + back / (double)np;
    }
}

/* the momentum step (:843-850) and the stop test (:868-872) */
__global__ void k_tr_update(TrainArgs a, double alpha, double lr, double eps)
{
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    if (!a.active[cf]) return;
    const uint32_t job = tr_job(a, cf);
    for (uint32_t l = 0; l < p.L; l++)
        for (uint32_t i = tid; i < p.P[l]; i += blockDim.x) {
            const size_t k = ((size_t)cf * LNN_MAXL + l) * LNN_MAXP + i;
            const double m = alpha * a.momentum[k] + lr * a.dparams[k];
            a.momentum[k] = m;
            p.lparams[((size_t)job * LNN_MAXL + l) * LNN_MAXP + i] -= m;
        }
    __syncthreads();
    if (tid == 0) {
        const double loss = a.loss[cf];
        if (fabs(loss - a.prev[cf]) < eps) a.active[cf] = 0u; else atomicAdd(a.nactive, 1u);
        a.prev[cf] = loss;
    }
}

#endif

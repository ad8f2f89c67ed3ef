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
 * channel-frame: the inputs of layers 1 .. L-1, and two gradient-signal buffers (the reference's in-place update reads the copy it
 * made in layer->dout: reading one buffer and writing the other is the same thing without the copy).
 */
#ifndef LNN_K_TRAIN_H_INCLUDED
#define LNN_K_TRAIN_H_INCLUDED

struct TrainArgs {
    Plan p;                             /* class tables, xint, scale, lparams / lunits of the jobs */
    const uint32_t *best;               /* [CF] the winning regulariser pass: channel-frame cf trains the parameters of job cf * R + best[cf]; NULL: job = cf (R = 1) */
    double *buf;                        /* [CF][4][S]: inputs of layers 1 and 2, gradient signal A / B (A also holds the cascade's output) */
    double *dparams, *momentum;         /* [CF][MAXL][MAXP] */
    double *loss, *prev;                /* [CF] */
    uint32_t *active;                   /* [CF] 1 while the channel-frame trains */
    uint32_t *nactive;                  /* count of channel-frames that go on after this step */
    uint32_t CF;
};
#define TR_IN1 0
#define TR_IN2 1
#define TR_GA  2
#define TR_GB  3
#define TR_THREADS 256

__device__ __forceinline__ uint32_t tr_job(const TrainArgs &a, uint32_t cf) { return a.best ? cf * a.p.R + a.best[cf] : cf; }
/* input sample s of layer `layer` of channel-frame cf */
__device__ __forceinline__ double tr_in(const TrainArgs &a, uint32_t layer, uint32_t cf, uint32_t s)
{
    const Plan &p = a.p;
    return (layer == 0) ? ((double)p.xint[(size_t)cf * p.S + s] * p.scale) : a.buf[((size_t)cf * 4 + (layer == 1 ? TR_IN1 : TR_IN2)) * p.S + s];
}
__device__ __forceinline__ double *tr_out(const TrainArgs &a, uint32_t layer, uint32_t cf)      /* where layer `layer` writes its output */
{
    const Plan &p = a.p;
    const uint32_t which = (layer + 1 == p.L) ? TR_GA : (layer == 0 ? TR_IN1 : TR_IN2);
    return a.buf + ((size_t)cf * 4 + which) * p.S;
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
 * first sample itself is copied */
__global__ __launch_bounds__(TR_THREADS) void k_tr_forward(TrainArgs a, uint32_t layer)
{
    __shared__ double sh[LNN_MAXP];
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    if (!a.active[cf]) return;
    const uint32_t job = tr_job(a, cf);
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    for (uint32_t i = tid; i < P; i += TR_THREADS) sh[i] = p.lparams[((size_t)job * LNN_MAXL + layer) * LNN_MAXP + i];
    __syncthreads();
    double *out = tr_out(a, layer, cf);
    for (uint32_t s = blockIdx.y * TR_THREADS * 4u + tid; s < c.na && s < (blockIdx.y + 1u) * TR_THREADS * 4u; s += TR_THREADS) {
        const double x = tr_in(a, layer, cf, s);
        if (s == 0) { out[0] = x; continue; }
        const double *h = sh + (s / n) * np;
        double predict = 0.0;
        for (uint32_t j = 0; j < np; j++) { const int64_t g = (int64_t)s - np + j; predict += h[j] * ((g >= 0) ? tr_in(a, layer, cf, (uint32_t)g) : 0.0); }
        out[s] = x + predict;
    }
}

/* LINNEL1Norm_Loss (:50-63): one ordered chain per channel-frame */
__global__ void k_tr_loss(TrainArgs a)
{
    const uint32_t cf = blockIdx.x * blockDim.x + threadIdx.x;
    if (cf >= a.CF || !a.active[cf]) return;
    const Plan &p = a.p;
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const double *o = a.buf + ((size_t)cf * 4 + TR_GA) * p.S;
    double norm = 0.0;
    for (uint32_t s = 0; s < c.na; s++) norm += fabs(o[s]);
    a.loss[cf] = norm / (double)c.na;
}

/* LINNEL1Norm_Backward (:66-75): sign / n, in place */
__global__ void k_tr_l1back(TrainArgs a)
{
    const uint32_t cf = blockIdx.x;
    if (!a.active[cf]) return;
    const Plan &p = a.p;
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    double *o = a.buf + ((size_t)cf * 4 + TR_GA) * p.S;
    for (uint32_t s = blockIdx.y * TR_THREADS * 4u + threadIdx.x; s < c.na && s < (blockIdx.y + 1u) * TR_THREADS * 4u; s += TR_THREADS) {
        const double d = o[s];
        o[s] = (double)((d > 0.0) - (d < 0.0)) / (double)c.na;
    }
}

/* LINNENetworkLayer_Backward, parameter gradients (:237-243): one chain per coefficient */
__global__ void k_tr_gradp(TrainArgs a, uint32_t layer, uint32_t gsrc)
{
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, e = blockIdx.y * blockDim.x + threadIdx.x;
    if (!a.active[cf]) return;
    const uint32_t P = p.P[layer];
    if (e >= P) return;
    const uint32_t job = tr_job(a, cf);
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    const uint32_t un = e / np, i = e - un * np, base = un * n;
    const double *pout = a.buf + ((size_t)cf * 4 + gsrc) * p.S + base;
    double acc = 0.0;
    for (uint32_t j = 0; j < n - np + i; j++) acc += tr_in(a, layer, cf, base + j) * pout[np - i + j];
    a.dparams[((size_t)cf * LNN_MAXL + layer) * LNN_MAXP + e] = acc;
}

/* LINNENetworkLayer_Backward, the signal (:246-263): gdst[i] = gsrc[i] + (sum_j h[j] gsrc[np + i - j]) / np, terms beyond the unit dropped */
__global__ __launch_bounds__(TR_THREADS) void k_tr_back(TrainArgs a, uint32_t layer, uint32_t gsrc, uint32_t gdst)
{
    __shared__ double sh[LNN_MAXP];
    const Plan &p = a.p;
    const uint32_t cf = blockIdx.x, tid = threadIdx.x;
    if (!a.active[cf]) return;
    const uint32_t job = tr_job(a, cf);
    const DevClass &c = p.cls[p.cls_of_frame[cf / p.C]];
    const uint32_t P = p.P[layer], u = p.lunits[(size_t)job * LNN_MAXL + layer], np = P / u, n = c.na / u;
    for (uint32_t i = tid; i < P; i += TR_THREADS) sh[i] = p.lparams[((size_t)job * LNN_MAXL + layer) * LNN_MAXP + i];
    __syncthreads();
    const double *src = a.buf + ((size_t)cf * 4 + gsrc) * p.S;
    double *dst = a.buf + ((size_t)cf * 4 + gdst) * p.S;
    for (uint32_t s = blockIdx.y * TR_THREADS * 4u + tid; s < c.na && s < (blockIdx.y + 1u) * TR_THREADS * 4u; s += TR_THREADS) {
        const uint32_t un = s / n, i = s - un * n;
        const double *h = sh + un * np, *pout = src + un * n;
        double back = 0.0;
        for (uint32_t j = 0; j < np; j++) if (np + i - j < n) back += h[j] * pout[np + i - j];
        dst[s] = src[s] + back / (double)np;
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

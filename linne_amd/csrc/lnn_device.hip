/*
 * lnn_device.hip -- gfx950 (MI355X) kernels of the LINNE per-frame prediction path and their C-ABI
 * launchers (declared in include/linne_amd.h).
 *
 * Bit-exactness rules (DESIGN.md "Arithmetic contract"):
 *   - this TU is compiled with -ffp-contract=off: every double multiply and add is a separate, correctly
 *     rounded IEEE-754 operation, issued in the reference's order.  A sum that the reference evaluates as
 *     one chain is owned by ONE thread here; parallelism comes only from independent chains (lags, units,
 *     unit-count trials, regulariser passes, samples, channels, frames).
 *   - libm values the reference takes from glibc (Welch divisor pow(n-1,-2), the SIN window) are computed
 *     on the host and passed in as tables.
 *   - int32 filters use 32-bit wrap-around arithmetic (uint32 multiply/add, arithmetic shift).
 *
 * Citations are file:line under /root/reference.
 */
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "linne_amd.h"
#include "lnn_common.h"

/* the kernels, in pipeline order */
#include "lnn_dev_common.h"
#define LEV_LDS(np_) (sizeof(double) * 64 * (size_t)(2 * (np_) + 3))       /* LDS columns of one Levinson problem set of order np_ */
#define LEV_LDS_BUDGET ((size_t)160 * 1024)                                   /* LDS of a CU */
#include "lnn_k_prep.h"
#include "lnn_k_autocorr.h"
#include "lnn_k_levinson.h"
#include "lnn_k_fir.h"
#include "lnn_k_search.h"
#include "lnn_k_fwdloss.h"
#include "lnn_k_lastlayer.h"
#include "lnn_k_af.h"
#include "lnn_k_train.h"
#include "lnn_k_decode.h"
#include "lnn_k_decode_rows.h"
#include "lnn_k_decode_fused.h"
#include "lnn_k_finalize.h"
#include "lnn_k_rice.h"

/* ================================================================================================
 * host side of this TU: context, scratch arena, launch sequences, C-ABI
 * ============================================================================================== */
#define LNN_RICE_STREAMS 4
struct LINNEAmdContext {
    int device;
    hipStream_t stream; int own_stream;
    void *arena; uint64_t arena_bytes;
    char err[256];
    int timing;
    uint32_t na_max;                    /* largest analysis length of the current batch */
    hipEvent_t ev[2]; int ev_valid;
    /* per-kernel spans of the last call (timing enabled): HIP events on the launch stream */
    hipEvent_t *span_ev; int *span_kind; int nspans, span_cap;
    /* cached class tables */
    uint32_t *d_ucount;
    /* frame groups of one call rotate over these streams so that the latency-bound phases of one group (short
     * layers, Levinson, ordered sums) overlap the throughput-bound phases of another */
    hipStream_t sub[LNN_MAXSUB]; hipEvent_t sub_done[LNN_MAXSUB]; hipEvent_t ev_start; int nsub, nsub_forced /* LINNE_AMD_STREAMS was given */;
    int enc_streams_done;
    hipStream_t side; hipEvent_t side_done; int has_side;     /* block-type statistics run beside the analysis */
    hipEvent_t fork_ev, join_ev;        /* side stream: the general autocorrelation kernel for the few frames the lanes = jobs kernels do not take */
    DevClass *d_cls; double *d_sin; uint64_t sin_cap; double *d_wt; uint64_t wt_cap; uint32_t *d_clsidx; uint64_t clsidx_cap; uint32_t *d_map;   /* class index per sorted row, then the sorted row's frame (same buffer) */ uint32_t *d_nsmp; uint64_t nsmp_cap;
    /* what the resident class tables were built for: a call with the same shape and frame lengths re-uses them */
    DevClass sig_cls[LNN_MAXCLS]; struct LINNEAmdShape sig_shape; int sig_valid; uint32_t sig_ncls; uint64_t sig_sin_total, sig_wt_total;
    /* pinned ring for the per-call frame metadata (class index, length), so that a call enqueues without a host sync */
    int fwd_loss;                       /* LINNE_AMD_FWD_LOSS: last layer's forward pass and loss in one kernel (k_fwd_loss); -1 = by batch size */
    int lev_ride;                       /* short Levinson trials ride along with the one-unit trial (LINNE_AMD_LEV_RIDE, default 1) */
    int lev_wave;                       /* batches of <= 64 jobs: a wave per Levinson problem (LINNE_AMD_LEV_WAVE, default 1) */
    int search_two;                     /* k_search_long in two passes over the window, five waves per SIMD (LINNE_AMD_SEARCH_TWO) */
    const uint32_t *cur_idx;            /* class index per frame of the call being enqueued (host copy, in the meta ring) */
    uint32_t *meta_h[LNN_META]; uint64_t meta_cap[LNN_META]; hipEvent_t meta_ev[LNN_META]; int meta_used[LNN_META]; int meta_next;
    /* copy streams of the staging slots (H2D of the next group and D2H of the previous one overlap the kernels) */
    hipStream_t copy_in, copy_out; int has_copy;
    /* decode slots in stream mode: the Rice decoders of a stream's groups run side by side on these (a launch is a few dozen waves
     * walking their blocks for ~8 ms), the slots take them in turn */
    hipStream_t rice_pool[LNN_RICE_STREAMS]; int n_rice_pool, rice_next;
    uint32_t *d_plan_nsmp; uint64_t plan_nsmp_cap; double rice_steps[32]; uint32_t rice_nsteps;
    int prod_ok;                        /* set per batch by build_classes, bit l: in layer l every class has all its trials and even unit lengths (k_autocorr_prod) */
    int fir_small;                      /* LINNE_AMD_FIR_SMALL (default 1): register-window search kernel for layers of <= 16 taps */
    uint32_t learning;                  /* -l: the SGD trainer after the analysis (LINNEAmd_SetLearning), 0 = off */
    uint32_t af_iters;                  /* -a N: auxiliary-function iterations of the final pass (LINNEAmd_SetAfIterations), 0 = off */
    double *af_h; uint32_t af_h_cap;    /* pinned: a Cholesky step's pivots on their way through the host's pow() */
    int pcm16_next;                     /* the next EncodeFramesDevice call reads narrow samples: 1 int16, 2 packed 3-byte (set by the staging slots, cleared by the call) */
    int force_exact;                    /* LINNE_AMD_EXACT=1: every unit-count search runs the exact ordered chains (diff against the certified search) */
    int fir_spec;                       /* LINNE_AMD_SPECULATE (default 1): fuse the one-unit forward into the search of layers 0 .. L-2 */
    void *hstage; uint64_t hstage_cap;  /* device staging of the host-buffer forms (EncodeFramesHost / DecodeFramesHost: block-at-a-time calls), kept between calls */
    /* debug / test knobs that select a kernel form per CALL (read_call_knobs: once at the top of an encode / decode call, never
     * inside the chunk loop; production never sets them and gets the batch-size rules) */
    struct { int sort, l0_products, wide, search_long, rows16, prep_general, stats_rows /* -1 = by batch size */, hist /* -1 = by batch size */, decode_kernel /* 0 = by batch size, 1 = wave, 2 = lanes, 3 = pipe, 4 = rows */, rows8, streams /* LINNE_AMD_STREAMS of this call, 0 = not given */, nostats, fwd_loss_mw /* LINNE_AMD_FWD_LOSS_MW (default 1): chunks below 65 536 jobs take the five-wave form of k_fwd_loss */, prep_defer /* LINNE_AMD_PREP_DEFER (default 1): inexact pre-emphasis sums go to k_prep_slow */, last_layer /* LINNE_AMD_LAST_LAYER (default 1): the last layer's search, forward pass and loss in one launch (k_last_layer) where it takes the chunk */, decode_fused /* LINNE_AMD_DECODE_FUSED (default 1): layer 0 + de-emphasis + MS -> LR in one launch */; uint32_t dbg_maxtr; } knob;
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return LNN_NG; } } while (0)

static const uint32_t k_layers_a[] = { 2, 32 }, k_layers_b[] = { 4, 64, 8 }, k_layers_c[] = { 4, 128, 16 };
static const double k_regs_1[] = { 0.0 }, k_regs_2[] = { 0.0, 1.0 / 512.0 }, k_regs_4[] = { 0.0, 1.0 / 2048.0, 1.0 / 512.0, 1.0 / 128.0 };

extern "C" int lnn_preset_info(uint32_t preset, uint32_t *num_layers, uint32_t *layers, uint32_t *num_regs, double *regs)
{   /* libs/linne_internal/src/linne_internal.c:16-41 */
    const uint32_t *L; const double *R; uint32_t nl, nr;
    if (preset >= 8) return -1;
    if (preset < 2) { L = k_layers_a; nl = 2; } else if (preset < 5) { L = k_layers_b; nl = 3; } else { L = k_layers_c; nl = 3; }
    switch (preset) { case 0: case 2: case 5: R = k_regs_1; nr = 1; break; case 1: case 3: case 6: R = k_regs_2; nr = 2; break; default: R = k_regs_4; nr = 4; }
    if (num_layers) *num_layers = nl;
    if (layers) for (uint32_t i = 0; i < nl; i++) layers[i] = L[i];
    if (num_regs) *num_regs = nr;
    if (regs) for (uint32_t i = 0; i < nr; i++) regs[i] = R[i];
    return 0;
}

/* A HIP stream is multiplexed onto one of a few hardware queues -- four per process by default -- and streams that share a
 * queue execute in submission order whatever their events allow.  A context pipelines over four streams (analysis, block-type
 * statistics, copy-in, copy-out) next to whatever the host application uses (torch has its own); when copy-in and copy-out
 * shared a queue, the H2D of group g + 1 sat behind the Rice emission of group g, which waits for the analysis of g: the
 * staging pipeline ran serially (measured: 28 ms per group instead of 23).  The runtime reads GPU_MAX_HW_QUEUES when it
 * initialises, i.e. at the first HIP call of the process; loading this library comes before that.  A value the user set is
 * left alone.  Round 4 tried more: DecodeWhole keeps eight groups in flight, and with a stream per slot 8 queues were too few -- a
 * slot's stream shared one with the copy-out stream and its Rice decoder started 3 ms late behind another group's D2H
 * (profiles/r04_decode_timeline.txt; 55 ms per 60-minute stream with 8 queues, 42 with 16, 37 with 24) -- but with 24 queues
 * EncodeWhole lost a quarter of its rate inside a process that holds other contexts (bench.py: 142 k -> 107 k frames/s, every
 * kernel of the analysis 6 % slower: profiles/r04_encode_whole_queues.txt).  So the count stays at 8 and the decoder asks for LESS:
 * its context creates no encode-side streams (ctx_encode_streams) and four pooled streams carry the Rice decoders
 * (LNN_RICE_STREAMS), so that what runs side by side in a decode -- synthesis, copy-in, copy-out, four decoders -- are seven streams
 * created one after the other: seven different queues out of eight. */
__attribute__((constructor)) static void lnn_more_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" int LINNEAmd_GetDeviceCount(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* The streams only the ENCODE side uses -- the compute sub-streams of a large call and the side stream of the block-type statistics --
 * are created at the first encode call, not with the context: a stream takes the next of the process's few hardware queues in turn
 * (GPU_MAX_HW_QUEUES), and a decoder's context that never encodes should not push its own streams -- the synthesis, the copy-out, the
 * Rice decoders -- onto queues that collide (round 4: DecodeWhole's Rice decoders started milliseconds late behind another group's D2H). */
static int ctx_encode_streams(LINNEAmdContext *ctx)
{
    if (ctx->enc_streams_done) return LNN_OK;
    ctx->enc_streams_done = 1;
    {
        const char *env = getenv("LINNE_AMD_STREAMS");
        int ns = env ? atoi(env) : 2;                /* (two by default since round 3: see the rule at the chunk loop) */
        ctx->nsub_forced = env != NULL;
        if (ns < 1) ns = 1;
        if (ns > LNN_MAXSUB) ns = LNN_MAXSUB;
        /* One stream of a process maps onto one of a few hardware queues (four by default); streams that share a queue run in
         * order whatever their events say.  A context therefore creates as few streams as it needs: two compute sub-streams
         * (LINNE_AMD_STREAMS=1: none -- the analysis then always runs on the context's own stream, as small batches do anyway). */
        ctx->nsub = 0;
        for (int i = 0; i < ns && ns >= 2; i++) {
            if (hipStreamCreateWithFlags(&ctx->sub[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->sub_done[i], hipEventDisableTiming) != hipSuccess) break;
            ctx->nsub++;
        }
        const bool have_start = hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming) == hipSuccess;
        if (!have_start) ctx->nsub = 0;
        ctx->has_side = have_start && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) == hipSuccess
                && hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming) == hipSuccess
                && hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming) == hipSuccess
                && hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming) == hipSuccess;
    }
    return LNN_OK;
}

extern "C" struct LINNEAmdContext *LINNEAmd_ContextCreate(int device, uint64_t scratch_bytes)
{
    int n = 0;
    hipError_t e;
#define CC_FAIL(what) do { fprintf(stderr, "liblinne_amd: ContextCreate(device=%d): %s: %s\n", device, what, hipGetErrorString(e)); } while (0)
    if ((e = hipGetDeviceCount(&n)) != hipSuccess) { CC_FAIL("hipGetDeviceCount"); return NULL; }
    if (n <= 0 || device < 0 || device >= n) { fprintf(stderr, "liblinne_amd: ContextCreate(device=%d): %d HIP device(s) visible\n", device, n); return NULL; }
    if ((e = hipSetDevice(device)) != hipSuccess) { CC_FAIL("hipSetDevice"); return NULL; }
    LINNEAmdContext *ctx = (LINNEAmdContext *)calloc(1, sizeof(*ctx));
    if (!ctx) return NULL;
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) { CC_FAIL("hipStreamCreate"); free(ctx); return NULL; }
    ctx->own_stream = 1;
    if (scratch_bytes == 0) scratch_bytes = 6ull << 30;
    if ((e = hipMalloc(&ctx->arena, scratch_bytes)) != hipSuccess) { CC_FAIL("hipMalloc(arena)"); hipStreamDestroy(ctx->stream); free(ctx); return NULL; }
    ctx->arena_bytes = scratch_bytes;
    if ((e = hipMalloc((void **)&ctx->d_cls, sizeof(DevClass) * LNN_MAXCLS)) != hipSuccess) { CC_FAIL("hipMalloc(classes)"); hipFree(ctx->arena); hipStreamDestroy(ctx->stream); free(ctx); return NULL; }
    if ((e = hipMalloc((void **)&ctx->d_ucount, 4 * sizeof(uint32_t))) != hipSuccess) { CC_FAIL("hipMalloc(counter)"); }
    { const char *ex = getenv("LINNE_AMD_EXACT"); ctx->force_exact = ex ? atoi(ex) : 0; }
    { const char *sp = getenv("LINNE_AMD_SPECULATE"); ctx->fir_spec = sp ? atoi(sp) : 1; }
    { const char *lr = getenv("LINNE_AMD_LEV_RIDE"); ctx->lev_ride = lr ? atoi(lr) : 1; }
    { const char *lw = getenv("LINNE_AMD_LEV_WAVE"); ctx->lev_wave = lw ? atoi(lw) : 1; }
    { const char *st_ = getenv("LINNE_AMD_SEARCH_TWO"); ctx->search_two = st_ ? atoi(st_) : 1; }
    { const char *fl = getenv("LINNE_AMD_FWD_LOSS"); ctx->fwd_loss = fl ? atoi(fl) : -1; }
    { const char *sp = getenv("LINNE_AMD_FIR_SMALL"); ctx->fir_small = sp ? atoi(sp) : 1; }
    (void)hipFuncSetAttribute((const void *)k_levinson_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEV_LDS_BUDGET);
    (void)hipFuncSetAttribute((const void *)k_synth_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEV_LDS_BUDGET);
    if ((e = hipEventCreate(&ctx->ev[0])) != hipSuccess || (e = hipEventCreate(&ctx->ev[1])) != hipSuccess) { CC_FAIL("hipEventCreate"); }
#undef CC_FAIL
    return ctx;
}

extern "C" void LINNEAmd_ContextDestroy(struct LINNEAmdContext *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->d_cls) hipFree(ctx->d_cls);
    if (ctx->d_ucount) hipFree(ctx->d_ucount);
    for (int i = 0; i < ctx->nsub; i++) { hipStreamSynchronize(ctx->sub[i]); hipStreamDestroy(ctx->sub[i]); hipEventDestroy(ctx->sub_done[i]); }
    if (ctx->ev_start) hipEventDestroy(ctx->ev_start);
    if (ctx->has_side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); hipEventDestroy(ctx->side_done); hipEventDestroy(ctx->fork_ev); hipEventDestroy(ctx->join_ev); }
    if (ctx->d_sin) hipFree(ctx->d_sin);
    if (ctx->d_wt) hipFree(ctx->d_wt);
    if (ctx->d_clsidx) hipFree(ctx->d_clsidx);
    if (ctx->d_nsmp) hipFree(ctx->d_nsmp);
    if (ctx->d_plan_nsmp) hipFree(ctx->d_plan_nsmp);
    if (ctx->hstage) hipFree(ctx->hstage);
    if (ctx->af_h) hipHostFree(ctx->af_h);
    for (int i = 0; i < LNN_META; i++) { if (ctx->meta_h[i]) hipHostFree(ctx->meta_h[i]); if (ctx->meta_ev[i]) hipEventDestroy(ctx->meta_ev[i]); }
    for (int i = 0; i < ctx->n_rice_pool; i++) { hipStreamSynchronize(ctx->rice_pool[i]); hipStreamDestroy(ctx->rice_pool[i]); }
    if (ctx->has_copy) { hipStreamSynchronize(ctx->copy_in); hipStreamSynchronize(ctx->copy_out); hipStreamDestroy(ctx->copy_in); hipStreamDestroy(ctx->copy_out); }
    hipEventDestroy(ctx->ev[0]); hipEventDestroy(ctx->ev[1]);
    for (int i = 0; i < 2 * ctx->span_cap; i++) hipEventDestroy(ctx->span_ev[i]);
    free(ctx->span_ev); free(ctx->span_kind);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    free(ctx);
}

extern "C" const char *LINNEAmd_GetLastError(const struct LINNEAmdContext *ctx) { return ctx ? ctx->err : "no context"; }

extern "C" int LINNEAmd_SetStream(struct LINNEAmdContext *ctx, void *hip_stream)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { HIPCHK(ctx, hipStreamDestroy(ctx->stream)); ctx->own_stream = 0; }
    ctx->stream = (hipStream_t)hip_stream;              /* NULL is the device's default (null) stream */
    return LNN_OK;
}

extern "C" int LINNEAmd_ReserveScratch(struct LINNEAmdContext *ctx, uint64_t bytes)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    if (ctx->arena_bytes >= bytes) return LNN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    void *fresh = NULL;
    if (hipMalloc(&fresh, bytes) != hipSuccess) { (void)hipGetLastError(); snprintf(ctx->err, sizeof(ctx->err), "cannot reserve %llu bytes of scratch", (unsigned long long)bytes); return LNN_NG; }
    if (ctx->arena) HIPCHK(ctx, hipFree(ctx->arena));
    ctx->arena = fresh; ctx->arena_bytes = bytes;
    return LNN_OK;
}

extern "C" int64_t LINNEAmd_GetLastFallbackCount(struct LINNEAmdContext *ctx)
{
    if (!ctx) return -1;
    uint32_t v = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess
            || hipMemcpy(&v, ctx->d_ucount, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

extern "C" double LINNEAmd_GetLastMinMargin(struct LINNEAmdContext *ctx)
{
    if (!ctx) return -1.0;
    double v = -1.0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess
            || hipMemcpy(&v, ctx->d_ucount + 2, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    return v;
}

extern "C" int LINNEAmd_Synchronize(struct LINNEAmdContext *ctx)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LNN_OK;
}

extern "C" int LINNEAmd_SetLearning(struct LINNEAmdContext *ctx, uint32_t enable)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->learning = enable ? 1u : 0u;
    return LNN_OK;
}
extern "C" int LINNEAmd_SetAfIterations(struct LINNEAmdContext *ctx, uint32_t iterations)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->af_iters = iterations;
    return LNN_OK;
}
extern "C" int LINNEAmd_EnableTiming(struct LINNEAmdContext *ctx, int enable) { if (!ctx) return LNN_INVALID_ARGUMENT; ctx->timing = enable; return LNN_OK; }
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static void read_call_knobs(LINNEAmdContext *ctx)
{
    ctx->knob.sort = env_int("LINNE_AMD_SORT", 1);
    ctx->knob.l0_products = env_int("LINNE_AMD_L0_PRODUCTS", 1);
    ctx->knob.wide = env_int("LINNE_AMD_WIDE", 1);
    ctx->knob.search_long = env_int("LINNE_AMD_SEARCH_LONG", 1);
    ctx->knob.rows16 = env_int("LINNE_AMD_ROWS16", 1);
    ctx->knob.prep_general = env_int("LINNE_AMD_PREP_GENERAL", 0);
    ctx->knob.prep_defer = env_int("LINNE_AMD_PREP_DEFER", 1);
    ctx->knob.fwd_loss_mw = env_int("LINNE_AMD_FWD_LOSS_MW", 1);
    ctx->knob.last_layer = env_int("LINNE_AMD_LAST_LAYER", 1);
    ctx->knob.stats_rows = env_int("LINNE_AMD_STATS_ROWS", -1);
    { const char *e = getenv("LINNE_AMD_HIST"); ctx->knob.hist = e ? (atoi(e) != 0) : -1; }
    ctx->knob.rows8 = env_int("LINNE_AMD_DECODE_ROWS8", -1);
    ctx->knob.decode_fused = env_int("LINNE_AMD_DECODE_FUSED", 1);
    { const char *e = getenv("LINNE_AMD_DECODE_KERNEL"); ctx->knob.decode_kernel = !e ? 0 : (strcmp(e, "wave") == 0 ? 1 : (strcmp(e, "pipe") == 0 ? 3 : (strcmp(e, "rows") == 0 ? 4 : 2))); }
    /* LINNE_AMD_STREAMS per call: a call may use fewer compute sub-streams than the context created (bench.py times one step on
     * one stream so that its per-kernel spans do not overlap); it cannot use more */
    ctx->knob.streams = env_int("LINNE_AMD_STREAMS", 0);
    if (ctx->knob.streams < 0) ctx->knob.streams = 0;
#ifdef LNN_TIMING_EXPERIMENTS       /* builds for timing experiments only (make EXPERIMENTS=1): with these set the results are WRONG */
    ctx->knob.dbg_maxtr = (uint32_t)env_int("LINNE_AMD_DBG_MAXTR", 0);
    ctx->knob.nostats = env_int("LINNE_AMD_DBG_NOSTATS", 0);
#else
    ctx->knob.dbg_maxtr = 0; ctx->knob.nostats = 0;
#endif
}
/* span bookkeeping: span_begin/span_end bracket one kernel launch with events when timing is on */
static int span_begin(LINNEAmdContext *ctx, int kind, hipStream_t st)
{
    if (!ctx->timing) return -1;
    if (ctx->nspans == ctx->span_cap) {
        const int ncap = ctx->span_cap ? ctx->span_cap * 2 : 256;
        hipEvent_t *ne = (hipEvent_t *)realloc(ctx->span_ev, sizeof(hipEvent_t) * 2 * ncap);
        if (!ne) return -1;
        ctx->span_ev = ne;
        int *nk = (int *)realloc(ctx->span_kind, sizeof(int) * ncap);
        if (!nk) return -1;
        ctx->span_kind = nk;
        for (int i = 2 * ctx->span_cap; i < 2 * ncap; i++) if (hipEventCreate(&ctx->span_ev[i]) != hipSuccess) return -1;
        ctx->span_cap = ncap;
    }
    const int id = ctx->nspans++;
    ctx->span_kind[id] = kind;
    (void)hipEventRecord(ctx->span_ev[2 * id], st);
    return id;
}
static void span_end(LINNEAmdContext *ctx, int id, hipStream_t st) { if (id >= 0) (void)hipEventRecord(ctx->span_ev[2 * id + 1], st); }

extern "C" double LINNEAmd_GetLastTimingMs(struct LINNEAmdContext *ctx, int which)
{
    if (!ctx || which < 0) return -1.0;
    if (which == 0) {
        float ms = -1.0f;
        if (!ctx->ev_valid) return -1.0;
        if (hipEventSynchronize(ctx->ev[1]) != hipSuccess || hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) != hipSuccess) return -1.0;
        return ms;
    }
    double sum = 0.0; int cnt = 0;
    for (int i = 0; i < ctx->nspans; i++) if (ctx->span_kind[i] == which) {
        float ms = 0.0f;
        if (hipEventSynchronize(ctx->span_ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms, ctx->span_ev[2 * i], ctx->span_ev[2 * i + 1]) != hipSuccess) return -1.0;
        sum += ms; cnt++;
    }
    return cnt ? sum : -1.0;
}
extern "C" int LINNEAmd_GetLastTimingLaunches(struct LINNEAmdContext *ctx, int which)
{
    if (!ctx) return 0;
    int cnt = 0;
    for (int i = 0; i < ctx->nspans; i++) if (ctx->span_kind[i] == which) cnt++;
    return cnt;
}

static int ensure_buf(LINNEAmdContext *ctx, void **ptr, uint64_t *cap, uint64_t need)
{
    if (*cap >= need) return LNN_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (*ptr) HIPCHK(ctx, hipFree(*ptr));
    *ptr = NULL; *cap = 0;
    HIPCHK(ctx, hipMalloc(ptr, need));
    *cap = need;
    return LNN_OK;
}

struct HostShape { uint32_t L, R, P[LNN_MAXL], coef_off[LNN_MAXL], maxP; double regs[LNN_MAXR]; };
static int shape_info(const struct LINNEAmdShape *s, HostShape *h)
{
    if (!s || s->preset >= 8 || s->num_channels == 0 || s->num_channels > LNN_MAXCH || s->bits_per_sample == 0 || s->bits_per_sample > 32
            || s->num_samples_per_block == 0 || s->ch_process_method > 1 || (s->ch_process_method == 1 && s->num_channels < 2)) return LNN_INVALID_FORMAT;
    lnn_preset_info(s->preset, &h->L, h->P, &h->R, h->regs);
    uint32_t off = 0; h->maxP = 0;
    for (uint32_t l = 0; l < h->L; l++) { h->coef_off[l] = off; off += h->P[l]; if (h->P[l] > h->maxP) h->maxP = h->P[l]; }
    for (uint32_t l = 0; l < h->L; l++) if (s->num_samples_per_block <= h->P[l]) return LNN_INVALID_FORMAT;   /* linne_encoder.c:176-181 */
    return LNN_OK;
}

/* next buffer of the pinned metadata ring, holding at least 4 * F words; waits for the copy that last read it */
static int meta_acquire(LINNEAmdContext *ctx, uint32_t F, int *m_out)
{
    const int m = ctx->meta_next;
    ctx->meta_next = (m + 1) % LNN_META;
    if (ctx->meta_used[m]) { HIPCHK(ctx, hipEventSynchronize(ctx->meta_ev[m])); ctx->meta_used[m] = 0; }
    if (!ctx->meta_ev[m]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->meta_ev[m], hipEventDisableTiming));
    if (ctx->meta_cap[m] < 4ull * F) {
        const uint64_t cap = 4ull * (F < 4096u ? 4096u : F);
        if (ctx->meta_h[m]) { HIPCHK(ctx, hipHostFree(ctx->meta_h[m])); ctx->meta_h[m] = NULL; ctx->meta_cap[m] = 0; }
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->meta_h[m], sizeof(uint32_t) * cap, hipHostMallocDefault));
        ctx->meta_cap[m] = cap;
    }
    *m_out = m;
    return LNN_OK;
}

/* frame lengths of a decode call: the synthesis kernels need nothing but each frame's length (any number of distinct
 * lengths: a stream written by EncodeBlock calls of varying num_samples, linne_decoder.c:671-742), so no class tables
 * are built and the encode side's resident tables stay valid */
static int upload_lengths(LINNEAmdContext *ctx, const struct LINNEAmdShape *shape, const uint32_t *h_num_samples, uint32_t F)
{
    const uint32_t S = shape->num_samples_per_block;
    int m, ret;
    if ((ret = meta_acquire(ctx, F, &m)) != LNN_OK) return ret;
    uint32_t *nsm = ctx->meta_h[m];
    for (uint32_t f = 0; f < F; f++) {
        const uint32_t n = h_num_samples ? h_num_samples[f] : S;
        if (n == 0 || n > S) { snprintf(ctx->err, sizeof(ctx->err), "frame %u: num_samples %u out of range", f, n); return LNN_INVALID_ARGUMENT; }
        nsm[f] = n;
    }
    if ((ret = ensure_buf(ctx, (void **)&ctx->d_nsmp, &ctx->nsmp_cap, sizeof(uint32_t) * (uint64_t)(F ? F : 1))) != LNN_OK) return ret;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_nsmp, nsm, sizeof(uint32_t) * F, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->meta_ev[m], ctx->stream));
    ctx->meta_used[m] = 1;
    return LNN_OK;
}

/* The libm values of the path, behind names of their own so that a test can compare the box's libm with the committed
 * values of the build container (tests/golden/libm_values.json): were they ever to differ, the test names the cause where the
 * parity tests would only show hashes that do not match. */
extern "C" double lnn_welch_divisor(uint32_t unit_samples) { return 4.0 * pow((double)(unit_samples - 1u), -2.0); }                        /* lpc.c:199 */
extern "C" double lnn_sin_window(uint32_t s, uint32_t n) { return sin((3.1415926535897932384626433832795029 * s) / (n - 1)); }            /* lpc.c:192 */
extern "C" double lnn_cholesky_pivot(double sum) { return pow(sum, -0.5); }                                                                /* lpc.c:421 */

/* one length class: analysis length, the unit counts each layer may try, their Welch divisors (host libm, lpc.c:199) and
 * the offsets of its tables; returns LNN_INVALID_FORMAT for a length the device path does not take */
static int make_class(LINNEAmdContext *ctx, const HostShape *hs, uint32_t S, uint32_t n, uint64_t *sin_total, uint64_t *wt_total, DevClass *out)
{
    DevClass c; memset(&c, 0, sizeof(c));
    c.n = n;
    uint32_t na = ((n + 7u) / 8u) * 8u;             /* linne_encoder.c:652-654 */
    if (na < hs->maxP) na = hs->maxP;
    if (na > S) na = S;
    c.na = na;
    c.sin_off = (uint32_t)*sin_total; *sin_total += n;
    for (uint32_t l = 0; l < hs->L; l++) {
        const uint32_t maxu = hs->P[l] < 128u ? hs->P[l] : 128u;    /* linne_network.c:586,594 */
        uint32_t nt = 0;
        for (uint32_t u = 1; u <= maxu; u <<= 1) {
            if ((hs->P[l] % u) != 0 || (na % u) != 0) continue;      /* linne_network.c:291-294 */
            c.trial_u[l][nt] = u;
            c.trial_div[l][nt] = lnn_welch_divisor(na / u);               /* lpc.c:199 */
            c.wt_off[l][nt] = (uint32_t)*wt_total;
            { const uint32_t pu = hs->P[l] / u; *wt_total += na / u + (pu > 4 ? pu : 4); *wt_total = (*wt_total + 3u) & ~(uint64_t)3u; }   /* tables start 32-byte aligned */
            nt++;
        }
        c.ntrials[l] = nt;
    }
    *out = c;
    return LNN_OK;
}

/* Builds the per-length classes of an encode batch (tables are host libm values) and the class-sorted
 * order the kernels work in: sorted row i is the caller's frame map[i]; rows of one class are contiguous (stable: the
 * caller's order inside a class), so that whatever the order of lengths in the batch -- many tracks back to back, each
 * with its ragged tail -- a chunk has at most one run per class and the lanes = rows kernels see class-homogeneous
 * blocks.  The class tables are CUMULATIVE: a length seen in an earlier call of the same shape keeps its slot, so a caller
 * that alternates between batches with and without a ragged tail (a pipelined stream, chunk after chunk) uploads tables
 * once per new length and never again; only then does the call synchronise the host.  The per-frame class index and the
 * map go through a pinned ring. */
static int build_classes(LINNEAmdContext *ctx, const struct LINNEAmdShape *shape, const HostShape *hs,
        const uint32_t *h_num_samples, uint32_t F)
{
    uint32_t count[LNN_MAXCLS + 1];
    uint32_t lens[LNN_MAXCLS], slot_of[LNN_MAXCLS], nlen = 0;
    const uint32_t S = shape->num_samples_per_block;
    int m, ret;
    { const int r_ = meta_acquire(ctx, F, &m); if (r_ != LNN_OK) return r_; }
    uint32_t *idx = ctx->meta_h[m], *map = ctx->meta_h[m] + F, *raw = ctx->meta_h[m] + 2 * (size_t)F;
    ctx->cur_idx = idx;
    /* pass 1: the distinct lengths of this call (raw[f] = index into lens[]) */
    {
        uint32_t last_n = 0, last_k = 0;
        for (uint32_t f = 0; f < F; f++) {
            const uint32_t n = h_num_samples ? h_num_samples[f] : S;
            if (n == 0 || n > S) { snprintf(ctx->err, sizeof(ctx->err), "frame %u: num_samples %u out of range", f, n); return LNN_INVALID_ARGUMENT; }
            uint32_t k = last_k;
            if (n != last_n || nlen == 0) {
                for (k = 0; k < nlen; k++) if (lens[k] == n) break;
                if (k == nlen) {
                    if (nlen == LNN_MAXCLS) { snprintf(ctx->err, sizeof(ctx->err), "more than %d distinct frame lengths in one batch", LNN_MAXCLS); return LNN_INVALID_ARGUMENT; }
                    lens[nlen++] = n;
                }
                last_n = n; last_k = k;
            }
            raw[f] = k;
        }
    }
    /* the resident table: keep it if it has (room for) every length of this call, else start over with this call's lengths */
    const bool same_shape = ctx->sig_valid && memcmp(&ctx->sig_shape, shape, sizeof(*shape)) == 0;
    uint32_t missing = 0;
    for (uint32_t k = 0; k < nlen; k++) {
        uint32_t j = 0;
        if (same_shape) for (; j < ctx->sig_ncls; j++) if (ctx->sig_cls[j].n == lens[k]) break;
        slot_of[k] = (same_shape && j < ctx->sig_ncls) ? j : 0xFFFFFFFFu;
        if (slot_of[k] == 0xFFFFFFFFu) missing++;
    }
    if (missing) {
        if (!same_shape || ctx->sig_ncls + missing > LNN_MAXCLS) {
            ctx->sig_valid = 0; ctx->sig_ncls = 0; ctx->sig_sin_total = 0; ctx->sig_wt_total = 0;
            memset(ctx->sig_cls, 0, sizeof(ctx->sig_cls));
            for (uint32_t k = 0; k < nlen; k++) slot_of[k] = 0xFFFFFFFFu;
        }
        for (uint32_t k = 0; k < nlen; k++) if (slot_of[k] == 0xFFFFFFFFu) {
            if ((ret = make_class(ctx, hs, S, lens[k], &ctx->sig_sin_total, &ctx->sig_wt_total, &ctx->sig_cls[ctx->sig_ncls])) != LNN_OK) { ctx->sig_valid = 0; ctx->sig_ncls = 0; return ret; }
            slot_of[k] = ctx->sig_ncls++;
        }
        /* (re)build and upload the tables of every resident class */
        const uint64_t sin_total = ctx->sig_sin_total, wt_total = ctx->sig_wt_total;
        double *tab = (double *)malloc(sizeof(double) * (sin_total ? sin_total : 1));
        double *wt = (double *)calloc(wt_total ? wt_total : 1, sizeof(double));
        if (!tab || !wt) { free(tab); free(wt); ctx->sig_valid = 0; ctx->sig_ncls = 0; snprintf(ctx->err, sizeof(ctx->err), "out of host memory"); return LNN_NG; }
        for (uint32_t k = 0; k < ctx->sig_ncls; k++) {
            const DevClass &c = ctx->sig_cls[k];
            const uint32_t n = c.n;
            for (uint32_t s = 0; s < n; s++) tab[c.sin_off + s] = lnn_sin_window(s, n);   /* lpc.c:192 */
            /* Welch weights per trial over one padded unit (lpc.c:199-204): w[loc] = (div * h) * (n-1-h), h = min(loc, n-1-loc);
             * zero in the zero zone; the (never written) middle of an odd unit is handled on the device (Q1) */
            for (uint32_t l = 0; l < hs->L; l++)
                for (uint32_t t = 0; t < c.ntrials[l]; t++) {
                    const uint32_t u = c.trial_u[l][t], nu = c.na / u;
                    const double div = c.trial_div[l][t];
                    double *w = wt + c.wt_off[l][t];
                    for (uint32_t loc = 0; loc < nu; loc++) {
                        const uint32_t h = (loc < (nu >> 1)) ? loc : (nu - 1 - loc);
                        w[loc] = div * (double)h * (double)(nu - 1 - h);
                    }
                }
        }
        /* the old tables may still be read by work enqueued on the sub-streams / side stream of an earlier call */
        hipError_t e = hipDeviceSynchronize();
        ret = (e == hipSuccess) ? ensure_buf(ctx, (void **)&ctx->d_sin, &ctx->sin_cap, sizeof(double) * (sin_total ? sin_total : 1)) : LNN_NG;
        if (ret == LNN_OK) ret = ensure_buf(ctx, (void **)&ctx->d_wt, &ctx->wt_cap, sizeof(double) * (wt_total ? wt_total : 1));
        if (ret != LNN_OK) { free(tab); free(wt); ctx->sig_valid = 0; ctx->sig_ncls = 0; return ret; }
        e = hipMemcpyAsync(ctx->d_sin, tab, sizeof(double) * sin_total, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_wt, wt, sizeof(double) * wt_total, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_cls, ctx->sig_cls, sizeof(DevClass) * LNN_MAXCLS, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          /* tab / wt are freed here */
        free(tab); free(wt);
        if (e != hipSuccess) { ctx->sig_valid = 0; ctx->sig_ncls = 0; snprintf(ctx->err, sizeof(ctx->err), "class table upload: %s", hipGetErrorString(e)); return LNN_NG; }
        ctx->sig_shape = *shape; ctx->sig_valid = 1;
    }
    /* pass 2: class slots in order of first appearance in this call, stable counting sort by class
     * (LINNE_AMD_SORT=0: the caller's order, for tests of the mixed-run fallback) */
    memset(count, 0, sizeof(count));
    ctx->na_max = 0;
    for (uint32_t k = 0; k < nlen; k++) if (ctx->sig_cls[slot_of[k]].na > ctx->na_max) ctx->na_max = ctx->sig_cls[slot_of[k]].na;
    {
        if (ctx->knob.sort == 0) { for (uint32_t f = 0; f < F; f++) { idx[f] = slot_of[raw[f]]; map[f] = f; } }
        else {
            for (uint32_t f = 0; f < F; f++) count[raw[f] + 1]++;
            for (uint32_t k = 0; k < nlen; k++) count[k + 1] += count[k];
            for (uint32_t f = 0; f < F; f++) { const uint32_t pos = count[raw[f]]++; idx[pos] = slot_of[raw[f]]; map[pos] = f; }
        }
    }
    {       /* short layers by products (k_autocorr_prod): all trials present and every unit length even, in every class of this call */
        ctx->prod_ok = 0;
        for (uint32_t l = 0; l < hs->L; l++) {
            if (hs->P[l] > 16u || !ctx->knob.l0_products) continue;
            uint32_t nt = 0; for (uint32_t u = 1; u <= hs->P[l]; u <<= 1) nt++;
            int ok = 1;
            for (uint32_t k = 0; k < nlen; k++) { const DevClass &c = ctx->sig_cls[slot_of[k]]; if (c.ntrials[l] != nt || (c.na % (1u << nt)) != 0) ok = 0; }
            if (ok) ctx->prod_ok |= 1 << l;
        }
        /* long layers by lanes = lags (k_autocorr_wide): a small batch, and every unit length of every trial a class has even */
        for (uint32_t l = 0; l < hs->L; l++) {
            if (hs->P[l] < 32u || hs->P[l] > 128u || !ctx->knob.wide || (uint64_t)F * shape->num_channels * (l == 0 ? 1u : hs->R) > 64u) continue;
            int ok = 1;
            for (uint32_t k = 0; k < nlen; k++) { const DevClass &c = ctx->sig_cls[slot_of[k]]; if (c.ntrials[l] == 0 || (c.na % (1u << c.ntrials[l])) != 0) ok = 0; }
            if (ok) ctx->prod_ok |= 1 << l;
        }
    }
    if ((ret = ensure_buf(ctx, (void **)&ctx->d_clsidx, &ctx->clsidx_cap, sizeof(uint32_t) * 2 * (uint64_t)(F ? F : 1))) != LNN_OK) return ret;
    ctx->d_map = ctx->d_clsidx + F;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_clsidx, idx, sizeof(uint32_t) * 2 * (size_t)F, hipMemcpyHostToDevice, ctx->stream));    /* class index and map, back to back */
    HIPCHK(ctx, hipEventRecord(ctx->meta_ev[m], ctx->stream));
    ctx->meta_used[m] = 1;
    return LNN_OK;
}

static uint64_t align_up(uint64_t v) { return (v + 255u) & ~(uint64_t)255u; }

/* RowRuns of a chunk of frames whose rows are `rpf` per frame (see lnn_dev_common.h) */
static void build_runs(RowRuns *rr, const uint32_t *idx, uint32_t F, uint32_t rpf)
{
    uint32_t n = 0, f = 0;
    rr->row_begin[0] = 0; rr->blk_begin[0] = 0;
    while (f < F) {
        uint32_t g = f + 1;
        while (g < F && idx[g] == idx[f]) g++;
        if (n == LNN_MAXRUN) { n = 0; break; }                  /* too many runs: one run over everything */
        rr->row_begin[n + 1] = g * rpf;
        rr->blk_begin[n + 1] = rr->blk_begin[n] + ((g - f) * rpf + 63u) / 64u;
        n++; f = g;
    }
    rr->mixed = 0;
    if (n == 0) { n = 1; rr->mixed = 1; rr->row_begin[1] = F * rpf; rr->blk_begin[1] = (F * rpf + 63u) / 64u; }
    rr->n = n;
}

/* bytes of scratch one frame needs (C channel-frames, R passes each) */
static uint64_t frame_scratch_bytes(const struct LINNEAmdShape *shape, const HostShape *hs, uint32_t af_iters = 0, uint32_t learning = 0)
{
    const uint64_t C = shape->num_channels, S = shape->num_samples_per_block, J = C * hs->R;
    uint64_t b = 0;
    b += 2 * C * S * sizeof(int32_t);
    b += J * 2 * S * sizeof(double);
    b += J * LNN_MAXT * LNN_ACW * sizeof(double);
    b += J * LNN_MAXT * LNN_MAXP * sizeof(double);
    b += J * LNN_MAXT * LNN_MAXU * (sizeof(double) + 1);
    b += J * LNN_MAXT * sizeof(double) * (1 + (uint64_t)((S + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64)) + J;
    b += J * sizeof(double) * (uint64_t)((S + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64) + 256;
    b += J * LNN_MAXT * sizeof(double) + 256;
    b += J * LNN_MAXL * LNN_MAXP * sizeof(double);
    b += J * LNN_MAXL * sizeof(uint32_t);
    b += J * 2 * sizeof(double);
    b += C * sizeof(uint32_t) + 512;                        /* k_prep_slow's list */
    if (af_iters)       /* the auxiliary-function pass: per channel-frame the normal matrices, reciprocals, vectors, problem lists */
        b += C * (sizeof(double) * ((uint64_t)hs->maxP * hs->maxP + S + 3 * LNN_MAXP + 3 * LNN_MAXU + 2) + sizeof(uint32_t) * (2 * LNN_MAXU + 1)) + 8192;
    if (af_iters || learning) b += C * (sizeof(uint32_t) + 2 * sizeof(double)) + 1024;        /* the winners of the search passes */
    if (learning)       /* the trainer: two layer inputs and a gradient-signal buffer per layer (TR_NBUF), gradients and momenta per channel-frame */
        b += C * (sizeof(double) * ((2 + LNN_MAXL) * S + 2 * LNN_MAXL * LNN_MAXP + 2) + sizeof(uint32_t)) + 4096;
    return b + 4096;
}

extern "C" uint64_t LINNEAmd_ScratchBytesPerFrame(const struct LINNEAmdShape *shape)
{
    HostShape hs;
    if (shape_info(shape, &hs) != LNN_OK) return 0;
    return frame_scratch_bytes(shape, &hs);
}

/* k_fir2 launcher: layer 0 reads the int32 channel (L0); `spec` = the search also writes the one-unit trial's forward output
 * (MODE 2) / the forward pass skips the jobs that chose one unit (MODE 1) */
template <int MODE> static void launch_fir(hipStream_t st, const Plan &p, uint32_t l, uint32_t cur, uint32_t J, uint32_t tiles, bool spec)
{
    const dim3 grid(J, (MODE == 1 && spec && J >= 4096u) ? 1u : tiles), blk(FIR_THREADS);      /* forward pass in batches: a block per job walks the tiles (few jobs have any work) */
    if (l == 0) { if (spec) hipLaunchKernelGGL((k_fir2<MODE, true, true>), grid, blk, 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir2<MODE, true, false>), grid, blk, 0, st, p, l, cur); }
    else        { if (spec) hipLaunchKernelGGL((k_fir2<MODE, false, true>), grid, blk, 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir2<MODE, false, false>), grid, blk, 0, st, p, l, cur); }
}

/* search of a short layer (P <= 16): register-window kernel */
static void launch_fir_small_search(hipStream_t st, const Plan &p, uint32_t l, uint32_t cur, uint32_t J, uint32_t tiles, bool spec, uint32_t P)
{
    const dim3 grid(l == 0 ? J / p.R : J, tiles), blk(FIR_THREADS);      /* layer 0: one block serves the R jobs of a channel-frame */
#define LNN_FS(PP) do { \
        if (l == 0) { if (spec) hipLaunchKernelGGL((k_fir_small<PP, true, true>), grid, blk, 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir_small<PP, true, false>), grid, blk, 0, st, p, l, cur); } \
        else        { if (spec) hipLaunchKernelGGL((k_fir_small<PP, false, true>), grid, blk, 0, st, p, l, cur); else hipLaunchKernelGGL((k_fir_small<PP, false, false>), grid, blk, 0, st, p, l, cur); } } while (0)
    switch (P) { case 2: LNN_FS(2); break; case 4: LNN_FS(4); break; case 8: LNN_FS(8); break; default: LNN_FS(16); break; }
#undef LNN_FS
}

extern "C" int LINNEAmd_EncodeFramesDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_pcm, const uint32_t *h_num_samples, uint32_t num_frames,
        int32_t *d_residual, int32_t *d_params, double *d_stats)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    const uint32_t pcm16 = (uint32_t)ctx->pcm16_next;        /* 0 int32, 1 int16, 2 packed 3-byte samples (set by the staging slots) */
    ctx->pcm16_next = 0;
    if (!shape || !d_pcm || !d_residual || !d_params || !d_stats) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HostShape hs;
    int ret = shape_info(shape, &hs);
    if (ret != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((ret = ctx_encode_streams(ctx)) != LNN_OK) return ret;
    read_call_knobs(ctx);
    if ((ret = build_classes(ctx, shape, &hs, h_num_samples, num_frames)) != LNN_OK) return ret;

    const uint32_t C = shape->num_channels, S = shape->num_samples_per_block;
    const uint64_t per_frame = frame_scratch_bytes(shape, &hs, ctx->af_iters, ctx->learning);
    if (ctx->arena_bytes < per_frame * 4 + 65536) {       /* grow the arena to hold at least a few frames */
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(ctx->arena)); ctx->arena = NULL; ctx->arena_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->arena, per_frame * 4 + 65536));
        ctx->arena_bytes = per_frame * 4 + 65536;
    }
    /* frame groups ("chunks") rotate over nsub streams, each with its own slice of the arena */
    uint32_t nsub = ctx->nsub > 0 ? (uint32_t)ctx->nsub : 1u;
    const bool streams_forced = ctx->nsub_forced || ctx->knob.streams > 0;
    if (ctx->knob.streams > 0 && (uint32_t)ctx->knob.streams < nsub) nsub = (uint32_t)ctx->knob.streams;
    while (nsub > 1 && ((ctx->arena_bytes - 65536) / nsub < per_frame * 2 || num_frames < nsub * 512u)) nsub--;
    /* By default a call is cut in two only if each half still fills the chip and keeps every large-batch kernel form (the rules below
     * go by the jobs of a chunk: k_fwd_loss from 24 576): the halves' latency-bound kernels (Levinson-Durbin, the short layers' search,
     * the selections) then run beside the other half's vector-unit-bound ones -- 83.1 -> 80.1 ms per step on the 60-minute batch
     * (tools/streams_ab.sh).  Smaller batches keep one stream and the context's own (no fork / join around a block-at-a-time call). */
    if (!streams_forced) while (nsub > 1 && (uint64_t)(num_frames / nsub) * C * hs.R < 32768u) nsub--;
    const uint64_t part_bytes = ((ctx->arena_bytes - 65536) / nsub) & ~(uint64_t)255;
    uint64_t chunk = part_bytes / per_frame;
    if (chunk == 0) chunk = 1;
    {   /* even split over the streams; every kernel carries the job index in grid.x: J = chunk * C * R is kept below 2^22 */
        const uint64_t even = (num_frames + nsub - 1) / nsub;
        if (chunk > even) chunk = even;
        const uint64_t lim = 4194304u / ((uint64_t)C * hs.R);
        if (chunk > lim) chunk = lim;
        /* chunks of equal size (a multiple of the stream count): a small last chunk would run the latency-bound kernels
         * nearly empty */
        uint64_t nchunks = (num_frames + chunk - 1) / chunk;
        nchunks = ((nchunks + nsub - 1) / nsub) * nsub;
        chunk = (num_frames + nchunks - 1) / nchunks;
    }
    ctx->nspans = 0;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ucount, 0, sizeof(uint32_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ucount + 2, 0x7F, 2 * sizeof(uint32_t), ctx->stream));      /* min margin: a huge double (0x7F7F...) */
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream)); }
    const bool use_sub = ctx->nsub > 0 && (nsub > 1 || (streams_forced && ctx->knob.streams != 1));
    if (use_sub || ctx->has_side) HIPCHK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    if (use_sub) for (uint32_t i = 0; i < nsub; i++) HIPCHK(ctx, hipStreamWaitEvent(ctx->sub[i], ctx->ev_start, 0));
    {   /* statistics of every frame of the call: one launch beside the analysis */
        Plan ps; memset(&ps, 0, sizeof(ps));
        ps.C = C; ps.S = S; ps.bits = shape->bits_per_sample; ps.L = hs.L; ps.R = hs.R; ps.F = num_frames;
        for (uint32_t l = 0; l < hs.L; l++) ps.P[l] = hs.P[l];
        ps.scale = ldexp(1.0, -(int)(shape->bits_per_sample - 1));
        ps.pcm = d_pcm; ps.pcm16 = pcm16; ps.stats = d_stats; ps.cls_of_frame = ctx->d_clsidx; ps.frame_map = ctx->d_map; ps.cls = ctx->d_cls; ps.sintab = ctx->d_sin;
        hipStream_t ss = ctx->stream;
        if (ctx->has_side) { ss = ctx->side; HIPCHK(ctx, hipStreamWaitEvent(ss, ctx->ev_start, 0)); }
        if (!ctx->knob.nostats) {      /* (always, except in a build for timing experiments: without the statistics the block types are wrong) */
        const int sp_ = span_begin(ctx, 13, ss);
        /* batches: lanes = channel-frames (k_stats_rows); a few channel-frames: a block each (k_stats finishes one block sooner).  LINNE_AMD_STATS_ROWS forces either */
        const bool rows_form = ctx->knob.stats_rows >= 0 ? (ctx->knob.stats_rows != 0) : ((uint64_t)num_frames * C >= 1024u);
        if (rows_form && (S & 3u) == 0 && (hs.P[0] == 2u || hs.P[0] == 4u)) {
            const dim3 g((num_frames * C + 63u) / 64u);
            if (hs.P[0] == 4u) { if (pcm16 == 1u) hipLaunchKernelGGL((k_stats_rows<5, 1>), g, dim3(320), 0, ss, ps); else if (pcm16 == 2u) hipLaunchKernelGGL((k_stats_rows<5, 2>), g, dim3(320), 0, ss, ps); else hipLaunchKernelGGL((k_stats_rows<5, 0>), g, dim3(320), 0, ss, ps); }
            else               { if (pcm16 == 1u) hipLaunchKernelGGL((k_stats_rows<3, 1>), g, dim3(192), 0, ss, ps); else if (pcm16 == 2u) hipLaunchKernelGGL((k_stats_rows<3, 2>), g, dim3(192), 0, ss, ps); else hipLaunchKernelGGL((k_stats_rows<3, 0>), g, dim3(192), 0, ss, ps); }
        }
        else hipLaunchKernelGGL(k_stats, dim3(num_frames, C), dim3(STAT_THREADS), 0, ss, ps);
        span_end(ctx, sp_, ss); }
        if (ss != ctx->stream) HIPCHK(ctx, hipEventRecord(ctx->side_done, ss));
    }
    /* the chunks are enqueued inside one function so that EVERY way out of the loop -- a failing HIP call, a failing layer pass --
     * comes by the join below: with sub-streams forked off, the caller may only reuse its buffers once they have drained */
    auto enqueue_chunks = [&]() -> int {
    uint32_t chunk_index = 0;
    for (uint32_t f0 = 0; f0 < num_frames; f0 += (uint32_t)chunk, chunk_index++) {
        const uint32_t slot = chunk_index % nsub;
        hipStream_t st = use_sub ? ctx->sub[slot] : ctx->stream;
        const uint32_t Fc = (num_frames - f0 < chunk) ? (num_frames - f0) : (uint32_t)chunk;
        const uint64_t CF = (uint64_t)Fc * C, J = CF * hs.R;
        Plan p; memset(&p, 0, sizeof(p));
        p.C = C; p.S = S; p.bits = shape->bits_per_sample; p.L = hs.L; p.R = hs.R; p.ms = shape->ch_process_method; p.F = Fc; p.J = (uint32_t)J;
        for (uint32_t l = 0; l < hs.L; l++) { p.P[l] = hs.P[l]; p.coef_off[l] = hs.coef_off[l]; }
        for (uint32_t r = 0; r < hs.R; r++) p.regs[r] = hs.regs[r];
        p.scale = ldexp(1.0, -(int)(shape->bits_per_sample - 1));
        p.pcm = d_pcm; p.pcm16 = pcm16; p.resid = d_residual; p.prm = d_params; p.stats = d_stats;      /* the caller's arrays: rows of the class-sorted chunk reach them through frame_map */
        /* last layer: forward pass + loss in one kernel for the jobs it takes (fwd_loss_takes); the two-kernel form runs only
         * when the chunk holds frames it does not take */
        const uint32_t Plast = hs.P[hs.L - 1];
        /* The lanes = jobs kernels need a batch that fills the chip with 64-job waves: below ~24 k jobs (k_fwd_loss: one wave per
         * 64 jobs) / ~12 k jobs (k_autocorr_hist: one block per 64 jobs and trial) the block-per-job kernels finish sooner --
         * a single stereo frame takes 2.6 ms with them, 6.2 ms without this rule.  The environment forces either form. */
        const bool fwd_loss_on = (ctx->fwd_loss < 0) ? (J >= 24576u) : (ctx->fwd_loss != 0);
        const bool fuse_cfg = fwd_loss_on && hs.L > 1 && (Plast == 2u || Plast == 4u || Plast == 8u || Plast == 16u);
        bool fuse_all = fuse_cfg;
        for (uint32_t f = f0; f < f0 + Fc && fuse_all; f++) if ((ctx->sig_cls[ctx->cur_idx[f]].na % (4u * Plast)) != 0) fuse_all = false;
        p.fused_last = fuse_cfg ? 1u : 0u;
        /* k_last_layer (search + forward pass + loss of the last layer in one launch) takes a chunk whole or not at all: every frame
         * k_fwd_loss's and with every trial (LINNE_AMD_EXACT keeps the certified search's exact fallback in use: that knob compares the two) */
        /* (from 49 152 jobs on, or when LINNE_AMD_LAST_LAYER=2 says always: with lanes = jobs and a wave per 64 of them a chunk of J jobs is
         * J / 64 waves on 1024 SIMDs, and a lone wave walks its frames in 4.1 ms however few they are -- the 31 008 jobs of a group of
         * EncodeWhole took 4.1 ms here and 1.8 in the three kernels: 108 -> 110.5 ms per 60-minute stream) */
        bool last_layer_all = fuse_all && ctx->knob.last_layer && !ctx->force_exact && !ctx->af_iters && !ctx->learning && (J >= (use_sub ? 49152u : 81920u) || (ctx->knob.last_layer == 2 && J > 256u));      /* (a chunk alone on the GPU has nothing beside its lone waves: it pays from ~78 k jobs on -- 124 k x 4.1 / 6.6) */
        if (last_layer_all) {
            uint32_t nt = 0; for (uint32_t u = 1; u <= Plast && u <= (uint32_t)LNN_MAXU; u <<= 1) nt++;
            for (uint32_t f = f0; f < f0 + Fc && last_layer_all; f++) if (ctx->sig_cls[ctx->cur_idx[f]].ntrials[hs.L - 1] != nt) last_layer_all = false;
        }
        p.search_long = ctx->knob.search_long ? 1u : 0u;
        p.rows16 = ctx->knob.rows16 ? 1u : 0u;
        p.prep_general = ctx->knob.prep_general ? 1u : 0u;
        p.prep_defer = (ctx->knob.prep_defer && !ctx->knob.prep_general && (S & 3u) == 0 && CF * S * sizeof(int32_t) <= 0xFFFFFFFFull) ? 1u : 0u;      /* (k_prep_slow addresses xtmp with 32-bit byte offsets) */
        build_runs(&p.runs[0], ctx->cur_idx + f0, Fc, C); build_runs(&p.runs[1], ctx->cur_idx + f0, Fc, C * hs.R);
        p.hist = (ctx->knob.hist >= 0 ? (ctx->knob.hist != 0) : (J >= 12288u)) ? 1u : 0u;
        if (p.runs[1].mixed) p.hist = 0;                        /* more class runs than RowRuns holds: blocks may mix classes, which only the general kernels serve */
        bool hist_all[LNN_MAXL];                                /* per layer: every frame of the chunk is k_autocorr_hist's (host copy of hist_takes) */
        for (uint32_t l = 0; l < hs.L; l++) {
            uint32_t nt = 0; for (uint32_t u = 1; u <= hs.P[l] && u <= (uint32_t)LNN_MAXU; u <<= 1) nt++;
            hist_all[l] = p.hist && hs.P[l] >= 64u && (S & 3u) == 0;
            for (uint32_t f = f0; f < f0 + Fc && hist_all[l]; f++) {
                const DevClass &c = ctx->sig_cls[ctx->cur_idx[f]];
                if (!(c.ntrials[l] == nt && (c.na % (16u << (nt - 1))) == 0 && (c.na >> (nt - 1)) >= 32u)) hist_all[l] = false;
            }
        }
        p.cls_of_frame = ctx->d_clsidx + f0; p.frame_map = ctx->d_map + f0; p.cls = ctx->d_cls; p.sintab = ctx->d_sin; p.wtab = ctx->d_wt; p.ucount = ctx->d_ucount; p.min_margin = (unsigned long long *)(ctx->d_ucount + 2); p.force_exact = ctx->force_exact ? 1u : 0u; p.dbg_maxtr = ctx->knob.dbg_maxtr;
        uint8_t *const abase = (uint8_t *)ctx->arena + (size_t)slot * part_bytes;
        uint8_t *a = abase;
#define TAKE(ptr, type, count) do { ptr = (type *)a; a += align_up(sizeof(type) * (uint64_t)(count)); } while (0)
        TAKE(p.xint, int32_t, CF * S); TAKE(p.xtmp, int32_t, CF * S);
        TAKE(p.sig, double, J * 2 * S);
        TAKE(p.acorr, double, J * LNN_MAXT * LNN_ACW); TAKE(p.tcoef, double, J * LNN_MAXT * LNN_MAXP);
        TAKE(p.ptail, double, J * LNN_MAXT * LNN_MAXU); TAKE(p.ptail_set, uint8_t, J * LNN_MAXT * LNN_MAXU);
        TAKE(p.tloss, double, J * LNN_MAXT); p.npart = ((S + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64); TAKE(p.tsum, double, J * LNN_MAXT * p.npart); TAKE(p.txmax, double, J * p.npart); TAKE(p.thsum, double, J * LNN_MAXT); TAKE(p.uncertain, uint8_t, J); TAKE(p.lparams, double, J * LNN_MAXL * LNN_MAXP);
        TAKE(p.lunits, uint32_t, J * LNN_MAXL); TAKE(p.jloss, double, J); TAKE(p.jtail, double, J);
        TAKE(p.prep_slow_n, uint32_t, 64); TAKE(p.prep_slow_rows, uint32_t, CF);
        uint32_t *af_best = NULL; double *af_loss = NULL, *af_reg = NULL;
        TrainArgs tr; memset(&tr, 0, sizeof(tr));
        if (ctx->af_iters || ctx->learning) { TAKE(af_best, uint32_t, CF); TAKE(af_loss, double, CF); TAKE(af_reg, double, CF); }
        if (ctx->learning) {
            TAKE(tr.buf, double, CF * TR_NBUF * S); TAKE(tr.dparams, double, CF * LNN_MAXL * LNN_MAXP); TAKE(tr.momentum, double, CF * LNN_MAXL * LNN_MAXP);
            TAKE(tr.loss, double, CF); TAKE(tr.prev, double, CF); TAKE(tr.active, uint32_t, CF); TAKE(tr.nactive, uint32_t, 64);
        }
        if (ctx->af_iters) {      /* the final pass works on CF jobs */
            TAKE(p.af_a, double, CF * LNN_MAXP); TAKE(p.af_inv, double, CF * S); TAKE(p.af_R, double, CF * hs.maxP * hs.maxP);
            TAKE(p.af_rv, double, CF * LNN_MAXP); TAKE(p.af_invd, double, CF * LNN_MAXP);
            TAKE(p.af_obj, double, CF * LNN_MAXU); TAKE(p.af_prev, double, CF * LNN_MAXU); TAKE(p.af_state, uint32_t, CF * LNN_MAXU);
            TAKE(p.af_prob, uint32_t, CF * LNN_MAXU); TAKE(p.af_nprob, uint32_t, 64); TAKE(p.af_pivot, double, CF * LNN_MAXU);
        }
#undef TAKE
        if ((uint64_t)(a - abase) > part_bytes) { snprintf(ctx->err, sizeof(ctx->err), "internal: arena overflow"); return LNN_NG; }
        const uint32_t sblocks = (S + 255) / 256;
        if (p.prep_defer) HIPCHK(ctx, hipMemsetAsync(p.prep_slow_n, 0, sizeof(uint32_t), st));
        {   /* k_prep, and behind it k_prep_slow for the channel-frames it listed (none for 16-bit material: its blocks leave at once) */
            const int sp_ = span_begin(ctx, 1, st);
            hipLaunchKernelGGL(k_prep, dim3(Fc, C), dim3(PREP_THREADS), 0, st, p);
            if (p.prep_defer) hipLaunchKernelGGL(k_prep_slow, dim3((uint32_t)((CF + 63) / 64)), dim3(256), 0, st, p);
            span_end(ctx, sp_, st);
        }
        /* -a N: the auxiliary-function iterations on the coefficients k_select kept for layer l (lnn_k_af.h); synchronous: every
         * Cholesky pivot goes through the host's pow() */
        auto run_af = [&](const Plan &q, uint64_t Jq, uint32_t l, uint32_t cur, uint32_t iters) -> int {
            const uint32_t P = hs.P[l];
            HIPCHK(ctx, hipMemsetAsync(q.af_nprob, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL(k_af_init, dim3(((uint32_t)Jq + 255) / 256), dim3(256), 0, st, q, l);
            uint32_t nprob = 0;
            HIPCHK(ctx, hipMemcpyAsync(&nprob, q.af_nprob, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            if (nprob == 0) return LNN_OK;
            if (ctx->af_h_cap < nprob) {
                if (ctx->af_h) HIPCHK(ctx, hipHostFree(ctx->af_h));
                ctx->af_h = NULL; ctx->af_h_cap = 0;
                HIPCHK(ctx, hipHostMalloc((void **)&ctx->af_h, sizeof(double) * (size_t)nprob, hipHostMallocDefault));
                ctx->af_h_cap = nprob;
            }
            uint32_t mblocks = 0;                                                      /* k_af_matrix: blocks per job, whatever unit count it chose */
            for (uint32_t uu = 1; uu <= P; uu <<= 1) { const uint32_t b_ = uu * afm_blocks_per_unit(P / uu); if (b_ > mblocks) mblocks = b_; }
            for (uint32_t it = 0; it < iters; it++) {
                hipLaunchKernelGGL(k_af_resid, dim3((uint32_t)Jq, (S + AFR_THREADS * 4 - 1) / (AFR_THREADS * 4)), dim3(AFR_THREADS), 0, st, q, l, cur);
                hipLaunchKernelGGL(k_af_obj, dim3(nprob), dim3(64), 0, st, q, l, cur);
                hipLaunchKernelGGL(k_af_matrix, dim3((uint32_t)Jq, mblocks), dim3(AFM_THREADS), 0, st, q, l, cur);
                for (uint32_t i = 0; i < P; i++) {
                    hipLaunchKernelGGL(k_af_pivot, dim3((nprob + 63) / 64), dim3(64), 0, st, q, l, i);
                    HIPCHK(ctx, hipMemcpyAsync(ctx->af_h, q.af_pivot, sizeof(double) * (size_t)nprob, hipMemcpyDeviceToHost, st));
                    HIPCHK(ctx, hipStreamSynchronize(st));
                    for (uint32_t k = 0; k < nprob; k++) { const double v = ctx->af_h[k]; ctx->af_h[k] = (v <= 0.0) ? -1.0 : lnn_cholesky_pivot(v); }      /* lpc.c:418-421, host libm */
                    HIPCHK(ctx, hipMemcpyAsync(q.af_pivot, ctx->af_h, sizeof(double) * (size_t)nprob, hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(k_af_column, dim3(nprob), dim3(128), 0, st, q, l, i);
                }
                hipLaunchKernelGGL(k_af_solve, dim3((nprob + 63) / 64), dim3(64), 0, st, q, l);
            }
            hipLaunchKernelGGL(k_af_finish, dim3(((uint32_t)Jq + 255) / 256), dim3(256), 0, st, q, l);
            HIPCHK(ctx, hipGetLastError());
            return LNN_OK;
        };
        /* the layers of one pass over the jobs of plan q (linne_network.c:582-602): lags, Levinson-Durbin, the unit-count search,
         * [the auxiliary-function refinement of the chosen coefficients], the forward pass.  Returns LNN_*; cur_out = which half of
         * `sig` holds the last layer's output */
        uint32_t cur_final = 0;
        auto run_layers = [&](const Plan &q, uint64_t Jq, bool spec_ok, const bool *hall, bool fcfg, bool fall, uint32_t af_iters, bool final_pass) -> int {
            int ret = LNN_OK;
            uint32_t cur = 0;
        for (uint32_t l = 0; l < hs.L; l++) {
            const uint32_t maxu = hs.P[l] < 128u ? hs.P[l] : 128u;
            /* the first two layers nearly always keep one unit: their search pass also writes that trial's forward output */
            const uint32_t fir_spec = (spec_ok && ctx->fir_spec && l + 1 < hs.L) ? 1u : 0u;
            {
                const bool hist_layer = q.hist && hs.P[l] >= 64u;
                /* the general kernels serve what the lanes = jobs kernels do not take -- usually one ragged frame, a launch that is
                 * all latency: it runs beside them on the side stream */
                const bool beside = hist_layer && !hall[l] && ctx->has_side;
                if (beside) {
                    HIPCHK(ctx, hipEventRecord(ctx->fork_ev, st)); HIPCHK(ctx, hipStreamWaitEvent(ctx->side, ctx->fork_ev, 0));
                    const int sp_ = span_begin(ctx, 3, ctx->side); dispatch_autocorr2(ctx->side, q, l, cur, ctx->na_max, (ctx->prod_ok >> l) & 1); span_end(ctx, sp_, ctx->side);
                    HIPCHK(ctx, hipEventRecord(ctx->join_ev, ctx->side));
                }
                if (hist_layer) {                               /* long layer: lanes = jobs kernels for the frames they take (hist_takes) */
                    for (int w = 0; w < 3; w++) {
                        if (hs.P[l] == 64u && w == 1) continue;
                        const int sp_ = span_begin(ctx, 21 + w, st); (void)launch_autocorr_hist(st, q, l, cur, w); span_end(ctx, sp_, st);
                    }
                }
                if (beside) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->join_ev, 0));
                else if (!hall[l]) {
                    const int sp_ = span_begin(ctx, (hs.P[l] >= 32u) ? 3 : 14, st); dispatch_autocorr2(st, q, l, cur, ctx->na_max, (ctx->prod_ok >> l) & 1); span_end(ctx, sp_, st);
                }
            }
            { const int sp_ = span_begin(ctx, 4, st);
              /* one launch per trial, every order on LDS columns -- except that the short trials whose columns fit beside the
               * one-unit trial's ride along with it on a second wave (k_levinson_lds) */
              uint32_t ride = LNN_MAXT;
              if (Jq <= 64u && ctx->lev_wave) {       /* a handful of jobs: a wave per problem, all trials in one launch */
                  hipLaunchKernelGGL(k_levinson_wave, dim3((uint32_t)Jq, 2u * maxu - 1u), dim3(64), 0, st, q, l);
              } else {
              for (uint32_t t = 1, u = 2; u <= maxu && ctx->lev_ride; u <<= 1, t++)
                  if (LEV_LDS(hs.P[l]) + LEV_MAXRIDE * LEV_LDS(hs.P[l] / u) <= LEV_LDS_BUDGET) { ride = t; break; }
              for (uint32_t t = 0, u = 1; u <= maxu && t < (ride < LNN_MAXT ? ride : LNN_MAXT); u <<= 1, t++) {
                  const uint32_t np = hs.P[l] / u;
                  const bool carry = (t == 0 && ride < LNN_MAXT);
                  const size_t lds = LEV_LDS(np) + (carry ? LEV_MAXRIDE * LEV_LDS(hs.P[l] >> ride) : 0);
                  hipLaunchKernelGGL(k_levinson_lds, dim3(((uint32_t)Jq + 63) / 64, u), dim3(carry ? 64 * (1 + LEV_MAXRIDE) : 64), lds, st, q, l, t, carry ? ride : (uint32_t)LNN_MAXT);
              }
              }
              span_end(ctx, sp_, st); }
            /* the last layer of a chunk k_last_layer takes: search, selection, forward pass and loss from one pass over the input */
            const bool ll = last_layer_all && l + 1 == hs.L && fcfg && fall && !final_pass && spec_ok;
            if (ll) {
                { const int sp_ = span_begin(ctx, 20, st);
                  const dim3 g(((uint32_t)Jq + 63) / 64), b(64);
                  switch (hs.P[l]) {
                  case 2: hipLaunchKernelGGL(k_last_layer<2>, g, b, 0, st, q, l, cur); break;
                  case 4: hipLaunchKernelGGL(k_last_layer<4>, g, b, 0, st, q, l, cur); break;
                  case 8: hipLaunchKernelGGL(k_last_layer<8>, g, b, 0, st, q, l, cur); break;
                  default: hipLaunchKernelGGL(k_last_layer<16>, g, b, 0, st, q, l, cur); break;
                  }
                  span_end(ctx, sp_, st); }
                { const int sp_ = span_begin(ctx, 7, st); hipLaunchKernelGGL(k_select, dim3(((uint32_t)Jq + 63) / 64), dim3(64), 0, st, q, l, 2u); span_end(ctx, sp_, st); }
            } else {
            {   /* unit-count search.  Short layers: the register-window kernel.  The long layer: k_search_long for the frames it takes
                 * (search_long_takes), k_fir2<2> for the others (it returns at once for the jobs taken there) */
                bool long_any = false, long_all = true;
                for (uint32_t f = f0; f < f0 + Fc; f++) { const bool t_ = fir_spec && search_long_takes(q, l, ctx->sig_cls[ctx->cur_idx[f]]); long_any |= t_; long_all &= t_; }
                if (long_any) {
                    const int sp_ = span_begin(ctx, 25, st);
                    const dim3 grid((uint32_t)Jq, (S + FIR_TILE - 1) / FIR_TILE), blk(FIR_THREADS);
                    if (ctx->search_two) { if (hs.P[l] == 128u) hipLaunchKernelGGL((k_search_long<128, true>), grid, blk, 0, st, q, l, cur); else hipLaunchKernelGGL((k_search_long<64, true>), grid, blk, 0, st, q, l, cur); }
                    else { if (hs.P[l] == 128u) hipLaunchKernelGGL((k_search_long<128, false>), grid, blk, 0, st, q, l, cur); else hipLaunchKernelGGL((k_search_long<64, false>), grid, blk, 0, st, q, l, cur); }
                    span_end(ctx, sp_, st);
                }
                if (!(long_any && long_all)) {
                    const int sp_ = span_begin(ctx, (l == 0) ? 15 : (fir_spec ? 5 : 18), st);
                    if (hs.P[l] <= 16u && ctx->fir_small) launch_fir_small_search(st, q, l, cur, (uint32_t)Jq, (S + FIR_TILE - 1) / FIR_TILE, fir_spec != 0, hs.P[l]);
                    else if (long_any) {
                        /* the frames k_search_long leaves -- usually the one ragged tail -- are runs of consecutive rows (the chunk is sorted by
                         * class): a launch per run, not 620 k blocks of which all but a handful look up their job and go (0.7 ms) */
                        const uint32_t rpf = (uint32_t)(Jq / Fc);
                        for (uint32_t f = f0; f < f0 + Fc; ) {
                            if (search_long_takes(q, l, ctx->sig_cls[ctx->cur_idx[f]])) { f++; continue; }
                            uint32_t g = f + 1;
                            while (g < f0 + Fc && !search_long_takes(q, l, ctx->sig_cls[ctx->cur_idx[g]])) g++;
                            Plan qq = q; qq.job_off = (f - f0) * rpf;
                            launch_fir<2>(st, qq, l, cur, (g - f) * rpf, (S + FIR_TILE - 1) / FIR_TILE, fir_spec != 0);
                            f = g;
                        }
                    }
                    else launch_fir<2>(st, q, l, cur, (uint32_t)Jq, (S + FIR_TILE - 1) / FIR_TILE, fir_spec != 0);
                    span_end(ctx, sp_, st);
                }
            }
            const bool sel_wave = Jq <= 256u && (uint64_t)((S + FIR_TILE - 1) / FIR_TILE) * (FIR_THREADS / 64) <= SELW_MAXPART;       /* a handful of jobs: a wave per job */
            { const int sp_ = span_begin(ctx, 7, st);
              if (sel_wave) hipLaunchKernelGGL(k_select_wave, dim3((uint32_t)Jq), dim3(64), 0, st, q, l, 0u);
              else hipLaunchKernelGGL(k_select, dim3(((uint32_t)Jq + 63) / 64), dim3(64), 0, st, q, l, 0u);
              span_end(ctx, sp_, st); }
            /* exact ordered chains for the (rare) jobs the certified search flagged; everything else exits at once */
            { const int sp_ = span_begin(ctx, 6, st); if (l == 0) hipLaunchKernelGGL((k_fir2<0, true, false>), dim3((uint32_t)Jq, 1), dim3(FIR_THREADS), 0, st, q, l, cur); else hipLaunchKernelGGL((k_fir2<0, false, false>), dim3((uint32_t)Jq, 1), dim3(FIR_THREADS), 0, st, q, l, cur);
              if (sel_wave) hipLaunchKernelGGL(k_select_wave, dim3((uint32_t)Jq), dim3(64), 0, st, q, l, 1u);
              else hipLaunchKernelGGL(k_select, dim3(((uint32_t)Jq + 63) / 64), dim3(64), 0, st, q, l, 1u);
              span_end(ctx, sp_, st); }
            /* the last layer's output is only ever summed: layers of <= 16 taps do the forward pass and the ordered loss in one
             * kernel and write nothing else */
            if (l + 1 == hs.L && fcfg && !final_pass) {
                const int sp_ = span_begin(ctx, 20, st);
                const dim3 g(((uint32_t)Jq + 63) / 64), b(64), b5(320);
                /* fewer than 1024 waves of 64 jobs: each would be alone on its SIMD (1.4 ms per launch however few) -- five waves per 64 jobs then (k_fwd_loss_mw) */
                if (Jq < 65536u && ctx->knob.fwd_loss_mw) switch (hs.P[l]) {
                case 2: hipLaunchKernelGGL(k_fwd_loss_mw<2>, g, b5, 0, st, q, l, cur); break;
                case 4: hipLaunchKernelGGL(k_fwd_loss_mw<4>, g, b5, 0, st, q, l, cur); break;
                case 8: hipLaunchKernelGGL(k_fwd_loss_mw<8>, g, b5, 0, st, q, l, cur); break;
                default: hipLaunchKernelGGL(k_fwd_loss_mw<16>, g, b5, 0, st, q, l, cur); break;
                }
                else switch (hs.P[l]) {
                case 2: hipLaunchKernelGGL(k_fwd_loss<2>, g, b, 0, st, q, l, cur); break;
                case 4: hipLaunchKernelGGL(k_fwd_loss<4>, g, b, 0, st, q, l, cur); break;
                case 8: hipLaunchKernelGGL(k_fwd_loss<8>, g, b, 0, st, q, l, cur); break;
                default: hipLaunchKernelGGL(k_fwd_loss<16>, g, b, 0, st, q, l, cur); break;
                }
                span_end(ctx, sp_, st);
            }
            }       /* (not k_last_layer's) */
            if (af_iters && (ret = run_af(q, Jq, l, cur, af_iters)) != LNN_OK) return ret;
            if (final_pass && l + 1 == hs.L) { cur ^= 1u; continue; }       /* the final pass needs no output of the last layer: only its parameters */
            if (!(l + 1 == hs.L && fall)) { const int sp_ = span_begin(ctx, (l == 0) ? 16 : (fir_spec ? 8 : 19), st); launch_fir<1>(st, q, l, cur, (uint32_t)Jq, (S + FIR_TILE - 1) / FIR_TILE, fir_spec != 0); span_end(ctx, sp_, st); }
            cur ^= 1u;
        }
            cur_final = cur;
            return ret;
        };
        /* -l: LINNENetworkTrainer_Train on the parameters the analysis left (lnn_k_train.h); synchronous: the host reads after every
         * step how many channel-frames go on */
        auto run_train = [&](const Plan &q, const uint32_t *best) -> int {
            tr.p = q; tr.best = best; tr.CF = (uint32_t)CF;
            const uint32_t tiles = (S + TR_TILE - 1) / TR_TILE;
            const int sp_ = span_begin(ctx, 27, st);
            hipLaunchKernelGGL(k_tr_init, dim3(((uint32_t)CF + 255) / 256), dim3(256), 0, st, tr);
            for (uint32_t it = 0; it < 2000u; it++) {                      /* LINNE_TRAINING_PARAMETER_MAX_NUM_ITRATION, linne_internal.h:29 */
                HIPCHK(ctx, hipMemsetAsync(tr.nactive, 0, sizeof(uint32_t), st));
                for (uint32_t l = 0; l < hs.L; l++) hipLaunchKernelGGL(k_tr_forward, dim3((uint32_t)CF, tiles), dim3(TR_THREADS), 0, st, tr, l);
                hipLaunchKernelGGL(k_tr_loss, dim3((uint32_t)CF), dim3(64), 0, st, tr);                                   /* the loss and, in place, its gradient */
                for (uint32_t l = hs.L - 1; l >= 1; l--) hipLaunchKernelGGL(k_tr_back, dim3((uint32_t)CF, tiles), dim3(TR_THREADS), 0, st, tr, l);
                hipLaunchKernelGGL(k_tr_gradp, dim3((uint32_t)CF, hs.L), dim3(128), 0, st, tr);
                hipLaunchKernelGGL(k_tr_update, dim3((uint32_t)CF), dim3(128), 0, st, tr, (double)0.8f, (double)0.1f, 1.0e-7);     /* linne_network.c:829, linne_internal.h:31-33 */
                uint32_t left = 0;
                HIPCHK(ctx, hipMemcpyAsync(&left, tr.nactive, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
                if (left == 0) break;
            }
            span_end(ctx, sp_, st);
            HIPCHK(ctx, hipGetLastError());
            return LNN_OK;
        };
        const uint32_t af_iters = ctx->af_iters;
        if ((ret = run_layers(p, J, true, hist_all, fuse_cfg, fuse_all, 0u, false)) != LNN_OK) return ret;
        uint32_t cur = cur_final;
        if (!fuse_all) { const int sp_ = span_begin(ctx, 9, st); if (J <= 1024u) hipLaunchKernelGGL(k_chain_sum_wave<1>, dim3((uint32_t)J), dim3(64), 0, st, p, 0u, cur);       /* few rows: a wave per row */
            else hipLaunchKernelGGL(k_chain_sum<1>, dim3(((uint32_t)J + 63) / 64), dim3(SUM_THREADS), 0, st, p, 0u, cur);
            span_end(ctx, sp_, st); }
        if (af_iters == 0) {        /* the final pass of linne_network.c:628-629 repeats the winning pass bit for bit: skipped */
            if (ctx->learning) {
                hipLaunchKernelGGL(k_af_best, dim3(((uint32_t)CF + 255) / 256), dim3(256), 0, st, p, af_best, af_loss, af_reg);
                if ((ret = run_train(p, af_best)) != LNN_OK) return ret;
            }
            const int sp_ = span_begin(ctx, 10, st); hipLaunchKernelGGL(k_quantize, dim3((uint32_t)CF), dim3(64), 0, st, p); hipLaunchKernelGGL(k_fir_cascade, dim3((uint32_t)CF, CF >= 1024u ? 1u : (S + FIN_TILE - 1) / FIN_TILE), dim3(FIN_THREADS), 0, st, p); span_end(ctx, sp_, st);
        } else {
            /* -a N: the final pass is real -- the winner's regulariser, the refinement after every layer's search, and therefore
             * new inputs (and possibly new unit counts) for the layers behind it.  One job per channel-frame, general kernels. */
            Plan q = p;
            q.R = 1; q.J = (uint32_t)CF; q.regs[0] = 0.0;
            q.hist = 0; q.fused_last = 0; q.search_long = 0;
            build_runs(&q.runs[1], ctx->cur_idx + f0, Fc, C);
            q.job_reg = af_reg; q.af_best = af_best; q.af_loss = af_loss;
            hipLaunchKernelGGL(k_af_best, dim3(((uint32_t)CF + 255) / 256), dim3(256), 0, st, p, af_best, af_loss, af_reg);
            const bool none[LNN_MAXL] = { false, false, false };
            const int sp_ = span_begin(ctx, 26, st);
            if ((ret = run_layers(q, CF, false, none, false, false, af_iters, true)) != LNN_OK) return ret;
            span_end(ctx, sp_, st);
            if (ctx->learning && (ret = run_train(q, NULL)) != LNN_OK) return ret;
            { const int sp2_ = span_begin(ctx, 10, st); hipLaunchKernelGGL(k_quantize, dim3((uint32_t)CF), dim3(64), 0, st, q); hipLaunchKernelGGL(k_fir_cascade, dim3((uint32_t)CF, CF >= 1024u ? 1u : (S + FIN_TILE - 1) / FIN_TILE), dim3(FIN_THREADS), 0, st, q); span_end(ctx, sp2_, st); }
        }
        HIPCHK(ctx, hipGetLastError());
    }
    return LNN_OK;
    };
    const int loop_ret = enqueue_chunks();
    if (loop_ret != LNN_OK) {       /* what was forked is waited for before the error goes back (the callers synchronise ctx->stream only) */
        if (use_sub) for (uint32_t i = 0; i < nsub; i++) (void)hipStreamSynchronize(ctx->sub[i]);
        if (ctx->has_side) (void)hipStreamSynchronize(ctx->side);
        return loop_ret;
    }
    if (ctx->has_side) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
    if (use_sub) {
        for (uint32_t i = 0; i < nsub; i++) { HIPCHK(ctx, hipEventRecord(ctx->sub_done[i], ctx->sub[i])); HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->sub_done[i], 0)); }
    }
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream)); ctx->ev_valid = 1; }
    return LNN_OK;
}

/* The throughput and latency forms of the synthesis carry a coefficient as int8 (the format's range: the coefficients are 8-bit
 * Huffman symbols, linne_decoder.c:452-470); the lanes form would take any int32.  A parameter record with a coefficient outside
 * [-128, 127] is no LINNE stream's: the entry points that see the records on the HOST refuse it, so that the result never depends on
 * which form the batch size picked (include/linne_amd.h states the contract for records that are already in HBM). */
static int params_in_range(LINNEAmdContext *ctx, const HostShape *hs, uint32_t C, const int32_t *h_params, uint32_t num_frames)
{
    uint32_t ncoef = 0;
    for (uint32_t l = 0; l < hs->L; l++) ncoef += hs->P[l];
    const uint64_t rows = (uint64_t)num_frames * C;
    uint32_t bad = 0;
    for (uint64_t r = 0; r < rows; r++) {
        const int32_t *c = h_params + r * LINNE_AMD_PARAM_WORDS + LINNE_AMD_PRM_COEF;
        uint32_t acc = 0;
        for (uint32_t i = 0; i < ncoef; i++) acc |= (uint32_t)(c[i] + 128) & ~255u;
        bad |= acc;
    }
    if (bad) { snprintf(ctx->err, sizeof(ctx->err), "a coefficient outside [-128, 127] in the parameter records (not a LINNE stream's)"); return LNN_INVALID_FORMAT; }
    return LNN_OK;
}

extern "C" int LINNEAmd_DecodeFramesDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        int32_t *d_data, const uint32_t *h_num_samples, uint32_t num_frames, const int32_t *d_params)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_data || !d_params) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HostShape hs;
    int ret = shape_info(shape, &hs);
    if (ret != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    read_call_knobs(ctx);
    if ((ret = upload_lengths(ctx, shape, h_num_samples, num_frames)) != LNN_OK) return ret;
    DecPlan p; memset(&p, 0, sizeof(p));
    p.C = shape->num_channels; p.S = shape->num_samples_per_block; p.L = hs.L; p.ms = shape->ch_process_method; p.F = num_frames;
    for (uint32_t l = 0; l < hs.L; l++) { p.P[l] = hs.P[l]; p.coef_off[l] = hs.coef_off[l]; }
    p.data = d_data; p.prm = d_params; p.nsmp = ctx->d_nsmp;
    ctx->nspans = 0;
    bool ms_done = false;
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream)); }
    {   /* layers in reverse order (linne_decoder.c:503-509): long layers one wave per channel-frame, short ones (order <= 16)
         * with lanes = channel-frames; the de-emphasis rides on layer 0's pass */
        const uint32_t CF = num_frames * p.C, gsmall = (CF + 63) / 64;
        if (hs.P[0] > 16) { snprintf(ctx->err, sizeof(ctx->err), "internal: layer 0 of order %u", hs.P[0]); return LNN_NG; }
        /* The lanes = channel-frames kernels have few, long-running waves: a pass over a short layer takes the time of one
         * wave's 10240-step recurrence however small the batch.  Below a few thousand channel-frames the one-wave-per-
         * channel-frame kernel (all layers and the de-emphasis in one launch) finishes sooner. */
        /* LINNE_AMD_DECODE_KERNEL = "wave" / "lanes" / "pipe": for tests and measurements.  Small batches -- block-at-a-time calls
         * above all -- take the pipelined latency form (k_synth_pipe: a wave per stage of the cascade, 16-sample blocks) when the
         * frame fits its LDS image; k_synthesize is its fallback for longer frames */
        const bool pipe_fits = SP_LDS_BYTES(p.S) <= LEV_LDS_BUDGET;
        /* (tools/decode_crossover.py: the pipelined form costs 0.85 ms per 1024 channel-frames, the throughput form 1.25 ms up to ~4000 and
         * 0.12 per 1024 beyond: they cross at 1536; the lanes form the throughput form falls back to, 5.1 ms whatever the batch: at 6144) */
        const bool rows_fit = (p.S & 3u) == 0u && ((uintptr_t)d_data & 15u) == 0u;
        const int form = ctx->knob.decode_kernel ? ctx->knob.decode_kernel : (CF < (rows_fit ? 1536u : 6144u) ? 3 : 0);
        const bool use_pipe = (form == 3) && pipe_fits, use_wave = (form == 1) || (form == 3 && !pipe_fits);
        /* timing kinds: 11 = k_synthesize (all layers in one launch), 30 = k_synth_big, 31 = k_synth_small, 32 = k_synth_pipe, 33 = k_synth_rows<NCH > 0> (a long layer), 36 = k_synth_rows<0> / k_synth_rows8 (a short layer), 34 = k_deemph_lr, 35 = k_synth_l0_de (layer 0 + de-emphasis + MS -> LR) */
        if (use_pipe) { const int sp_ = span_begin(ctx, 32, ctx->stream); hipLaunchKernelGGL(k_synth_pipe, dim3(CF), dim3(64 * (hs.L + 1)), SP_LDS_BYTES(p.S), ctx->stream, p); span_end(ctx, sp_, ctx->stream); }
        else if (use_wave) { const int sp_ = span_begin(ctx, 11, ctx->stream); hipLaunchKernelGGL(k_synthesize, dim3(CF), dim3(64), 0, ctx->stream, p, 0xFFFFFFFFu, 1u); span_end(ctx, sp_, ctx->stream); }
        else for (int32_t l = (int32_t)hs.L - 1; l >= 0; l--) {
            const bool de = (l == 0);
            /* k_synth_rows (four channel-frames per wave, the old taps on the matrix unit) takes the layers without de-emphasis whose
             * order is a preset's, when the samples can travel as 16-byte groups (LINNE_AMD_DECODE_KERNEL=lanes: none) */
            const int nch = hs.P[l] <= 16u ? 0 : (hs.P[l] == 32u ? 1 : (hs.P[l] == 64u ? 3 : (hs.P[l] == 128u ? 7 : -1)));
            if (de && nch == 0 && form != 2 && rows_fit && hs.P[l] <= 4u && ctx->knob.decode_fused) {
                /* the end of the cascade in ONE launch: layer 0, the de-emphasis and MS -> LR on tiles in LDS (lnn_k_decode_fused.h) */
                const int sp_ = span_begin(ctx, 35, ctx->stream);
                if (p.ms && p.C >= 2u && p.C <= 64u && (p.C & (p.C - 1u)) == 0u) { hipLaunchKernelGGL((k_synth_l0_de<true>), dim3(gsmall), dim3(64 * SF_WAVES), 0, ctx->stream, p); ms_done = true; }
                else hipLaunchKernelGGL((k_synth_l0_de<false>), dim3(gsmall), dim3(64 * SF_WAVES), 0, ctx->stream, p);
                span_end(ctx, sp_, ctx->stream);
                continue;
            }
            if (nch >= 0 && form != 2 && rows_fit) {
                const int sp_ = span_begin(ctx, nch > 0 ? 33 : 36, ctx->stream);
                const dim3 grows((CF + 3) / 4);
                switch (nch) {
                case 0:     /* (LINNE_AMD_DECODE_ROWS8=0 / 1: the four- / eight-channel-frame form whatever the batch) */
                        if (!(ctx->knob.rows8 < 0 ? CF >= 20480u : ctx->knob.rows8 != 0)) {     /* (eight per wave are twice the blocks per wave: they pay once the four-per-wave form fills the SIMDs, tools/decode_crossover.py) */ if (hs.P[l] <= 4u) hipLaunchKernelGGL((k_synth_rows<0, 4>), grows, dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_rows<0>), grows, dim3(64), 0, ctx->stream, p, (uint32_t)l); }
                        else if (hs.P[l] <= 4u) hipLaunchKernelGGL((k_synth_rows8<4>), dim3((CF + 7) / 8), dim3(64), 0, ctx->stream, p, (uint32_t)l);
                        else if (hs.P[l] <= 8u) hipLaunchKernelGGL((k_synth_rows8<8>), dim3((CF + 7) / 8), dim3(64), 0, ctx->stream, p, (uint32_t)l);
                        else hipLaunchKernelGGL((k_synth_rows8<16>), dim3((CF + 7) / 8), dim3(64), 0, ctx->stream, p, (uint32_t)l);
                        break;
                case 1: hipLaunchKernelGGL((k_synth_rows<1>), grows, dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
                case 3: hipLaunchKernelGGL((k_synth_rows<3>), grows, dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
                default: hipLaunchKernelGGL((k_synth_rows<7>), grows, dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
                }
                span_end(ctx, sp_, ctx->stream);
                if (de) {       /* the de-emphasis behind layer 0 (lanes = channel-frames), MS -> LR on its way out when a block of 64 rows holds whole frames */
                    const int sd_ = span_begin(ctx, 34, ctx->stream);
                    if (p.ms && p.C >= 2u && p.C <= 64u && (p.C & (p.C - 1u)) == 0u) { hipLaunchKernelGGL((k_deemph_lr<true>), dim3(gsmall), dim3(64 * (2 + DL_STORERS)), 0, ctx->stream, p); ms_done = true; }
                    else hipLaunchKernelGGL((k_deemph_lr<false>), dim3(gsmall), dim3(64 * (2 + DL_STORERS)), 0, ctx->stream, p);
                    span_end(ctx, sd_, ctx->stream);
                }
                continue;
            }
            const int sp_ = span_begin(ctx, hs.P[l] <= 16u ? 31 : (hs.P[l] <= 128u && (hs.P[l] & (hs.P[l] - 1u)) == 0 ? 30 : 11), ctx->stream);
            switch (hs.P[l]) {
            case 2:  if (de) hipLaunchKernelGGL((k_synth_small<2, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<2, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 4:  if (de) hipLaunchKernelGGL((k_synth_small<4, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<4, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 8:  if (de) hipLaunchKernelGGL((k_synth_small<8, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<8, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 16: if (de) hipLaunchKernelGGL((k_synth_small<16, true>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); else hipLaunchKernelGGL((k_synth_small<16, false>), dim3(gsmall), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 32:  hipLaunchKernelGGL((k_synth_big<32>), dim3((CF + 15) / 16), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 64:  hipLaunchKernelGGL((k_synth_big<64>), dim3((CF + 15) / 16), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            case 128: hipLaunchKernelGGL((k_synth_big<128>), dim3((CF + 15) / 16), dim3(64), 0, ctx->stream, p, (uint32_t)l); break;
            default: hipLaunchKernelGGL(k_synthesize, dim3(CF), dim3(64), 0, ctx->stream, p, (uint32_t)l, 0u); break;      /* not a preset size */
            }
            span_end(ctx, sp_, ctx->stream);
        }
    }
    if (p.ms && !ms_done)
        { const int sp_ = span_begin(ctx, 12, ctx->stream); hipLaunchKernelGGL(k_ms_to_lr, dim3(num_frames, (p.S + 255) / 256), dim3(256), 0, ctx->stream, p); span_end(ctx, sp_, ctx->stream); }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream)); ctx->ev_valid = 1; }
    return LNN_OK;
}

/* device staging of the host-buffer forms: one allocation kept between calls (a block-at-a-time caller pays no hipMalloc /
 * hipFree -- and with it no device-wide synchronisation -- per block); batches beyond HSTAGE_KEEP get a transient one */
#define HSTAGE_KEEP ((uint64_t)64 << 20)
static int hstage_get(LINNEAmdContext *ctx, uint64_t bytes, void **out, int *transient)
{
    *transient = 0;
    if (bytes > HSTAGE_KEEP) {
        if (hipMalloc(out, bytes) != hipSuccess) { (void)hipGetLastError(); snprintf(ctx->err, sizeof(ctx->err), "hipMalloc of staging buffers failed"); return LNN_NG; }
        *transient = 1;
        return LNN_OK;
    }
    if (ctx->hstage_cap < bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->hstage) { HIPCHK(ctx, hipFree(ctx->hstage)); ctx->hstage = NULL; ctx->hstage_cap = 0; }
        uint64_t cap = bytes < ((uint64_t)1 << 20) ? ((uint64_t)1 << 20) : bytes;
        if (hipMalloc(&ctx->hstage, cap) != hipSuccess) { (void)hipGetLastError(); ctx->hstage = NULL; snprintf(ctx->err, sizeof(ctx->err), "hipMalloc of staging buffers failed"); return LNN_NG; }
        ctx->hstage_cap = cap;
    }
    *out = ctx->hstage;
    return LNN_OK;
}

/* host-buffer forms (what the block-at-a-time API calls; batch callers use the device entry points or the staging slots) */
extern "C" int LINNEAmd_EncodeFramesHost(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *pcm, const uint32_t *num_samples, uint32_t num_frames,
        int32_t *residual, int32_t *params, double *stats)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    if (!shape || !pcm || !residual || !params || !stats) return LNN_INVALID_ARGUMENT;
    if (num_frames == 0) return LNN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t CS = (uint64_t)shape->num_channels * shape->num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (uint64_t)shape->num_channels * num_frames,
                   sb = sizeof(double) * LINNE_AMD_STAT_WORDS * (uint64_t)shape->num_channels * num_frames;
    int32_t *d_pcm = NULL, *d_res = NULL, *d_prm = NULL; double *d_st = NULL;
    void *stage = NULL; int transient = 0;
    const uint64_t nb_a = align_up(nb), pb_a = align_up(pb);
    int ret = hstage_get(ctx, 2 * nb_a + pb_a + align_up(sb), &stage, &transient);
    if (ret != LNN_OK) return ret;
    d_pcm = (int32_t *)stage; d_res = (int32_t *)((uint8_t *)stage + nb_a); d_prm = (int32_t *)((uint8_t *)stage + 2 * nb_a); d_st = (double *)((uint8_t *)stage + 2 * nb_a + pb_a);
    ret = LNN_NG;
    if (hipMemcpyAsync(d_pcm, pcm, nb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "H2D failed"); goto done; }
    if (hipMemsetAsync(d_prm, 0, pb, ctx->stream) != hipSuccess || hipMemsetAsync(d_st, 0, sb, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "memset failed"); goto done; }
    ret = LINNEAmd_EncodeFramesDevice(ctx, shape, d_pcm, num_samples, num_frames, d_res, d_prm, d_st);
    if (ret != LNN_OK) goto done;
    ret = LNN_NG;
    if (hipMemcpyAsync(residual, d_res, nb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess
            || hipMemcpyAsync(params, d_prm, pb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess
            || hipMemcpyAsync(stats, d_st, sb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "D2H failed"); goto done; }
    {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "stream sync: %s", hipGetErrorString(e)); goto done; }
    }
    ret = LNN_OK;
done:
    if (ret != LNN_OK || transient) hipStreamSynchronize(ctx->stream);
    if (transient) hipFree(stage);
    return ret;
}

extern "C" int LINNEAmd_DecodeFramesHost(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        int32_t *data, const uint32_t *num_samples, uint32_t num_frames, const int32_t *params)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    if (!shape || !data || !params) return LNN_INVALID_ARGUMENT;
    if (num_frames == 0) return LNN_OK;
    { HostShape hs_; int r_ = shape_info(shape, &hs_); if (r_ != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return r_; }
      if ((r_ = params_in_range(ctx, &hs_, shape->num_channels, params, num_frames)) != LNN_OK) return r_; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t CS = (uint64_t)shape->num_channels * shape->num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (uint64_t)shape->num_channels * num_frames;
    int32_t *d_data = NULL, *d_prm = NULL;
    void *stage = NULL; int transient = 0;
    int ret = hstage_get(ctx, align_up(nb) + pb, &stage, &transient);
    if (ret != LNN_OK) return ret;
    d_data = (int32_t *)stage; d_prm = (int32_t *)((uint8_t *)stage + align_up(nb));
    ret = LNN_NG;
    if (hipMemcpyAsync(d_data, data, nb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess
            || hipMemcpyAsync(d_prm, params, pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "H2D failed"); goto done; }
    ret = LINNEAmd_DecodeFramesDevice(ctx, shape, d_data, num_samples, num_frames, d_prm);
    if (ret != LNN_OK) goto done;
    ret = LNN_NG;
    if (hipMemcpyAsync(data, d_data, nb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "D2H failed"); goto done; }
    {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "stream sync: %s", hipGetErrorString(e)); goto done; }
    }
    ret = LNN_OK;
done:
    if (ret != LNN_OK || transient) hipStreamSynchronize(ctx->stream);
    if (transient) hipFree(stage);
    return ret;
}


extern "C" int LINNEAmd_RicePlanDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_residual, const uint32_t *h_num_samples, uint32_t num_frames, uint8_t *d_plan)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_residual || !d_plan) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HostShape hs;
    int ret = shape_info(shape, &hs);
    if (ret != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid shape"); return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->rice_nsteps == 0) ctx->rice_nsteps = lnn_rice_k2_steps(ctx->rice_steps);
    int m;
    if ((ret = meta_acquire(ctx, num_frames, &m)) != LNN_OK) return ret;
    uint32_t *nsm = ctx->meta_h[m];
    for (uint32_t f = 0; f < num_frames; f++) {
        const uint32_t n = h_num_samples ? h_num_samples[f] : shape->num_samples_per_block;
        if (n == 0 || n > shape->num_samples_per_block) { snprintf(ctx->err, sizeof(ctx->err), "frame %u: num_samples %u out of range", f, n); return LNN_INVALID_ARGUMENT; }
        nsm[f] = n;
    }
    if ((ret = ensure_buf(ctx, (void **)&ctx->d_plan_nsmp, &ctx->plan_nsmp_cap, sizeof(uint32_t) * (uint64_t)num_frames)) != LNN_OK) return ret;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_plan_nsmp, nsm, sizeof(uint32_t) * num_frames, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->meta_ev[m], ctx->stream));
    ctx->meta_used[m] = 1;
    RicePlanArgs a; memset(&a, 0, sizeof(a));
    a.resid = d_residual; a.nsmp = ctx->d_plan_nsmp; a.plan = d_plan; a.C = shape->num_channels; a.S = shape->num_samples_per_block;
    a.nsteps = ctx->rice_nsteps;
    for (uint32_t i = 0; i < 32; i++) a.steps[i] = ctx->rice_steps[i];
    const uint64_t CF = (uint64_t)num_frames * shape->num_channels;
    for (uint64_t c0 = 0; c0 < CF; ) {        /* grid.x limit: split very large batches on frame boundaries */
        uint64_t cnt = CF - c0;
        const uint64_t lim = (0x7FFFFFFFull / a.C) * a.C;
        if (cnt > lim) cnt = lim;
        RicePlanArgs b = a;
        b.resid = d_residual + c0 * a.S; b.plan = d_plan + c0 * LINNE_AMD_RICE_PLAN_BYTES; b.nsmp = ctx->d_plan_nsmp + c0 / a.C;
        const int sp_ = span_begin(ctx, 17, ctx->stream);
        if (b.S <= REMIT_LDS_SAMPLES) hipLaunchKernelGGL(k_rice_plan<true>, dim3((uint32_t)cnt), dim3(RICE_THREADS), sizeof(uint32_t) * (b.S + 1025u), ctx->stream, b);
        else hipLaunchKernelGGL(k_rice_plan<false>, dim3((uint32_t)cnt), dim3(RICE_THREADS), 0, ctx->stream, b);
        span_end(ctx, sp_, ctx->stream);
        c0 += cnt;
    }
    HIPCHK(ctx, hipGetLastError());
    return LNN_OK;
}

/* Rice emission on the device: see lnn_k_rice.h.  d_plan is RicePlanDevice's output for the same batch (it carries every
 * channel's code length); enqueues the scan and the emission on the context's stream. */
extern "C" int LINNEAmd_RiceEmitDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_residual, uint32_t num_frames, const uint8_t *d_plan,
        uint32_t *d_offsets, uint8_t *d_packed, uint64_t packed_capacity)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_residual || !d_plan || !d_offsets || !d_packed) { snprintf(ctx->err, sizeof(ctx->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    const uint64_t CF = (uint64_t)num_frames * shape->num_channels;
    if (CF > 0x7FFFFFFFull || ctx->plan_nsmp_cap < sizeof(uint32_t) * (uint64_t)num_frames) { snprintf(ctx->err, sizeof(ctx->err), "RiceEmitDevice: call RicePlanDevice for the same batch first"); return LNN_INVALID_ARGUMENT; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RiceEmitArgs a; memset(&a, 0, sizeof(a));
    a.resid = d_residual; a.nsmp = ctx->d_plan_nsmp; a.plan = d_plan; a.offsets = d_offsets; a.packed = d_packed;
    a.packed_cap = packed_capacity & ~(uint64_t)15; a.C = shape->num_channels; a.S = shape->num_samples_per_block; a.CF = (uint32_t)CF;
    a.cap_bytes = shape->num_samples_per_block * 4u;
    { const char *e_ = getenv("LINNE_AMD_RICE_EMIT_CAP"); if (e_) a.cap_bytes = (uint32_t)atol(e_); }      /* test knob: forces the host fallback */
    { const int sp_ = span_begin(ctx, 24, ctx->stream);
      hipLaunchKernelGGL(k_rice_scan, dim3(1), dim3(RSCAN_THREADS), 0, ctx->stream, a);
      if (a.S <= REMIT_LDS_SAMPLES) hipLaunchKernelGGL(k_rice_emit<true>, dim3((uint32_t)CF), dim3(REMIT_THREADS), sizeof(uint32_t) * (a.S + REMIT_THREADS + 1u), ctx->stream, a);
      else hipLaunchKernelGGL(k_rice_emit<false>, dim3((uint32_t)CF), dim3(REMIT_THREADS), 0, ctx->stream, a);
      span_end(ctx, sp_, ctx->stream); }
    HIPCHK(ctx, hipGetLastError());
    return LNN_OK;
}

/* Rice decoding on the device: see lnn_k_rice.h.  d_stream: the bytes of a group of blocks (4-byte aligned; stream_bytes of it
 * valid, readable up to the next multiple of 8); d_bitpos[f]: where frame f's first channel's code starts (~0: skip the frame);
 * d_endbit[f] receives the bit position behind its last channel's code (~0: the stream held something no encoder writes: the host
 * must decode this group itself).  Enqueues on the context's stream. */
extern "C" int LINNEAmd_RiceDecodeDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const uint8_t *d_stream, uint64_t stream_bytes, const uint64_t *d_bitpos, const uint32_t *h_num_samples, uint32_t num_frames,
        int32_t *d_residual, uint64_t *d_endbit)
{
    if (!ctx) return LNN_INVALID_ARGUMENT;
    ctx->err[0] = 0;
    if (!shape || !d_stream || !d_bitpos || !d_residual || !d_endbit || ((uintptr_t)d_stream & 3u)) { snprintf(ctx->err, sizeof(ctx->err), "RiceDecodeDevice: null or misaligned argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int ret = upload_lengths(ctx, shape, h_num_samples, num_frames);
    if (ret != LNN_OK) return ret;
    RiceDecodeArgs a; memset(&a, 0, sizeof(a));
    a.words = (const uint32_t *)d_stream; a.nbytes = stream_bytes; a.bitpos = d_bitpos; a.nsmp = ctx->d_nsmp; a.resid = d_residual; a.endbit = d_endbit;
    a.F = num_frames; a.C = shape->num_channels; a.S = shape->num_samples_per_block;
    const int sp_ = span_begin(ctx, 28, ctx->stream);
    hipLaunchKernelGGL(k_rice_decode, dim3((num_frames + RDEC_THREADS - 1) / RDEC_THREADS), dim3(RDEC_THREADS), 0, ctx->stream, a);
    span_end(ctx, sp_, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return LNN_OK;
}

/* ================================================================================================
 * staging slots: pinned host buffers + device buffers for a group of frames.  Submit enqueues H2D (copy-in
 * stream), the kernels (context stream) and D2H (copy-out stream) chained by events and returns at once, so a
 * caller that rotates over a few slots overlaps its own host work (entropy stage), PCIe and the kernels.
 * ============================================================================================== */
struct LINNEAmdSlot {
    LINNEAmdContext *ctx; struct LINNEAmdShape shape; uint32_t max_frames; int for_encode; uint32_t flags;
    int32_t *h_pcm, *h_data, *h_prm; double *h_st; uint8_t *h_plan;
    int32_t *d_pcm, *d_data, *d_prm; double *d_st; uint8_t *d_plan;
    /* emit mode: the channels' Rice code, packed back to back, instead of the residual */
    uint8_t *h_packed, *d_packed; uint64_t packed_cap; uint32_t *h_offsets, *d_offsets;
    /* decode, stream mode: the blocks' bytes instead of the residual; PCM optionally as int16 */
    uint8_t *h_stream, *d_stream; uint64_t stream_cap; uint64_t *h_bitpos, *d_bitpos, *h_endbit, *d_endbit; int16_t *h_out16, *d_out16; uint32_t *d_flag, *h_flag;
    hipStream_t in_stream;      /* stream mode: the stream of this slot's Rice decoder, one of the context's pool (not owned) */
    hipEvent_t ev_in, ev_k, ev_done; int pending;
};

static int ctx_copy_streams(LINNEAmdContext *ctx)
{
    if (ctx->has_copy) return LNN_OK;
    /* The three kinds of work a staging pipeline overlaps must sit on different hardware queues: streams that share one run in order
     * whatever their events allow (an H2D behind a kernel that waits for another group's decoder: DecodeWhole took 67 ms in one process
     * and 25 in another, by which queue the copy-in stream had drawn).  Which queue of a priority level's pool a stream gets is
     * round-robin over everything the process ever created at that level -- so the copy-in stream (and the Rice decoders' streams) are
     * created at HIGH priority and the copy-out stream at LOW: three pools, and in each only this library's streams. */
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          /* (numerically lowest = highest priority) */
    if (hipStreamCreateWithPriority(&ctx->copy_in, hipStreamNonBlocking, prio_hi) != hipSuccess) {        /* (a runtime without priorities: plain streams) */
        (void)hipGetLastError();
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
    }
    if (hipStreamCreateWithPriority(&ctx->copy_out, hipStreamNonBlocking, prio_lo) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
    }
    ctx->has_copy = 1;
    return LNN_OK;
}

extern "C" void LINNEAmd_SlotDestroy(struct LINNEAmdSlot *s)
{
    if (!s) return;
    hipSetDevice(s->ctx->device);
    if (s->pending) hipEventSynchronize(s->ev_done);
    if (s->in_stream) hipStreamSynchronize(s->in_stream);
    if (s->h_pcm) hipHostFree(s->h_pcm);
    if (s->h_data) hipHostFree(s->h_data);
    if (s->h_prm) hipHostFree(s->h_prm);
    if (s->h_st) hipHostFree(s->h_st);
    if (s->h_plan) hipHostFree(s->h_plan);
    if (s->d_plan) hipFree(s->d_plan);
    if (s->h_packed) hipHostFree(s->h_packed);
    if (s->d_packed) hipFree(s->d_packed);
    if (s->h_offsets) hipHostFree(s->h_offsets);
    if (s->d_offsets) hipFree(s->d_offsets);
    if (s->h_stream) hipHostFree(s->h_stream);
    if (s->d_stream) hipFree(s->d_stream);
    if (s->h_bitpos) hipHostFree(s->h_bitpos);
    if (s->d_bitpos) hipFree(s->d_bitpos);
    if (s->h_endbit) hipHostFree(s->h_endbit);
    if (s->d_endbit) hipFree(s->d_endbit);
    if (s->h_out16) hipHostFree(s->h_out16);
    if (s->d_out16) hipFree(s->d_out16);
    if (s->h_flag) hipHostFree(s->h_flag);
    if (s->d_flag) hipFree(s->d_flag);
    if (s->d_pcm) hipFree(s->d_pcm);
    if (s->d_data) hipFree(s->d_data);
    if (s->d_prm) hipFree(s->d_prm);
    if (s->d_st) hipFree(s->d_st);
    if (s->ev_in) hipEventDestroy(s->ev_in);
    if (s->ev_k) hipEventDestroy(s->ev_k);
    if (s->ev_done) hipEventDestroy(s->ev_done);
    free(s);
}

extern "C" struct LINNEAmdSlot *LINNEAmd_SlotCreate(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        uint32_t max_frames, int for_encode)
{
    return LINNEAmd_SlotCreateEx(ctx, shape, max_frames, for_encode, 0u);
}

extern "C" struct LINNEAmdSlot *LINNEAmd_SlotCreateEx(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        uint32_t max_frames, int for_encode, uint32_t flags)
{
    HostShape hs;
    if (!ctx) return NULL;
    ctx->err[0] = 0;
    if (!shape || max_frames == 0 || shape_info(shape, &hs) != LNN_OK) { snprintf(ctx->err, sizeof(ctx->err), "SlotCreate: invalid shape"); return NULL; }
    if (hipSetDevice(ctx->device) != hipSuccess || ctx_copy_streams(ctx) != LNN_OK) return NULL;
    LINNEAmdSlot *s = (LINNEAmdSlot *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    if (!for_encode) flags &= (LINNE_AMD_SLOT_STREAM | LINNE_AMD_SLOT_PCM16); else flags &= ~(uint32_t)LINNE_AMD_SLOT_STREAM;
    if (shape->bits_per_sample > 24) flags &= ~(uint32_t)LINNE_AMD_SLOT_PCM16;          /* narrow staging: 2 bytes up to 16 bits, 3 packed bytes up to 24 */
    s->ctx = ctx; s->shape = *shape; s->max_frames = max_frames; s->for_encode = for_encode; s->flags = flags;
    const uint64_t CS = (uint64_t)shape->num_channels * shape->num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * max_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (uint64_t)shape->num_channels * max_frames,
                   sb = sizeof(double) * LINNE_AMD_STAT_WORDS * (uint64_t)shape->num_channels * max_frames;
    hipError_t e = hipSuccess;
    const bool emit = (flags & LINNE_AMD_SLOT_EMIT) != 0, pcm16 = (flags & LINNE_AMD_SLOT_PCM16) != 0;
    const uint64_t nbn = pcm16 ? nb / 4u * (shape->bits_per_sample <= 16 ? 2u : 3u) : nb;      /* bytes of the narrow PCM image */
    /* emit mode: the residual stays on the device; stream-mode decode with int16 PCM: the int32 PCM is only fetched when a block's
     * samples leave the 16-bit range (LINNEAmd_SlotFetchPcm32 allocates then) */
    const bool lazy_data = !for_encode && (flags & LINNE_AMD_SLOT_STREAM) && pcm16;
    if (e == hipSuccess && !emit && !lazy_data) e = hipHostMalloc((void **)&s->h_data, nb, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_prm, pb, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_data, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_prm, pb);
    if (for_encode) {
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_pcm, nbn, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_st, sb, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_pcm, nbn);
        if (emit) {
            /* room for the code of a group: what the samples' own width would take, a little more than any audio that is
             * not emitted RAW anyway needs; channels that do not fit fall back to the host (offset 0xFFFFFFFF) */
            const uint64_t CF = (uint64_t)shape->num_channels * max_frames;
            s->packed_cap = (CF * ((uint64_t)shape->num_samples_per_block * ((shape->bits_per_sample + 7u) / 8u) + 64u) + 4095u) & ~(uint64_t)4095u;
            if (s->packed_cap > 0xFFFFFFF0ull) s->packed_cap = 0xFFFFF000ull;
            if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_packed, s->packed_cap, hipHostMallocDefault);
            if (e == hipSuccess) e = hipMalloc((void **)&s->d_packed, s->packed_cap);
            if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_offsets, sizeof(uint32_t) * (CF + 1), hipHostMallocDefault);
            if (e == hipSuccess) e = hipMalloc((void **)&s->d_offsets, sizeof(uint32_t) * (CF + 1));
        }
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_st, sb);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_plan, (uint64_t)LINNE_AMD_RICE_PLAN_BYTES * shape->num_channels * max_frames, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_plan, (uint64_t)LINNE_AMD_RICE_PLAN_BYTES * shape->num_channels * max_frames);
    }
    if (!for_encode && (flags & LINNE_AMD_SLOT_STREAM)) {
        /* room for the blocks' bytes: what RAW blocks take, and a little more (a COMPRESS block is chosen on an estimate and may
         * come out larger: a group that does not fit is decoded the other way, lnn_api.c) */
        s->stream_cap = ((uint64_t)max_frames * (CS * ((shape->bits_per_sample + 7u) / 8u) + CS / 8u + 1024u) + 4095u) & ~(uint64_t)4095u;
        if (e == hipSuccess) {
            const int k = ctx->rice_next++ % LNN_RICE_STREAMS;
            if (k >= ctx->n_rice_pool) {
                /* HIGH priority: the runtime keeps a pool of hardware queues per priority level, so these streams cannot land on the
                 * queue of the synthesis or of a copy stream whatever else the process created before (which of the normal pool's
                 * queues a stream gets is round-robin over the process's whole history: DecodeWhole took 25 ms in one process and
                 * 67 ms in another before this) -- and a launch of a few dozen latency-bound waves is what should go first anyway */
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      /* (numerically lowest = highest priority) */
                e = hipStreamCreateWithPriority(&ctx->rice_pool[k], hipStreamNonBlocking, hi);
                if (e != hipSuccess) { (void)hipGetLastError(); e = hipStreamCreateWithFlags(&ctx->rice_pool[k], hipStreamNonBlocking); }
                if (e == hipSuccess) ctx->n_rice_pool = k + 1;
            }
            if (e == hipSuccess) s->in_stream = ctx->rice_pool[k];
        }
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_stream, s->stream_cap + 16, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_stream, s->stream_cap + 16);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_bitpos, (2 * sizeof(uint64_t) + sizeof(uint32_t)) * max_frames, hipHostMallocDefault);      /* positions, the blocks' ends, the frames' lengths */
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_bitpos, (2 * sizeof(uint64_t) + sizeof(uint32_t)) * max_frames);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_endbit, sizeof(uint64_t) * max_frames, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_endbit, sizeof(uint64_t) * max_frames);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_flag, sizeof(uint32_t) * 4, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_flag, sizeof(uint32_t) * 4);
        if (flags & LINNE_AMD_SLOT_PCM16) {
            if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_out16, nbn, hipHostMallocDefault);
            if (e == hipSuccess) e = hipMalloc((void **)&s->d_out16, nbn);
        }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_k, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "SlotCreate: %s", hipGetErrorString(e)); LINNEAmd_SlotDestroy(s); return NULL; }
    return s;
}

extern "C" int32_t *LINNEAmd_SlotPcm(struct LINNEAmdSlot *s) { return (s && !(s->flags & LINNE_AMD_SLOT_PCM16)) ? s->h_pcm : NULL; }
extern "C" int16_t *LINNEAmd_SlotPcm16(struct LINNEAmdSlot *s) { return (s && (s->flags & LINNE_AMD_SLOT_PCM16)) ? (s->for_encode ? (int16_t *)s->h_pcm : s->h_out16) : NULL; }
extern "C" uint32_t LINNEAmd_SlotPcmWidth(const struct LINNEAmdSlot *s) { return !s ? 0u : (!(s->flags & LINNE_AMD_SLOT_PCM16) ? 4u : (s->shape.bits_per_sample <= 16 ? 2u : 3u)); }
extern "C" uint8_t *LINNEAmd_SlotStream(struct LINNEAmdSlot *s) { return s ? s->h_stream : NULL; }
extern "C" uint64_t LINNEAmd_SlotStreamCapacity(const struct LINNEAmdSlot *s) { return s ? s->stream_cap : 0; }
extern "C" uint64_t *LINNEAmd_SlotBitPos(struct LINNEAmdSlot *s) { return s ? s->h_bitpos : NULL; }
extern "C" uint64_t *LINNEAmd_SlotBitEnd(struct LINNEAmdSlot *s) { return (s && s->h_bitpos) ? s->h_bitpos + s->max_frames : NULL; }
extern "C" const uint64_t *LINNEAmd_SlotEndBits(struct LINNEAmdSlot *s) { return s ? s->h_endbit : NULL; }
extern "C" int LINNEAmd_SlotPcm16Valid(const struct LINNEAmdSlot *s) { return (s && s->h_out16 && s->h_flag) ? (s->h_flag[0] == 0u) : 0; }
extern "C" const uint8_t *LINNEAmd_SlotPacked(struct LINNEAmdSlot *s) { return s ? s->h_packed : NULL; }
extern "C" const uint32_t *LINNEAmd_SlotOffsets(struct LINNEAmdSlot *s) { return s ? s->h_offsets : NULL; }
extern "C" uint32_t LINNEAmd_SlotFlags(const struct LINNEAmdSlot *s) { return s ? s->flags : 0; }

/* emit mode keeps the residual on the device; the host asks for a frame's residual only when it has to code a channel
 * itself (flagged plan, code that did not fit): synchronous, valid until the slot is submitted again */
extern "C" int LINNEAmd_SlotFetchResidual(struct LINNEAmdSlot *s, uint32_t frame, int32_t *dst)
{
    if (!s || !dst || !s->for_encode || frame >= s->max_frames) return LNN_INVALID_ARGUMENT;
    const uint64_t fb = sizeof(int32_t) * (uint64_t)s->shape.num_channels * s->shape.num_samples_per_block;
    if (hipSetDevice(s->ctx->device) != hipSuccess || hipMemcpy(dst, (const uint8_t *)s->d_data + fb * frame, fb, hipMemcpyDeviceToHost) != hipSuccess) return LNN_NG;
    return LNN_OK;
}
extern "C" int32_t *LINNEAmd_SlotData(struct LINNEAmdSlot *s) { return s ? s->h_data : NULL; }
extern "C" int32_t *LINNEAmd_SlotParams(struct LINNEAmdSlot *s) { return s ? s->h_prm : NULL; }
extern "C" double *LINNEAmd_SlotStats(struct LINNEAmdSlot *s) { return s ? s->h_st : NULL; }
extern "C" uint8_t *LINNEAmd_SlotRicePlan(struct LINNEAmdSlot *s) { return s ? s->h_plan : NULL; }
extern "C" uint32_t LINNEAmd_SlotCapacity(const struct LINNEAmdSlot *s) { return s ? s->max_frames : 0; }

extern "C" int LINNEAmd_SlotWait(struct LINNEAmdSlot *s)
{
    if (!s) return LNN_INVALID_ARGUMENT;
    if (!s->pending) return LNN_OK;
    HIPCHK(s->ctx, hipEventSynchronize(s->ev_done));
    s->pending = 0;
    return LNN_OK;
}

extern "C" int LINNEAmd_SlotEncodeSubmit(struct LINNEAmdSlot *s, const uint32_t *num_samples, uint32_t num_frames)
{
    if (!s || !s->for_encode) return LNN_INVALID_ARGUMENT;
    LINNEAmdContext *ctx = s->ctx;
    if (num_frames == 0 || num_frames > s->max_frames) { snprintf(ctx->err, sizeof(ctx->err), "SlotEncodeSubmit: %u frames in a slot of %u", num_frames, s->max_frames); return LNN_INVALID_ARGUMENT; }
    int ret = LINNEAmd_SlotWait(s);
    if (ret != LNN_OK) return ret;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t C = s->shape.num_channels, CS = C * s->shape.num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * C * num_frames, sb = sizeof(double) * LINNE_AMD_STAT_WORDS * C * num_frames;
    const bool emit = (s->flags & LINNE_AMD_SLOT_EMIT) != 0, pcm16 = (s->flags & LINNE_AMD_SLOT_PCM16) != 0;
    const uint32_t width = LINNEAmd_SlotPcmWidth(s);
    HIPCHK(ctx, hipMemcpyAsync(s->d_pcm, s->h_pcm, nb / 4u * width, hipMemcpyHostToDevice, ctx->copy_in));
    HIPCHK(ctx, hipEventRecord(s->ev_in, ctx->copy_in));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_in, 0));
    HIPCHK(ctx, hipMemsetAsync(s->d_st, 0, sb, ctx->stream));
    ctx->pcm16_next = pcm16 ? (width == 2u ? 1 : 2) : 0;
    if ((ret = LINNEAmd_EncodeFramesDevice(ctx, &s->shape, s->d_pcm, num_samples, num_frames, s->d_data, s->d_prm, s->d_st)) != LNN_OK) return ret;
    /* The Rice planning and emission are light integer kernels behind the analysis: they run on the copy-out stream, where
     * they overlap the next group's (FP64-bound) analysis instead of delaying it; the slots of a context share that stream,
     * so their use of the context's plan metadata stays ordered. */
    HIPCHK(ctx, hipEventRecord(s->ev_k, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_out, s->ev_k, 0));
    {
        hipStream_t keep = ctx->stream;
        ctx->stream = ctx->copy_out;
        ret = LINNEAmd_RicePlanDevice(ctx, &s->shape, s->d_data, num_samples, num_frames, s->d_plan);
        if (ret == LNN_OK && emit) ret = LINNEAmd_RiceEmitDevice(ctx, &s->shape, s->d_data, num_frames, s->d_plan, s->d_offsets, s->d_packed, s->packed_cap);
        ctx->stream = keep;
        if (ret != LNN_OK) return ret;
    }
    if (emit) {         /* the code's used bytes (a size only the device knows so far) by a copy kernel, then the offsets */
        hipLaunchKernelGGL(k_copy_out, dim3(96), dim3(256), 0, ctx->copy_out, (const uint4 *)s->d_packed, (uint4 *)s->h_packed, (const uint32_t *)(s->d_offsets + (size_t)C * num_frames));
        HIPCHK(ctx, hipMemcpyAsync(s->h_offsets, s->d_offsets, sizeof(uint32_t) * (C * num_frames + 1), hipMemcpyDeviceToHost, ctx->copy_out));
    } else
    HIPCHK(ctx, hipMemcpyAsync(s->h_data, s->d_data, nb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipMemcpyAsync(s->h_prm, s->d_prm, pb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipMemcpyAsync(s->h_st, s->d_st, sb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipMemcpyAsync(s->h_plan, s->d_plan, (uint64_t)LINNE_AMD_RICE_PLAN_BYTES * C * num_frames, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipEventRecord(s->ev_done, ctx->copy_out));
    s->pending = 1;
    return LNN_OK;
}

extern "C" int LINNEAmd_SlotDecodeSubmit(struct LINNEAmdSlot *s, const uint32_t *num_samples, uint32_t num_frames)
{
    if (!s) return LNN_INVALID_ARGUMENT;
    LINNEAmdContext *ctx = s->ctx;
    if (num_frames == 0 || num_frames > s->max_frames) { snprintf(ctx->err, sizeof(ctx->err), "SlotDecodeSubmit: %u frames in a slot of %u", num_frames, s->max_frames); return LNN_INVALID_ARGUMENT; }
    int ret = LINNEAmd_SlotWait(s);
    if (ret != LNN_OK) return ret;
    { HostShape hs_; if ((ret = shape_info(&s->shape, &hs_)) != LNN_OK || (ret = params_in_range(ctx, &hs_, s->shape.num_channels, s->h_prm, num_frames)) != LNN_OK) return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t C = s->shape.num_channels, CS = C * s->shape.num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * C * num_frames;
    HIPCHK(ctx, hipMemcpyAsync(s->d_data, s->h_data, nb, hipMemcpyHostToDevice, ctx->copy_in));
    HIPCHK(ctx, hipMemcpyAsync(s->d_prm, s->h_prm, pb, hipMemcpyHostToDevice, ctx->copy_in));
    HIPCHK(ctx, hipEventRecord(s->ev_in, ctx->copy_in));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_in, 0));
    if ((ret = LINNEAmd_DecodeFramesDevice(ctx, &s->shape, s->d_data, num_samples, num_frames, s->d_prm)) != LNN_OK) return ret;
    HIPCHK(ctx, hipEventRecord(s->ev_k, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_out, s->ev_k, 0));
    HIPCHK(ctx, hipMemcpyAsync(s->h_data, s->d_data, nb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipEventRecord(s->ev_done, ctx->copy_out));
    s->pending = 1;
    return LNN_OK;
}

/* decode slot in stream mode: SlotStream holds the blocks' bytes (stream_bytes of them), SlotBitPos[f] where frame f's Rice code
 * starts, SlotParams the parameters; H2D, Rice decoding, synthesis, [int16 narrowing], D2H are enqueued; after SlotWait:
 * SlotEndBits (~0 anywhere: decode this group on the host instead), SlotData or -- if SlotPcm16Valid -- SlotPcm16 */
extern "C" int LINNEAmd_SlotDecodeStreamSubmit(struct LINNEAmdSlot *s, uint64_t stream_bytes, const uint32_t *num_samples, uint32_t num_frames)
{
    if (!s || s->for_encode || !s->h_stream) return LNN_INVALID_ARGUMENT;
    LINNEAmdContext *ctx = s->ctx;
    if (num_frames == 0 || num_frames > s->max_frames || stream_bytes > s->stream_cap) { snprintf(ctx->err, sizeof(ctx->err), "SlotDecodeStreamSubmit: %u frames / %llu bytes in a slot of %u / %llu", num_frames, (unsigned long long)stream_bytes, s->max_frames, (unsigned long long)s->stream_cap); return LNN_INVALID_ARGUMENT; }
    int ret = LINNEAmd_SlotWait(s);
    if (ret != LNN_OK) return ret;
    { HostShape hs_; if ((ret = shape_info(&s->shape, &hs_)) != LNN_OK || (ret = params_in_range(ctx, &hs_, s->shape.num_channels, s->h_prm, num_frames)) != LNN_OK) return ret; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    /* the copies in on the context's copy-in stream, one group behind the other; the Rice decoder on one of the pooled streams, so that
     * the groups' decoders run side by side (a pooled stream is shared by every fourth group: its decoder waits for the one four groups
     * back, which has had 9 ms by then) */
    hipStream_t cin = ctx->copy_in, rst = s->in_stream ? s->in_stream : ctx->copy_in;
    const uint64_t C = s->shape.num_channels, CS = C * s->shape.num_samples_per_block;
    const uint64_t nb = sizeof(int32_t) * CS * num_frames, pb = sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * C * num_frames;
    memset(s->h_stream + stream_bytes, 0, 16);                 /* the reader loads whole 8-byte words */
    HIPCHK(ctx, hipMemcpyAsync(s->d_stream, s->h_stream, (stream_bytes + 15u) & ~(uint64_t)7u, hipMemcpyHostToDevice, cin));
    {   /* the frames' lengths travel with the bit positions: the Rice decoder runs on the copy-in stream, beside the synthesis of
         * the group before (k_rice_decode has a wave per 64 frames -- a few dozen waves -- and leaves the chip to it) */
        uint32_t *h_nsm = (uint32_t *)(s->h_bitpos + 2 * (size_t)s->max_frames);
        const uint32_t S_ = s->shape.num_samples_per_block;
        for (uint32_t f = 0; f < num_frames; f++) {
            const uint32_t n_ = num_samples ? num_samples[f] : S_;
            if (n_ == 0 || n_ > S_) { snprintf(ctx->err, sizeof(ctx->err), "frame %u: num_samples %u out of range", f, n_); return LNN_INVALID_ARGUMENT; }
            h_nsm[f] = n_;
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(s->d_bitpos, s->h_bitpos, 2 * sizeof(uint64_t) * s->max_frames + sizeof(uint32_t) * num_frames, hipMemcpyHostToDevice, cin));
    HIPCHK(ctx, hipMemcpyAsync(s->d_prm, s->h_prm, pb, hipMemcpyHostToDevice, cin));
    {
        RiceDecodeArgs a; memset(&a, 0, sizeof(a));
        a.words = (const uint32_t *)s->d_stream; a.nbytes = stream_bytes; a.bitpos = s->d_bitpos; a.nsmp = (const uint32_t *)(s->d_bitpos + 2 * (size_t)s->max_frames); a.bitend = s->d_bitpos + s->max_frames;
        a.resid = s->d_data; a.endbit = s->d_endbit; a.F = num_frames; a.C = s->shape.num_channels; a.S = s->shape.num_samples_per_block;
        if (rst != cin) { HIPCHK(ctx, hipEventRecord(s->ev_in, cin)); HIPCHK(ctx, hipStreamWaitEvent(rst, s->ev_in, 0)); }
        const int sp_ = span_begin(ctx, 28, rst);
        hipLaunchKernelGGL(k_rice_decode, dim3((num_frames + RDEC_THREADS - 1) / RDEC_THREADS), dim3(RDEC_THREADS), 0, rst, a);
        span_end(ctx, sp_, rst);
    }
    HIPCHK(ctx, hipEventRecord(s->ev_in, rst));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_in, 0));
    if ((ret = LINNEAmd_DecodeFramesDevice(ctx, &s->shape, s->d_data, num_samples, num_frames, s->d_prm)) != LNN_OK) return ret;
    if (s->d_out16) {
        HIPCHK(ctx, hipMemsetAsync(s->d_flag, 0, sizeof(uint32_t), ctx->stream));
        if (LINNEAmd_SlotPcmWidth(s) == 2u)
            hipLaunchKernelGGL(k_narrow16, dim3(1024), dim3(256), 0, ctx->stream, (const int32_t *)s->d_data, s->d_out16, CS * num_frames, s->d_flag,
                    (const uint32_t *)(s->d_bitpos + 2 * (size_t)s->max_frames), (uint32_t)C, s->shape.num_samples_per_block);
        else
            hipLaunchKernelGGL(k_narrow24, dim3(1024), dim3(256), 0, ctx->stream, (const int32_t *)s->d_data, (uint8_t *)s->d_out16, CS * num_frames, s->d_flag,
                    (const uint32_t *)(s->d_bitpos + 2 * (size_t)s->max_frames), (uint32_t)C, s->shape.num_samples_per_block);
    }
    HIPCHK(ctx, hipEventRecord(s->ev_k, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_out, s->ev_k, 0));
    HIPCHK(ctx, hipMemcpyAsync(s->h_endbit, s->d_endbit, sizeof(uint64_t) * num_frames, hipMemcpyDeviceToHost, ctx->copy_out));
    if (s->d_out16) {
        HIPCHK(ctx, hipMemcpyAsync(s->h_flag, s->d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->copy_out));
        HIPCHK(ctx, hipMemcpyAsync(s->h_out16, s->d_out16, nb / 4u * LINNEAmd_SlotPcmWidth(s), hipMemcpyDeviceToHost, ctx->copy_out));
    } else
        HIPCHK(ctx, hipMemcpyAsync(s->h_data, s->d_data, nb, hipMemcpyDeviceToHost, ctx->copy_out));
    HIPCHK(ctx, hipEventRecord(s->ev_done, ctx->copy_out));
    s->pending = 1;
    return LNN_OK;
}

/* the int32 PCM of a stream-mode decode slot whose int16 copy is not valid (LINNEAmd_SlotPcm16Valid == 0): synchronous */
extern "C" int LINNEAmd_SlotFetchPcm32(struct LINNEAmdSlot *s, uint32_t num_frames)
{
    if (!s || s->for_encode || num_frames > s->max_frames) return LNN_INVALID_ARGUMENT;
    const uint64_t nb = sizeof(int32_t) * (uint64_t)s->shape.num_channels * s->shape.num_samples_per_block * num_frames;
    if (hipSetDevice(s->ctx->device) != hipSuccess) return LNN_NG;
    if (!s->h_data && hipHostMalloc((void **)&s->h_data, sizeof(int32_t) * (uint64_t)s->shape.num_channels * s->shape.num_samples_per_block * s->max_frames, hipHostMallocDefault) != hipSuccess) { s->h_data = NULL; return LNN_NG; }
    if (hipMemcpy(s->h_data, s->d_data, nb, hipMemcpyDeviceToHost) != hipSuccess) return LNN_NG;
    return LNN_OK;
}

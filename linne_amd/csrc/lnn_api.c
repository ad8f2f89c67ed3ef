/*
 * lnn_api.c -- the LINNE public C API (include/linne_encoder.h, include/linne_decoder.h) on top of the
 * gfx950 hot path (lnn_device.hip) and the host entropy stage (lnn_entropy.c).
 *
 * Same signatures, ownership and error conventions as the reference; citations are
 * file:line under /root/reference.  There is no CPU fallback for the prediction path: without a HIP device
 * EncodeBlock / EncodeWhole / DecodeBlock / DecodeWhole of COMPRESS data return LINNE_APIRESULT_NG and say
 * so on stderr.
 */
#define _GNU_SOURCE
#include <sched.h>
#include "linne_encoder.h"
#include "linne_decoder.h"
#include "lnn_host.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define LNN_ALIGN 16u
#define LNN_ENC_SLOTS 4u               /* EncodeWhole rotates its groups over four slots per device (a slot is a GPU-sized group) */
#define LNN_SLOTS 8u                    /* groups of frames in flight in EncodeWhole / DecodeWhole */
#define ALIGN_UP(v) (((v) + (LNN_ALIGN - 1u)) & ~(uintptr_t)(LNN_ALIGN - 1u))


/* host threads of the entropy stage: LINNE_AMD_THREADS, else the CPUs this process may actually use (affinity mask,
 * capped by the cgroup v2 CPU quota: a container often sees every core of the machine but owns a few) */
static uint32_t default_threads(void)
{
    const char *e = getenv("LINNE_AMD_THREADS");
    long n;
    if (e) n = atol(e);
    else {
        cpu_set_t set;
        FILE *fp;
        n = sysconf(_SC_NPROCESSORS_ONLN);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
        if ((fp = fopen("/sys/fs/cgroup/cpu.max", "r")) != NULL) {
            long long quota = 0, period = 0;
            if (fscanf(fp, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
                const long q = (long)((quota + period - 1) / period);
                if (q < n) n = q;
            }
            fclose(fp);
        }
    }
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return (uint32_t)n;
}
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int trace_on(void) { static int v = -1; if (v < 0) { const char *e = getenv("LINNE_AMD_TRACE"); v = e ? atoi(e) : 0; } return v; }
/* Frames per staging slot: LINNE_AMD_GROUP, else a group large enough that the GPU runs its large-batch kernel forms on it
 * (EncodeFramesDevice picks them from 24 576 jobs = channel-frames x regulariser passes, DecodeFramesDevice from 6 144
 * channel-frames: smaller groups are latency-bound and took 1.7x the GPU time for the same stream), capped by what a slot may
 * pin (about 1 GiB per buffer); a long stream then splits into equal groups so that the host stage, PCIe and the kernels
 * overlap from the second group on. */
static uint32_t default_group(uint32_t num_frames, const struct LINNEAmdShape *shape, const struct lnn_layers *ly, int for_encode)
{
    const char *e = getenv("LINNE_AMD_GROUP");
    const uint64_t rows = (uint64_t)shape->num_channels * (for_encode ? ly->num_regs : 1u);
    const char *ej = for_encode ? getenv("LINNE_AMD_GROUP_JOBS") : NULL;
    const uint64_t jobs = (ej && atol(ej) >= 64) ? (uint64_t)atol(ej) : (for_encode ? 30720u : 6144u);
    const uint64_t want = (jobs + rows - 1) / rows;     /* a quarter above the threshold: the lanes = jobs kernels are still filling the chip there */
    const uint64_t cap = (1ull << 30) / ((uint64_t)shape->num_channels * shape->num_samples_per_block * sizeof(int32_t)) + 1;
    uint64_t n, ngroups;
    if (e) { long v = atol(e); if (v < 1) v = 1; if (v > 4096) v = 4096; return (uint32_t)v; }
    n = want < cap ? want : cap;
    if (n < 256) n = 256;
    if (n >= num_frames) return num_frames ? num_frames : 1u;
    ngroups = num_frames / n;                               /* equal groups, none below the threshold */
    return (uint32_t)((num_frames + ngroups - 1) / ngroups);
}
/* The GPUs of a handle: LINNE_AMD_DEVICES="0,1,..." (whole streams fan out over them: group g of frames goes to device
 * g mod G), else LINNE_AMD_DEVICE (one device), else device 0.  Block-at-a-time calls use the first one. */
struct lnn_gpus {
    uint32_t ndev; int device[LNN_MAX_DEVICES];
    struct LINNEAmdContext *ctx[LNN_MAX_DEVICES];
    struct LINNEAmdSlot *slot[LNN_MAX_DEVICES][LNN_SLOTS];      /* whole-stream staging (pinned + device), created at the first Whole call */
    struct LINNEAmdShape slot_shape; uint32_t slot_frames, slot_flags;
};
static void drop_slots(struct lnn_gpus *g)
{
    uint32_t d, i;
    for (d = 0; d < LNN_MAX_DEVICES; d++)
        for (i = 0; i < LNN_SLOTS; i++) { if (g->slot[d][i]) LINNEAmd_SlotDestroy(g->slot[d][i]); g->slot[d][i] = NULL; }
}
/* (re)creates the staging slots of a handle for `frames` frames per slot, `count` per device; keeps what already fits */
static int want_slots(struct lnn_gpus *g, const struct LINNEAmdShape *shape, uint32_t frames, uint32_t count, int mode)
{
    /* mode: 1 encode, 0 decode (residual in, PCM out), 2 decode in stream mode (the blocks' bytes in, int16 PCM out where it fits) */
    const int for_encode = (mode == 1);
    uint32_t d, i, flags = (mode == 2) ? (LINNE_AMD_SLOT_STREAM | (shape->bits_per_sample <= 24 ? LINNE_AMD_SLOT_PCM16 : 0u)) : 0;
    if (for_encode) {       /* 16-bit staging and Rice emission on the device (LINNE_AMD_EMIT=0: int32 both ways, the host codes the residual) */
        const char *e = getenv("LINNE_AMD_EMIT");
        if (!e || atoi(e) != 0) flags = LINNE_AMD_SLOT_PCM16 | LINNE_AMD_SLOT_EMIT;
    }
    if (memcmp(&g->slot_shape, shape, sizeof(*shape)) != 0 || g->slot_frames < frames || g->slot_flags != flags) { drop_slots(g); g->slot_shape = *shape; g->slot_frames = frames; g->slot_flags = flags; }
    for (d = 0; d < g->ndev; d++)
        for (i = 0; i < count && i < LNN_SLOTS; i++)
            if (!g->slot[d][i] && !(g->slot[d][i] = LINNEAmd_SlotCreateEx(g->ctx[d], shape, g->slot_frames, for_encode, flags))) return (int)d + 1;
    return 0;
}
/* opens the first device (block-at-a-time calls) or all of them (whole streams) */
static int open_gpus(struct lnn_gpus *g, const char *who, int all)
{
    uint32_t d;
    if (g->ndev == 0) {
        g->ndev = lnn_parse_device_list(getenv("LINNE_AMD_DEVICES"), g->device, LNN_MAX_DEVICES);
        if (g->ndev == 0) { const char *e = getenv("LINNE_AMD_DEVICE"); g->device[0] = e ? atoi(e) : 0; g->ndev = 1; }
    }
    for (d = 0; d < (all ? g->ndev : 1u); d++) {
        if (g->ctx[d]) continue;
        g->ctx[d] = LINNEAmd_ContextCreate(g->device[d], 256ull << 20);
        if (!g->ctx[d]) {
            fprintf(stderr, "liblinne_amd: %s: no usable HIP device %d (LINNE_AMD_DEVICES / LINNE_AMD_DEVICE); the prediction path has no CPU fallback\n", who, g->device[d]);
            return LNN_NG;
        }
    }
    return LNN_OK;
}
static void close_gpus(struct lnn_gpus *g)
{
    uint32_t d;
    drop_slots(g);
    for (d = 0; d < LNN_MAX_DEVICES; d++) if (g->ctx[d]) { LINNEAmd_ContextDestroy(g->ctx[d]); g->ctx[d] = NULL; }
}

static void put_be16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static uint32_t get_be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }
static uint32_t get_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* header validity shared by encoder and decoder (linne_encoder.c:68-102, linne_decoder.c:134-184) */
static int header_fields_ok(const struct LINNEHeader *h)
{
    if (h->num_channels == 0 || h->num_samples == 0 || h->sampling_rate == 0 || h->bits_per_sample == 0
            || h->num_samples_per_block == 0 || h->preset >= LINNE_NUM_PARAMETER_PRESETS
            || (int)h->ch_process_method >= (int)LINNE_CH_PROCESS_METHOD_INVALID || (int)h->ch_process_method < 0) return 0;
    if (h->ch_process_method == LINNE_CH_PROCESS_METHOD_MS && h->num_channels == 1) return 0;
    return 1;
}

/* ================================================================================================ encoder */
struct LINNEEncoder {
    struct LINNEHeader header;
    uint32_t max_num_channels, max_num_samples_per_block, max_num_layers, max_num_parameters_per_layer;
    uint8_t set_parameter;
    uint8_t alloced_by_own;
    void *work;
    struct LINNEAmdShape shape;
    struct lnn_layers layers;
    struct lnn_gpus gpus;               /* GPU resources: outside the work area, released by Destroy */
    double parcor_state;                /* oracle quirk Q2, carried from block to block */
    uint32_t af_iters, learning;        /* num_afmethod_iterations / enable_learning of the last SetEncodeParameter */
    int32_t *pcm, *residual, *params;   /* one-frame staging, inside the work area */
    double *stats;
};

LINNEApiResult LINNEEncoder_EncodeHeader(const struct LINNEHeader *header, uint8_t *data, uint32_t data_size)
{
    if (header == NULL || data == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (data_size < LINNE_HEADER_SIZE) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    if (!header_fields_ok(header)) return LINNE_APIRESULT_INVALID_FORMAT;
    memcpy(data, "IBRA", 4);
    put_be32(data + 4, LINNE_FORMAT_VERSION);       /* the macro values are written, not the struct's */
    put_be32(data + 8, LINNE_CODEC_VERSION);
    put_be16(data + 12, header->num_channels);
    put_be32(data + 14, header->num_samples);
    put_be32(data + 18, header->sampling_rate);
    put_be16(data + 22, header->bits_per_sample);
    put_be32(data + 24, header->num_samples_per_block);
    data[28] = header->preset;
    data[29] = (uint8_t)header->ch_process_method;
    return LINNE_APIRESULT_OK;
}

static int encoder_config_ok(const struct LINNEEncoderConfig *c)
{
    if (c == NULL) return 0;
    if (c->max_num_samples_per_block == 0 || c->max_num_channels == 0 || c->max_num_layers == 0 || c->max_num_parameters_per_layer == 0) return 0;
    if (c->max_num_parameters_per_layer > c->max_num_samples_per_block) return 0;
    return 1;
}

int32_t LINNEEncoder_CalculateWorkSize(const struct LINNEEncoderConfig *config)
{
    uint64_t sz, cs;
    if (!encoder_config_ok(config)) return -1;
    cs = (uint64_t)config->max_num_channels * config->max_num_samples_per_block;
    sz = sizeof(struct LINNEEncoder) + LNN_ALIGN;
    sz += 2 * (cs * sizeof(int32_t) + LNN_ALIGN);
    sz += (uint64_t)config->max_num_channels * (LINNE_AMD_PARAM_WORDS * sizeof(int32_t) + LINNE_AMD_STAT_WORDS * sizeof(double)) + 2 * LNN_ALIGN;
    if (sz > 0x7FFFFFFFull) return -1;
    return (int32_t)sz;
}

struct LINNEEncoder *LINNEEncoder_Create(const struct LINNEEncoderConfig *config, void *work, int32_t work_size)
{
    struct LINNEEncoder *enc;
    uint8_t own = 0, *p;
    uint64_t cs;
    if (work == NULL && work_size == 0) {
        if ((work_size = LINNEEncoder_CalculateWorkSize(config)) < 0) return NULL;
        work = malloc((size_t)work_size);
        own = 1;
    }
    if (config == NULL || work == NULL || !encoder_config_ok(config) || work_size < LINNEEncoder_CalculateWorkSize(config)) {
        if (own) free(work);
        return NULL;
    }
    p = (uint8_t *)ALIGN_UP((uintptr_t)work);
    enc = (struct LINNEEncoder *)p; p += sizeof(*enc);
    memset(enc, 0, sizeof(*enc));
    enc->work = work; enc->alloced_by_own = own;
    enc->max_num_channels = config->max_num_channels;
    enc->max_num_samples_per_block = config->max_num_samples_per_block;
    enc->max_num_layers = config->max_num_layers;
    enc->max_num_parameters_per_layer = config->max_num_parameters_per_layer;
    cs = (uint64_t)config->max_num_channels * config->max_num_samples_per_block;
    p = (uint8_t *)ALIGN_UP((uintptr_t)p); enc->pcm = (int32_t *)p; p += cs * sizeof(int32_t);
    p = (uint8_t *)ALIGN_UP((uintptr_t)p); enc->residual = (int32_t *)p; p += cs * sizeof(int32_t);
    p = (uint8_t *)ALIGN_UP((uintptr_t)p); enc->params = (int32_t *)p; p += (uint64_t)config->max_num_channels * LINNE_AMD_PARAM_WORDS * sizeof(int32_t);
    p = (uint8_t *)ALIGN_UP((uintptr_t)p); enc->stats = (double *)p;
    lnn_tables_init();
    return enc;
}

void LINNEEncoder_Destroy(struct LINNEEncoder *encoder)
{
    if (encoder == NULL) return;
    close_gpus(&encoder->gpus);
    if (encoder->alloced_by_own == 1) free(encoder->work);
}

LINNEApiResult LINNEEncoder_SetEncodeParameter(struct LINNEEncoder *encoder, const struct LINNEEncodeParameter *parameter)
{
    struct LINNEHeader h;
    struct LINNEAmdShape shape;
    struct lnn_layers ly;
    uint32_t l;
    if (encoder == NULL || parameter == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    /* linne_encoder.c:141-198 */
    if (parameter->num_channels == 0 || parameter->bits_per_sample == 0 || parameter->sampling_rate == 0
            || parameter->num_samples_per_block == 0 || parameter->preset >= LINNE_NUM_PARAMETER_PRESETS
            || (int)parameter->ch_process_method >= (int)LINNE_CH_PROCESS_METHOD_INVALID || (int)parameter->ch_process_method < 0) return LINNE_APIRESULT_INVALID_FORMAT;
    shape.num_channels = parameter->num_channels; shape.bits_per_sample = parameter->bits_per_sample;
    shape.num_samples_per_block = parameter->num_samples_per_block; shape.preset = parameter->preset;
    shape.ch_process_method = (uint32_t)parameter->ch_process_method;
    if (lnn_shape_layers(&shape, &ly) != 0) return LINNE_APIRESULT_INVALID_FORMAT;
    for (l = 0; l < ly.num_layers; l++) if (parameter->num_samples_per_block <= ly.size[l]) return LINNE_APIRESULT_INVALID_FORMAT;
    /* capacity (linne_encoder.c:426-444) */
    if (encoder->max_num_samples_per_block < parameter->num_samples_per_block || encoder->max_num_channels < parameter->num_channels) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    if (encoder->max_num_layers < ly.num_layers) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    for (l = 0; l < ly.num_layers; l++) if (encoder->max_num_parameters_per_layer < ly.size[l]) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    encoder->learning = parameter->enable_learning ? 1u : 0u;   /* -l: linne_network.c:805-873 on the device (lnn_k_train.h) */
    encoder->af_iters = parameter->num_afmethod_iterations;       /* -a N: lpc.c:578-633 on the device (lnn_k_af.h) */
    if (parameter->num_channels > LINNE_MAX_NUM_CHANNELS) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    memset(&h, 0, sizeof(h));
    h.num_channels = parameter->num_channels; h.sampling_rate = parameter->sampling_rate; h.bits_per_sample = parameter->bits_per_sample;
    h.num_samples_per_block = parameter->num_samples_per_block; h.preset = parameter->preset; h.ch_process_method = parameter->ch_process_method;
    encoder->header = h; encoder->shape = shape; encoder->layers = ly;
    encoder->set_parameter = 1;
    return LINNE_APIRESULT_OK;
}

static LINNEApiResult encoder_device(struct LINNEEncoder *enc, int all)
{
    uint32_t d;
    if (open_gpus(&enc->gpus, "LINNEEncoder", all) != LNN_OK) return LINNE_APIRESULT_NG;
    for (d = 0; d < LNN_MAX_DEVICES; d++) if (enc->gpus.ctx[d]) { (void)LINNEAmd_SetAfIterations(enc->gpus.ctx[d], enc->af_iters); (void)LINNEAmd_SetLearning(enc->gpus.ctx[d], enc->learning); }
    return LINNE_APIRESULT_OK;
}
static void report(const struct LINNEAmdContext *ctx, const char *what, int ret)
{
    fprintf(stderr, "liblinne_amd: %s failed (%d): %s\n", what, ret, LINNEAmd_GetLastError(ctx));
}

LINNEApiResult LINNEEncoder_EncodeBlock(struct LINNEEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
        uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    uint32_t ch, S, C;
    int ret;
    if (encoder == NULL || input == NULL || num_samples == 0 || data == NULL || data_size == 0 || output_size == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (encoder->set_parameter != 1) return LINNE_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > encoder->header.num_samples_per_block) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    if (encoder_device(encoder, 0) != LINNE_APIRESULT_OK) return LINNE_APIRESULT_NG;
    S = encoder->shape.num_samples_per_block; C = encoder->shape.num_channels;
    for (ch = 0; ch < C; ch++) {
        if (input[ch] == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
        memcpy(encoder->pcm + (size_t)ch * S, input[ch], sizeof(int32_t) * num_samples);
        if (num_samples < S) memset(encoder->pcm + (size_t)ch * S + num_samples, 0, sizeof(int32_t) * (S - num_samples));
    }
    ret = LINNEAmd_EncodeFramesHost(encoder->gpus.ctx[0], &encoder->shape, encoder->pcm, &num_samples, 1, encoder->residual, encoder->params, encoder->stats);
    if (ret != LNN_OK) { report(encoder->gpus.ctx[0], "EncodeFramesHost", ret); return (LINNEApiResult)ret; }
    ret = LINNEAmd_PackFrames(&encoder->shape, encoder->pcm, &num_samples, 1, encoder->residual, encoder->params, encoder->stats,
            data, data_size, output_size, &encoder->parcor_state, 1);
    return (LINNEApiResult)ret;
}

/* whole stream (linne_encoder.c:865-932): groups of frames rotate over LNN_SLOTS staging slots -- while the GPU analyses
 * one group, the host threads pack the previous one into the stream and fill the next */
struct fill_job { const int32_t *const *input; int32_t *pcm; int16_t *pcm16; uint32_t width; uint32_t *nsm; uint32_t C, S, num_samples, base; };
static void fill_frames(void *arg, uint32_t first, uint32_t count)
{
    const struct fill_job *j = arg;
    uint32_t f, ch, s;
    for (f = first; f < first + count; f++) {
        const uint64_t start = (uint64_t)(j->base + f) * j->S;
        const uint32_t n = (j->num_samples - start < j->S) ? (uint32_t)(j->num_samples - start) : j->S;
        j->nsm[f] = n;
        for (ch = 0; ch < j->C; ch++) {
            const int32_t *src = j->input[ch] + start;
            if (j->pcm16 && j->width == 3) {    /* 17 .. 24 bits per sample: packed 3-byte samples (tools/linne_codec right-justifies them into int32, linne_codec.c:101-105) */
                uint8_t *dst = (uint8_t *)j->pcm16 + 3u * (((size_t)f * j->C + ch) * j->S);
                for (s = 0; s < n; s++) { const uint32_t v = (uint32_t)src[s]; dst[3u * s] = (uint8_t)v; dst[3u * s + 1u] = (uint8_t)(v >> 8); dst[3u * s + 2u] = (uint8_t)(v >> 16); }
                if (n < j->S) memset(dst + 3u * n, 0, 3u * (j->S - n));
            } else if (j->pcm16) {      /* <= 16 bits per sample: half the bytes over PCIe, widened again by k_prep */
                int16_t *dst = j->pcm16 + ((size_t)f * j->C + ch) * j->S;
                for (s = 0; s < n; s++) dst[s] = (int16_t)src[s];
                if (n < j->S) memset(dst + n, 0, sizeof(int16_t) * (j->S - n));
            } else {
                int32_t *dst = j->pcm + ((size_t)f * j->C + ch) * j->S;
                memcpy(dst, src, sizeof(int32_t) * n);
                if (n < j->S) memset(dst + n, 0, sizeof(int32_t) * (j->S - n));
            }
        }
    }
}
static int fetch_from_slot(void *arg, uint32_t frame, int32_t *dst) { return LINNEAmd_SlotFetchResidual((struct LINNEAmdSlot *)arg, frame, dst); }

LINNEApiResult LINNEEncoder_EncodeWhole(struct LINNEEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
        uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    LINNEApiResult r;
    uint32_t S, C, F, f, ch, group, ngroups, nslots, window, ndev, submitted = 0, packed = 0;
    struct lnn_gpus *gp;
    uint64_t off = LINNE_HEADER_SIZE;
    uint32_t *nsm = NULL, *sizes = NULL, *gbase = NULL;
    const uint32_t threads = default_threads();
    double t_begin = 0, t_setup = 0, t_fill = 0, t_submit = 0, t_wait = 0, t_pack = 0, t0;
    int ret = LNN_OK;
    if (encoder == NULL || input == NULL || data == NULL || output_size == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (encoder->set_parameter != 1) return LINNE_APIRESULT_PARAMETER_NOT_SET;
    encoder->header.num_samples = num_samples;
    if ((r = LINNEEncoder_EncodeHeader(&encoder->header, data, data_size)) != LINNE_APIRESULT_OK) return r;
    S = encoder->shape.num_samples_per_block; C = encoder->shape.num_channels;
    for (ch = 0; ch < C; ch++) if (input[ch] == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (encoder_device(encoder, 1) != LINNE_APIRESULT_OK) return LINNE_APIRESULT_NG;
    gp = &encoder->gpus; ndev = gp->ndev;
    t_begin = now_s();
    F = (uint32_t)(((uint64_t)num_samples + S - 1) / S);
    group = default_group((F + ndev - 1) / ndev, &encoder->shape, &encoder->layers, 1); if (group > F) group = F;
    /* The schedule of groups.  A long stream starts and ends with a half-size group per device: the GPU has work after half
     * the fill time, and what the host still has to stitch when the GPU is through is half a group (LINNE_AMD_RAMP=n: groups
     * of 1/n there; 0: equal groups only.  Measured on the 60-minute stream: 120 ms with 2, 123 with 4, 124 with 0 -- the
     * pipeline is bound by the GPU's work per group, 7.2 us per frame with the Rice kernels beside the analysis). */
    {
        const char *e_ = getenv("LINNE_AMD_RAMP");
        const uint32_t div_ = (e_ && atoi(e_) > 0) ? (uint32_t)atoi(e_) : 2u;
        const uint32_t small = group / div_, ramp = (e_ && atoi(e_) == 0) ? 0u : ((small >= 256u && (uint64_t)F >= 3ull * group * ndev) ? ndev : 0u);
        const uint32_t mid_frames = F - 2u * ramp * small, nmid = (mid_frames + group - 1) / group;
        uint32_t g_ = 0, pos = 0, left = mid_frames;
        ngroups = 2u * ramp + nmid;
        gbase = malloc(sizeof(uint32_t) * (ngroups + 1u));
        if (!gbase) return LINNE_APIRESULT_NG;
        for (f = 0; f < ramp; f++) { gbase[g_++] = pos; pos += small; }
        for (f = 0; f < nmid; f++) { const uint32_t take = (left + (nmid - f) - 1u) / (nmid - f); gbase[g_++] = pos; pos += take; left -= take; }
        for (f = 0; f < ramp; f++) { gbase[g_++] = pos; pos += small; }
        gbase[g_] = pos;        /* = F */
    }
    /* group g goes to device g mod ndev, slot (g / ndev) mod nslots of that device: ndev * nslots groups in flight */
    nslots = (ngroups + ndev - 1) / ndev; if (nslots > LNN_ENC_SLOTS) nslots = LNN_ENC_SLOTS;
    window = ndev * nslots;
    if (F > 32) for (f = 0; f < ndev; f++) (void)LINNEAmd_ReserveScratch(gp->ctx[f], LINNEAmd_ScratchBytesPerFrame(&encoder->shape) * group + (1ull << 20));     /* a group = one launch chunk */
    if ((ret = want_slots(gp, &encoder->shape, group, nslots, 1)) != 0) {
        report(gp->ctx[ret - 1], "SlotCreate", LNN_NG); free(gbase); return LINNE_APIRESULT_NG;
    }
    nsm = malloc(sizeof(uint32_t) * (size_t)group * window); sizes = malloc(sizeof(uint32_t) * group);
    if (!nsm || !sizes) { ret = LNN_NG; goto done; }
    t_setup = now_s() - t_begin;
    while (packed < ngroups) {
        while (submitted < ngroups && submitted - packed < window) {
            struct LINNEAmdSlot *sl = gp->slot[submitted % ndev][(submitted / ndev) % nslots];
            struct fill_job fj;
            const uint32_t base = gbase[submitted], cnt = gbase[submitted + 1] - base;
            fj.input = input; fj.pcm = LINNEAmd_SlotPcm(sl); fj.pcm16 = LINNEAmd_SlotPcm16(sl); fj.width = LINNEAmd_SlotPcmWidth(sl); fj.nsm = nsm + (size_t)(submitted % window) * group;
            fj.C = C; fj.S = S; fj.num_samples = num_samples; fj.base = base;
            t0 = now_s();
            lnn_parallel_for(cnt, threads, fill_frames, &fj);
            t_fill += now_s() - t0; t0 = now_s();
            ret = LINNEAmd_SlotEncodeSubmit(sl, fj.nsm, cnt);
            if (trace_on() > 1) fprintf(stderr, "liblinne_amd:   group %u submitted at %.1f ms (the call took %.1f ms)\n", submitted, (now_s() - t_begin) * 1e3, (now_s() - t0) * 1e3);
            t_submit += now_s() - t0;
            if (ret != LNN_OK) { report(gp->ctx[submitted % ndev], "SlotEncodeSubmit", ret); goto done; }
            submitted++;
        }
        {
            struct LINNEAmdSlot *sl = gp->slot[packed % ndev][(packed / ndev) % nslots];
            const uint32_t base = gbase[packed], cnt = gbase[packed + 1] - base;
            t0 = now_s();
            ret = LINNEAmd_SlotWait(sl);
            if (trace_on() > 1) fprintf(stderr, "liblinne_amd:   group %u (%u frames): waited %.1f ms, ready at %.1f ms\n", packed, cnt, (now_s() - t0) * 1e3, (now_s() - t_begin) * 1e3);
            t_wait += now_s() - t0; t0 = now_s();
            if (ret != LNN_OK) { report(gp->ctx[packed % ndev], "SlotWait", ret); goto done; }
            if (LINNEAmd_SlotFlags(sl) & LINNE_AMD_SLOT_EMIT)       /* the device wrote the Rice codes: stitch */
                ret = LINNEAmd_PackFramesEmitted(&encoder->shape, input, (uint64_t)base * S, nsm + (size_t)(packed % window) * group, cnt,
                        LINNEAmd_SlotParams(sl), LINNEAmd_SlotStats(sl), LINNEAmd_SlotRicePlan(sl), LINNEAmd_SlotPacked(sl), LINNEAmd_SlotOffsets(sl),
                        fetch_from_slot, sl, data + off, data_size - off, sizes, &encoder->parcor_state, threads);
            else
                ret = LINNEAmd_PackFramesPlanned(&encoder->shape, LINNEAmd_SlotPcm(sl), nsm + (size_t)(packed % window) * group, cnt,
                        LINNEAmd_SlotData(sl), LINNEAmd_SlotParams(sl), LINNEAmd_SlotStats(sl), LINNEAmd_SlotRicePlan(sl),
                        data + off, data_size - off, sizes, &encoder->parcor_state, threads);
            t_pack += now_s() - t0;
            if (ret != LNN_OK) goto done;
            for (f = 0; f < cnt; f++) off += sizes[f];
            packed++;
        }
    }
    *output_size = (uint32_t)off;
    if (trace_on()) fprintf(stderr, "liblinne_amd: EncodeWhole %u frames, %u GPU(s), %u threads: setup %.1f ms, fill %.1f, submit %.1f, wait %.1f, pack %.1f, total %.1f ms\n",
            F, ndev, threads, t_setup * 1e3, t_fill * 1e3, t_submit * 1e3, t_wait * 1e3, t_pack * 1e3, (now_s() - t_begin) * 1e3);
done:
    for (ch = 0; ch < LNN_MAX_DEVICES; ch++) for (f = 0; f < LNN_SLOTS; f++) if (encoder->gpus.slot[ch][f]) (void)LINNEAmd_SlotWait(encoder->gpus.slot[ch][f]);
    free(nsm); free(sizes); free(gbase);
    return (LINNEApiResult)ret;
}

/* ================================================================================================ decoder */
struct LINNEDecoder {
    struct LINNEHeader header;
    uint32_t max_num_channels, max_num_layers, max_num_parameters_per_layer;
    uint8_t set_header, check_crc, alloced_by_own;
    void *work;
    struct LINNEAmdShape shape;
    struct lnn_layers layers;
    struct lnn_gpus gpus;
    int32_t *samples; uint64_t samples_cap;     /* one-frame staging (heap: the block size is unknown at Create) */
    int32_t *params;                            /* inside the work area */
};

LINNEApiResult LINNEDecoder_DecodeHeader(const uint8_t *data, uint32_t data_size, struct LINNEHeader *header)
{
    struct LINNEHeader h;
    if (data == NULL || header == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (data_size < LINNE_HEADER_SIZE) return LINNE_APIRESULT_INSUFFICIENT_DATA;
    if (memcmp(data, "IBRA", 4) != 0) return LINNE_APIRESULT_INVALID_FORMAT;
    memset(&h, 0, sizeof(h));
    h.format_version = get_be32(data + 4); h.codec_version = get_be32(data + 8);
    h.num_channels = (uint16_t)get_be16(data + 12); h.num_samples = get_be32(data + 14); h.sampling_rate = get_be32(data + 18);
    h.bits_per_sample = (uint16_t)get_be16(data + 22); h.num_samples_per_block = get_be32(data + 24);
    h.preset = data[28]; h.ch_process_method = (LINNEChannelProcessMethod)data[29];
    *header = h;
    return LINNE_APIRESULT_OK;
}

static int decoder_config_ok(const struct LINNEDecoderConfig *c)
{
    return c != NULL && c->max_num_channels != 0 && c->max_num_layers != 0 && c->max_num_parameters_per_layer != 0;
}

int32_t LINNEDecoder_CalculateWorkSize(const struct LINNEDecoderConfig *config)
{
    uint64_t sz;
    if (!decoder_config_ok(config)) return -1;
    sz = sizeof(struct LINNEDecoder) + LNN_ALIGN;
    sz += (uint64_t)config->max_num_channels * LINNE_AMD_PARAM_WORDS * sizeof(int32_t) + LNN_ALIGN;
    if (sz > 0x7FFFFFFFull) return -1;
    return (int32_t)sz;
}

struct LINNEDecoder *LINNEDecoder_Create(const struct LINNEDecoderConfig *config, void *work, int32_t work_size)
{
    struct LINNEDecoder *dec;
    uint8_t own = 0, *p;
    if (work == NULL && work_size == 0) {
        if ((work_size = LINNEDecoder_CalculateWorkSize(config)) < 0) return NULL;
        work = malloc((size_t)work_size);
        own = 1;
    }
    if (config == NULL || work == NULL || !decoder_config_ok(config) || work_size < LINNEDecoder_CalculateWorkSize(config)) {
        if (own) free(work);
        return NULL;
    }
    p = (uint8_t *)ALIGN_UP((uintptr_t)work);
    dec = (struct LINNEDecoder *)p; p += sizeof(*dec);
    memset(dec, 0, sizeof(*dec));
    dec->work = work; dec->alloced_by_own = own;
    dec->max_num_channels = config->max_num_channels; dec->max_num_layers = config->max_num_layers;
    dec->max_num_parameters_per_layer = config->max_num_parameters_per_layer;
    dec->check_crc = (config->check_crc == 1);
    p = (uint8_t *)ALIGN_UP((uintptr_t)p); dec->params = (int32_t *)p;
    lnn_tables_init();
    return dec;
}

void LINNEDecoder_Destroy(struct LINNEDecoder *decoder)
{
    if (decoder == NULL) return;
    close_gpus(&decoder->gpus);
    free(decoder->samples); decoder->samples = NULL;
    if (decoder->alloced_by_own) free(decoder->work);
}

LINNEApiResult LINNEDecoder_SetHeader(struct LINNEDecoder *decoder, const struct LINNEHeader *header)
{
    struct LINNEAmdShape shape;
    struct lnn_layers ly;
    uint32_t l;
    if (decoder == NULL || header == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (header->format_version != LINNE_FORMAT_VERSION || header->codec_version != LINNE_CODEC_VERSION || !header_fields_ok(header)) return LINNE_APIRESULT_INVALID_FORMAT;
    if (decoder->max_num_channels < header->num_channels) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    shape.num_channels = header->num_channels; shape.bits_per_sample = header->bits_per_sample;
    shape.num_samples_per_block = header->num_samples_per_block; shape.preset = header->preset;
    shape.ch_process_method = (uint32_t)header->ch_process_method;
    if (lnn_shape_layers(&shape, &ly) != 0) return LINNE_APIRESULT_INVALID_FORMAT;
    if (decoder->max_num_layers < ly.num_layers) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    for (l = 0; l < ly.num_layers; l++) if (decoder->max_num_parameters_per_layer < ly.size[l]) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    if (header->num_channels > LINNE_MAX_NUM_CHANNELS) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    decoder->header = *header; decoder->shape = shape; decoder->layers = ly;
    decoder->set_header = 1;
    return LINNE_APIRESULT_OK;
}

static LINNEApiResult decoder_device(struct LINNEDecoder *dec, int all)
{
    return open_gpus(&dec->gpus, "LINNEDecoder", all) == LNN_OK ? LINNE_APIRESULT_OK : LINNE_APIRESULT_NG;
}

LINNEApiResult LINNEDecoder_DecodeBlock(struct LINNEDecoder *decoder, const uint8_t *data, uint32_t data_size,
        int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples,
        uint32_t *decode_size, uint32_t *num_decode_samples)
{
    uint32_t type = 0, n = 0, consumed = 0, ch, C, S;
    uint64_t need;
    int ret;
    if (decoder == NULL || data == NULL || buffer == NULL || decode_size == NULL || num_decode_samples == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (!decoder->set_header) return LINNE_APIRESULT_PARAMETER_NOT_SET;
    if (buffer_num_channels < decoder->header.num_channels) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    C = decoder->shape.num_channels; S = decoder->shape.num_samples_per_block;
    need = (uint64_t)C * S;
    if (decoder->samples_cap < need) {
        free(decoder->samples);
        decoder->samples = malloc(sizeof(int32_t) * need);
        decoder->samples_cap = decoder->samples ? need : 0;
        if (!decoder->samples) return LINNE_APIRESULT_NG;
    }
    ret = lnn_parse_block(&decoder->shape, &decoder->layers, data, data_size, decoder->check_crc, buffer_num_samples,
            &type, &n, &consumed, decoder->samples, decoder->params);
    if (ret != LNN_OK) return (LINNEApiResult)ret;
    if (type == LNN_BLOCK_COMPRESS) {
        if (decoder_device(decoder, 0) != LINNE_APIRESULT_OK) return LINNE_APIRESULT_NG;
        ret = LINNEAmd_DecodeFramesHost(decoder->gpus.ctx[0], &decoder->shape, decoder->samples, &n, 1, decoder->params);
        if (ret != LNN_OK) { report(decoder->gpus.ctx[0], "DecodeFramesHost", ret); return (LINNEApiResult)ret; }
    }
    for (ch = 0; ch < C; ch++) memcpy(buffer[ch], decoder->samples + (size_t)ch * S, sizeof(int32_t) * n);
    *decode_size = consumed;
    *num_decode_samples = n;
    return LINNE_APIRESULT_OK;
}

/* ---- whole stream (linne_decoder.c:671-742): scan block boundaries from the size fields, entropy-decode a group of
 * blocks on the host threads straight into a staging slot, synthesise it on the GPU while the next group is being
 * entropy-decoded, then scatter the PCM into the caller's planes ---- */
struct dgroup {
    uint32_t nblk, ncomp; int streamed; uint64_t seg_first, seg_bytes;
    uint64_t *offs, *avail; uint32_t *room, *types, *ns, *prog, *cidx, *cn, *cons, *bsz /* the block's bytes by its size field (the scan) */; int *rets;
};
struct unpack_job {
    const struct LINNEDecoder *dec; const uint8_t *data; struct dgroup *g; int32_t *sdata, *sprm; int32_t **buffer;
    /* stream mode (the device decodes the Rice codes): the group's bytes go to the slot as they are, from seg_first on */
    uint8_t *sstream; uint64_t *sbitpos, *sbitend; uint64_t seg_first, seg_bytes; const int16_t *s16; uint32_t swidth;
};
static void copy_to_staging(uint8_t *dst, const uint8_t *src, size_t n);
static void unpack_blocks(void *arg, uint32_t first, uint32_t count)
{
    struct unpack_job *j = arg;
    struct dgroup *g = j->g;
    const struct LINNEAmdShape *sh = &j->dec->shape;
    const uint32_t C = sh->num_channels, S = sh->num_samples_per_block;
    const uint64_t CS = (uint64_t)C * S;
    int32_t *tmp = NULL, tprm[LINNE_MAX_NUM_CHANNELS * LINNE_AMD_PARAM_WORDS];
    uint32_t f, ch;
    for (f = first; f < first + count; f++) {
        if (g->cidx[f] != 0xFFFFFFFFu && j->sstream) {      /* stream mode: header, CRC and parameters here, the Rice code on the device */
            uint64_t rbit = 0;
            /* the block's bytes go to the slot's pinned stream buffer by the thread that checks its CRC next: the second reader
             * finds them in its cache (a pass of its own over the group cost a third of the parsing: 0.48 GB more from memory) */
            copy_to_staging(j->sstream + (g->offs[f] - j->seg_first), j->data + g->offs[f], g->bsz[f]);
            g->rets[f] = lnn_parse_block_head(sh, &j->dec->layers, j->data + g->offs[f], g->avail[f], j->dec->check_crc, g->room[f],
                    &g->types[f], &g->ns[f], &g->cons[f], NULL, j->sprm + (size_t)g->cidx[f] * C * LINNE_AMD_PARAM_WORDS, &rbit);
            j->sbitpos[g->cidx[f]] = (g->offs[f] - j->seg_first) * 8u + rbit;
            j->sbitend[g->cidx[f]] = (g->offs[f] - j->seg_first + g->cons[f]) * 8u;
            continue;
        }
        if (g->cidx[f] != 0xFFFFFFFFu) {            /* COMPRESS by its header: residual and parameters go to the slot */
            g->rets[f] = lnn_parse_block(sh, &j->dec->layers, j->data + g->offs[f], g->avail[f], j->dec->check_crc, g->room[f],
                    &g->types[f], &g->ns[f], &g->cons[f], j->sdata + g->cidx[f] * CS, j->sprm + (size_t)g->cidx[f] * C * LINNE_AMD_PARAM_WORDS);
            continue;
        }
        if (!tmp && !(tmp = malloc(sizeof(int32_t) * CS))) { g->rets[f] = LNN_NG; continue; }
        g->rets[f] = lnn_parse_block(sh, &j->dec->layers, j->data + g->offs[f], g->avail[f], j->dec->check_crc, g->room[f],
                &g->types[f], &g->ns[f], &g->cons[f], tmp, tprm);
        if (g->rets[f] == LNN_OK)                   /* RAW / SILENT carry PCM: straight to the caller's planes */
            for (ch = 0; ch < C; ch++) memcpy(j->buffer[ch] + g->prog[f], tmp + (size_t)ch * S, sizeof(int32_t) * g->ns[f]);
    }
    free(tmp);
}
/* int16 -> int32 into the caller's planes with streaming stores: the destination (1.27 GB per 60-minute stereo stream) is written
 * once and not read here, so fetching its lines for ownership first would be half again the memory traffic of the scatter */
#include <immintrin.h>
__attribute__((target("avx2"))) static void widen16_avx2(int32_t *dst, const int16_t *src, uint32_t n)
{
    uint32_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 31u)) { dst[i] = src[i]; i++; }
    for (; i + 16u <= n; i += 16u) {
        const __m256i a = _mm256_cvtepi16_epi32(_mm_loadu_si128((const __m128i *)(src + i)));
        const __m256i b = _mm256_cvtepi16_epi32(_mm_loadu_si128((const __m128i *)(src + i + 8u)));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 8u), b);
    }
    for (; i < n; i++) dst[i] = src[i];
    _mm_sfence();
}
/* bytes into a pinned staging buffer that only the DMA engine will read: streaming stores, no read for ownership of the destination */
__attribute__((target("avx2"))) static void copy_nt_avx2(uint8_t *dst, const uint8_t *src, size_t n)
{
    size_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 31u)) { dst[i] = src[i]; i++; }
    for (; i + 128u <= n; i += 128u) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32u));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64u)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96u));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32u), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64u), c); _mm256_stream_si256((__m256i *)(dst + i + 96u), d);
    }
    for (; i < n; i++) dst[i] = src[i];
    _mm_sfence();
}
static void copy_to_staging(uint8_t *dst, const uint8_t *src, size_t n)
{
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("avx2") ? 1 : 0;
    if (have && n >= 4096u) copy_nt_avx2(dst, src, n); else memcpy(dst, src, n);
}
static void widen16(int32_t *dst, const int16_t *src, uint32_t n)
{
    static int have = -1;
    uint32_t i;
    if (have < 0) have = __builtin_cpu_supports("avx2") ? 1 : 0;
    if (have && n >= 64u) { widen16_avx2(dst, src, n); return; }
    for (i = 0; i < n; i++) dst[i] = src[i];
}
static void scatter_blocks(void *arg, uint32_t first, uint32_t count)
{
    struct unpack_job *j = arg;
    struct dgroup *g = j->g;
    const uint32_t C = j->dec->shape.num_channels, S = j->dec->shape.num_samples_per_block;
    uint32_t f, ch;
    for (f = first; f < first + count; f++) {
        if (g->types[f] != LNN_BLOCK_COMPRESS) continue;
        for (ch = 0; ch < C; ch++) {
            if (j->s16 && j->swidth == 3) {   /* the PCM came back as packed 3-byte samples: widen (sign-extend) on the way into the caller's planes */
                const uint8_t *src = (const uint8_t *)j->s16 + 3u * (((size_t)g->cidx[f] * C + ch) * S);
                int32_t *dst = j->buffer[ch] + g->prog[f];
                uint32_t s_;
                for (s_ = 0; s_ < g->ns[f]; s_++) dst[s_] = (int32_t)((uint32_t)src[3u * s_] | ((uint32_t)src[3u * s_ + 1u] << 8)) | ((int32_t)(int8_t)src[3u * s_ + 2u] << 16);
            } else if (j->s16) {        /* the PCM came back as int16: widen on the way into the caller's planes */
                const int16_t *src = j->s16 + ((size_t)g->cidx[f] * C + ch) * S;
                widen16(j->buffer[ch] + g->prog[f], src, g->ns[f]);
            } else memcpy(j->buffer[ch] + g->prog[f], j->sdata + ((size_t)g->cidx[f] * C + ch) * S, sizeof(int32_t) * g->ns[f]);
        }
    }
}
static int dgroup_alloc(struct dgroup *g, uint32_t n)
{
    memset(g, 0, sizeof(*g));
    g->offs = malloc(sizeof(*g->offs) * n); g->avail = malloc(sizeof(*g->avail) * n); g->room = malloc(sizeof(uint32_t) * n);
    g->types = malloc(sizeof(uint32_t) * n); g->ns = malloc(sizeof(uint32_t) * n); g->prog = malloc(sizeof(uint32_t) * n);
    g->cidx = malloc(sizeof(uint32_t) * n); g->cn = malloc(sizeof(uint32_t) * n); g->rets = malloc(sizeof(int) * n); g->cons = malloc(sizeof(uint32_t) * n); g->bsz = malloc(sizeof(uint32_t) * n);
    return (g->offs && g->avail && g->room && g->types && g->ns && g->prog && g->cidx && g->cn && g->rets && g->cons && g->bsz) ? 0 : -1;
}
static void dgroup_free(struct dgroup *g)
{
    free(g->offs); free(g->avail); free(g->room); free(g->types); free(g->ns); free(g->prog); free(g->cidx); free(g->cn); free(g->rets); free(g->cons); free(g->bsz);
}

static __thread uint32_t g_last_decode_mode;
/* how this thread's last LINNEDecoder_DecodeWhole ran: bit 0 = it finished with the device decoding the Rice codes, bit 1 = it
 * had started that way and went over everything again with the host's Rice decoder */
uint32_t LINNEAmd_LastDecodeWholeMode(void) { return g_last_decode_mode; }

LINNEApiResult LINNEDecoder_DecodeWhole(struct LINNEDecoder *decoder, const uint8_t *data, uint32_t data_size,
        int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples)
{
    LINNEApiResult r;
    struct LINNEHeader h;
    const struct LINNEHeader *hd;
    struct dgroup grp[LNN_MAX_DEVICES * LNN_SLOTS];
    struct unpack_job uj;
    struct lnn_gpus *gp = &decoder->gpus;
    uint32_t group, f, produced = 0, consumed_groups = 0, progress = 0, i, ngalloc = 0, ndev = 1, window = LNN_SLOTS, nslots = LNN_SLOTS;
    const uint32_t threads = default_threads();
    uint64_t off;
    double t_begin = now_s(), t_parse = 0, t_submit = 0, t_wait = 0, t_scatter = 0, t0;
    int ret = LNN_OK, scanning = 1;
    /* stream mode (the default; LINNE_AMD_DECODE_STREAM=0 turns it off): the device decodes the Rice codes
     * (LINNEAmd_SlotDecodeStreamSubmit), the host threads only scan the block headers, check the CRCs and decode the parameters.
     * Only for streams whose CRCs are checked: a block that passes is what an encoder wrote.  Should the device still meet
     * something no encoder writes, the whole call starts over with the host's Rice decoder (stream_mode = 0, and the host path's
     * group size and slots), which restates the reference's; PCM beyond the 16-bit range only makes that group come back as int32.  60-minute stream, one GPU, 16 host threads: 62 ms against 94 with the
     * host's Rice decoder (profiles/r02_decode_stream.txt). */
    int stream_mode, stream_auto;
    { const char *e_ = getenv("LINNE_AMD_DECODE_STREAM"); stream_mode = (e_ ? atoi(e_) != 0 : 1); stream_auto = (e_ == NULL); }
    if (decoder == NULL || data == NULL || buffer == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if ((r = LINNEDecoder_DecodeHeader(data, data_size, &h)) != LINNE_APIRESULT_OK) return r;
    if ((r = LINNEDecoder_SetHeader(decoder, &h)) != LINNE_APIRESULT_OK) return r;
    hd = &decoder->header;
    if (buffer_num_channels < hd->num_channels || buffer_num_samples < hd->num_samples) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    for (i = 0; i < hd->num_channels; i++) if (buffer[i] == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (!decoder->check_crc) stream_mode = 0;
    g_last_decode_mode = 0;
setup:      /* (again after the device's Rice decoder refused something: the host path gets the host path's group size and slots) */
    for (i = 0; i < ngalloc; i++) dgroup_free(&grp[i]);
    ngalloc = 0;
    {
        const uint32_t S = decoder->shape.num_samples_per_block;
        const uint32_t F = (uint32_t)(((uint64_t)hd->num_samples + S - 1) / S);
        /* group g of blocks goes to device g mod ndev, slot (g / ndev) mod LNN_SLOTS of that device */
        if (gp->ndev == 0) { gp->ndev = lnn_parse_device_list(getenv("LINNE_AMD_DEVICES"), gp->device, LNN_MAX_DEVICES); if (gp->ndev == 0) { const char *e = getenv("LINNE_AMD_DEVICE"); gp->device[0] = e ? atoi(e) : 0; gp->ndev = 1; } }
        ndev = gp->ndev; window = ndev * LNN_SLOTS; nslots = LNN_SLOTS;
        group = default_group((F + ndev - 1) / ndev, &decoder->shape, &decoder->layers, 0); if (group > F) group = F ? F : 1;
        /* Short streams: the device's Rice decoder walks a block's channels one after the other, 0.29 us per sample -- 6 ms for a
         * stereo block of 10 240 samples, 24-28 ms for eight channels, whether the launch holds one block or thirty thousand -- while the
         * host threads decode 150 samples per microsecond each.  Measured on stereo 44.1 kHz with 16 threads (tools/e2e.py,
         * LINNE_AMD_DECODE_STREAM=1 / 0): 6 s 7.0 / 1.2 ms, 30 s 7.6 / 2.2, 3 min 9.8 / 7.7, 6 min 11.2 / 13.1, 10 min 12.5 / 20.6:
         * they cross near 4.3 minutes.  Unless the environment says which, the call takes the way this estimate makes shorter */
        if (stream_mode && stream_auto) {
            const double smp = (double)S * (double)hd->num_channels, all = smp * (double)F;
            const double t_host = 1.0 + all * ((uint64_t)F * hd->num_channels >= 1024u ? 5.4e-6 : 6.7e-6) / (double)threads;      /* ms; (three groups from 1024 channel-frames on, below: their decoding overlaps the GPU's work) */
            const double t_stream = 1.0 + 0.29e-3 * smp + 1.05e-7 * all;
            if (t_host < t_stream) stream_mode = 0;
            stream_auto = 0;
        }
        /* the host's way with a stream that would be ONE group: three, so that the threads decode the second's codes while the GPU
         * reconstructs the first (3 minutes of stereo: 7.9 -> 6.1 ms; LINNE_AMD_GROUP says otherwise) */
        if (!stream_mode && !getenv("LINNE_AMD_GROUP") && group >= F && (uint64_t)F * hd->num_channels >= 1024u) group = (F + 2u) / 3u;
    }
    if (stream_mode) {
        /* The device's Rice decoder is serial per block: a launch takes ~9 ms for 300 blocks as for 30 000 -- a LATENCY, not a cost:
         * the groups' launches run side by side on the slots' own streams.  What bounds the call is the host's work (parsing 10 ms +
         * widening the PCM into the caller's planes 10 ms per 60-minute stream on 16 threads) and the PCIe link (1.1 GB at 57 GB/s,
         * both directions share it: tools/pcie_bw.py), so the stream goes in EIGHT groups per device (LINNE_AMD_DECODE_GROUPS), every
         * one with a slot of its own: all are parsed and submitted back to back, then widened in order while the later ones are still
         * on the link or in the decoder */
        const char *eg = getenv("LINNE_AMD_DECODE_GROUPS");
        const uint32_t S_ = decoder->shape.num_samples_per_block, F_ = (uint32_t)(((uint64_t)hd->num_samples + S_ - 1) / S_);
        /* eight groups for mono / stereo, four for more channels: a block's channels are ONE serial code, so the decoder's latency per
         * launch grows with them and a group's launch should not outlast the host's work on the others (8 channels, 24 bits, 96 kHz,
         * 10 minutes: 80 ms with two groups, 67 with four, 72 with six or eight; 4 channels: 34.5 with four, 37.8 with eight) */
        const uint32_t ng_auto = decoder->shape.num_channels <= 2u ? 8u : 4u;
        const uint32_t ng = (eg && atoi(eg) > 0) ? (uint32_t)atoi(eg) : ng_auto;
        const uint64_t cap = (1ull << 30) / ((uint64_t)decoder->shape.num_channels * S_ * sizeof(int32_t)) + 1;
        /* a group keeps the synthesis in its throughput form (from 1536 channel-frames on, lnn_device.hip) */
        const uint32_t floor_ = (1536u + decoder->shape.num_channels - 1u) / decoder->shape.num_channels < 256u ? 256u : (1536u + decoder->shape.num_channels - 1u) / decoder->shape.num_channels;
        uint32_t g_ = (F_ + ng * ndev - 1) / (ng * ndev);
        if (g_ < floor_) g_ = floor_;
        if (g_ > cap) g_ = (uint32_t)cap;
        if (g_ > F_) g_ = F_ ? F_ : 1u;
        if (!getenv("LINNE_AMD_GROUP")) group = g_;
        nslots = (((F_ + group - 1) / group) + ndev - 1) / ndev;          /* slots per device: as many as it will see groups */
        if (nslots < 1) nslots = 1;
        if (nslots > LNN_SLOTS) nslots = LNN_SLOTS;
        window = ndev * nslots;
    }
    for (ngalloc = 0; ngalloc < window; ngalloc++) if (dgroup_alloc(&grp[ngalloc], group) != 0) { ngalloc++; ret = LNN_NG; goto done; }
    memset(&uj, 0, sizeof(uj));
    uj.dec = decoder; uj.data = data; uj.buffer = buffer;
    off = LINNE_HEADER_SIZE; produced = 0; consumed_groups = 0; progress = 0; scanning = 1; ret = LNN_OK;
    while (scanning || consumed_groups < produced) {
        /* parsing and submitting come first: what a group waits for longest is the device's Rice decoder (a latency of ~9 ms however
         * small the group), so every group should be on its way before the host turns to widening the ones that are back */
        if (scanning && produced - consumed_groups < window && progress < hd->num_samples && off < data_size) {
            struct dgroup *g = &grp[produced % window];
            struct LINNEAmdSlot *sl;
            uint32_t scan_progress = progress, ncomp = 0;
            g->nblk = 0;
            while (g->nblk < group && scan_progress < hd->num_samples && off < data_size) {
                const uint64_t rem = data_size - off;
                const uint32_t k = g->nblk++;
                uint32_t bsize;
                g->offs[k] = off; g->avail[k] = rem; g->room[k] = buffer_num_samples - scan_progress; g->prog[k] = scan_progress;
                g->cidx[k] = 0xFFFFFFFFu; g->types[k] = 0xFFFFFFFFu; g->ns[k] = 0; g->bsz[k] = 0;
                if (rem < 11 || get_be16(data + off) != 0xFFFF) break;                  /* the parser reports the error */
                bsize = get_be32(data + off + 2);
                if ((uint64_t)bsize + 6 > rem) break;
                g->bsz[k] = bsize + 6u;
                if (data[off + 8] == LNN_BLOCK_COMPRESS) g->cidx[k] = ncomp++;
                scan_progress += get_be16(data + off + 9);
                off += (uint64_t)bsize + 6;
            }
            if (ncomp) {
                int wret;
                if (decoder_device(decoder, 1) != LINNE_APIRESULT_OK) { ret = LNN_NG; goto done; }
                if ((wret = want_slots(gp, &decoder->shape, group, nslots, stream_mode ? 2 : 0)) != 0) {
                    report(gp->ctx[wret - 1], "SlotCreate", LNN_NG); ret = LNN_NG; goto done;
                }
            }
            sl = gp->slot[produced % ndev][(produced / ndev) % nslots];
            uj.g = g; uj.sdata = sl ? LINNEAmd_SlotData(sl) : NULL; uj.sprm = sl ? LINNEAmd_SlotParams(sl) : NULL;
            uj.sstream = NULL; uj.sbitpos = NULL; uj.sbitend = NULL; uj.s16 = NULL;
            g->seg_first = g->nblk ? g->offs[0] : off; g->seg_bytes = 0;
            t0 = now_s();
            if (stream_mode && sl && ncomp) {
                g->seg_bytes = off - g->seg_first;                /* the scan above stopped at `off`: the group's bytes are [seg_first, off) */
                if (g->seg_bytes > LINNEAmd_SlotStreamCapacity(sl)) { stream_mode = 0; g_last_decode_mode |= 2u; for (i = 0; i < LNN_MAX_DEVICES; i++) for (f = 0; f < LNN_SLOTS; f++) if (gp->slot[i][f]) (void)LINNEAmd_SlotWait(gp->slot[i][f]); goto setup; }
                uj.sstream = LINNEAmd_SlotStream(sl); uj.sbitpos = LINNEAmd_SlotBitPos(sl); uj.sbitend = LINNEAmd_SlotBitEnd(sl); uj.seg_first = g->seg_first; uj.seg_bytes = g->seg_bytes;
            }
            lnn_parallel_for(g->nblk, threads, unpack_blocks, &uj);
            t_parse += now_s() - t0;
            for (f = 0; f < g->nblk; f++) {
                if (g->rets[f] != LNN_OK) { ret = g->rets[f]; g->nblk = f; scanning = 0; break; }    /* first failing block wins */
                /* The scan above trusted the blocks' size fields (that is what lets the host threads parse a group in parallel);
                 * the reference steps by the bytes its bit reader actually consumed (linne_decoder.c:725-733).  They differ only
                 * in a damaged stream (CRC check off): the blocks behind such a block were located wrongly -- drop them and scan
                 * on from where the reference would. */
                if (f + 1 < g->nblk ? (g->offs[f] + g->cons[f] != g->offs[f + 1]) : (g->offs[f] + g->cons[f] != off)) {
                    g->nblk = f + 1; off = g->offs[f] + g->cons[f];
                    break;
                }
            }
            g->ncomp = 0;
            for (f = 0; f < g->nblk; f++) {
                if (g->types[f] == LNN_BLOCK_COMPRESS) g->cn[g->ncomp++] = g->ns[f];
                progress = g->prog[f] + g->ns[f];
            }
            if (g->ncomp) {
                int dret;
                t0 = now_s();
                g->streamed = (uj.sstream != NULL);
                dret = g->streamed ? LINNEAmd_SlotDecodeStreamSubmit(sl, g->seg_bytes, g->cn, g->ncomp) : LINNEAmd_SlotDecodeSubmit(sl, g->cn, g->ncomp);
                t_submit += now_s() - t0;
                if (dret != LNN_OK) { report(gp->ctx[produced % ndev], "SlotDecodeSubmit", dret); ret = dret; goto done; }
            }
            produced++;
            continue;
        }
        scanning = 0;                       /* (armed again below once a slot is free, if the stream goes on) */
        if (consumed_groups < produced) {
            struct dgroup *g = &grp[consumed_groups % window];
            struct LINNEAmdSlot *sl = gp->slot[consumed_groups % ndev][(consumed_groups / ndev) % nslots];
            if (g->ncomp) {
                int dret;
                t0 = now_s();
                dret = LINNEAmd_SlotWait(sl);
                t_wait += now_s() - t0;
                if (dret != LNN_OK) { report(gp->ctx[consumed_groups % ndev], "SlotWait", dret); ret = dret; goto done; }
                uj.g = g; uj.sdata = LINNEAmd_SlotData(sl); uj.sprm = NULL; uj.s16 = NULL;
                if (g->streamed) {      /* did the device's decoder walk every block to the end its size field names? */
                    const uint64_t *eb = LINNEAmd_SlotEndBits(sl);
                    int anomaly = 0;
                    for (f = 0; f < g->nblk && !anomaly; f++) {
                        uint64_t pay;
                        if (g->types[f] != LNN_BLOCK_COMPRESS) continue;
                        pay = (g->offs[f] - g->seg_first + 11u) * 8u;
                        if (eb[g->cidx[f]] == ~(uint64_t)0 || eb[g->cidx[f]] < pay || 11u + ((eb[g->cidx[f]] - pay + 7u) >> 3) != g->cons[f]) anomaly = 1;
                    }
                    if (!anomaly && LINNEAmd_SlotPcm16(sl)) {
                        if (LINNEAmd_SlotPcm16Valid(sl) && !getenv("LINNE_AMD_DEBUG_NO_PCM16")) { uj.s16 = LINNEAmd_SlotPcm16(sl); uj.swidth = LINNEAmd_SlotPcmWidth(sl); }      /* (the knob: tests take the int32 way) */
                        else if (LINNEAmd_SlotFetchPcm32(sl, g->ncomp) != LNN_OK) anomaly = 1;
                        else uj.sdata = LINNEAmd_SlotData(sl);             /* (allocated by the fetch) */
                    }
                    if (anomaly) {      /* not what an encoder writes: the host's decoder defines the result */
                        stream_mode = 0; g_last_decode_mode |= 2u;
                        for (i = 0; i < LNN_MAX_DEVICES; i++) for (f = 0; f < LNN_SLOTS; f++) if (gp->slot[i][f]) (void)LINNEAmd_SlotWait(gp->slot[i][f]);
                        goto setup;
                    }
                }
                t0 = now_s();
                lnn_parallel_for(g->nblk, threads, scatter_blocks, &uj);
                t_scatter += now_s() - t0;
            }
            consumed_groups++;
            if (ret == LNN_OK && progress < hd->num_samples && off < data_size) scanning = 1;
        }
    }
    if (stream_mode && produced) g_last_decode_mode |= 1u;
    if (trace_on()) fprintf(stderr, "liblinne_amd: DecodeWhole %u threads%s: parse %.1f ms, submit %.1f, wait %.1f, scatter %.1f, total %.1f ms\n",
            threads, stream_mode ? ", Rice codes decoded on the device" : "", t_parse * 1e3, t_submit * 1e3, t_wait * 1e3, t_scatter * 1e3, (now_s() - t_begin) * 1e3);
done:
    for (f = 0; f < LNN_MAX_DEVICES; f++) for (i = 0; i < LNN_SLOTS; i++) if (gp->slot[f][i]) (void)LINNEAmd_SlotWait(gp->slot[f][i]);
    for (i = 0; i < ngalloc; i++) dgroup_free(&grp[i]);
    return (LINNEApiResult)ret;
}

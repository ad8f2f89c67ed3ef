/*
 * lnn_api.c -- the LINNE public C API (include/linne_encoder.h, include/linne_decoder.h) on top of the
 * gfx950 hot path (lnn_device.hip) and the host entropy stage (lnn_entropy.c).
 *
 * Same signatures, ownership and error conventions as the reference; citations are
 * file:line under /root/reference.  There is no CPU fallback for the prediction path: without a HIP device
 * EncodeBlock / EncodeWhole / DecodeBlock / DecodeWhole of COMPRESS data return LINNE_APIRESULT_NG and say
 * so on stderr.
 */
#include "linne_encoder.h"
#include "linne_decoder.h"
#include "lnn_host.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define LNN_ALIGN 16u
#define ALIGN_UP(v) (((v) + (LNN_ALIGN - 1u)) & ~(uintptr_t)(LNN_ALIGN - 1u))

extern int LINNEAmd_ReserveScratch(struct LINNEAmdContext *ctx, uint64_t bytes);

static uint32_t default_threads(void)
{
    const char *e = getenv("LINNE_AMD_THREADS");
    long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return (uint32_t)n;
}
static int default_device(void)
{
    const char *e = getenv("LINNE_AMD_DEVICE");
    return e ? atoi(e) : 0;
}
static struct LINNEAmdContext *open_context(const char *who)
{
    struct LINNEAmdContext *ctx = LINNEAmd_ContextCreate(default_device(), 256ull << 20);
    if (!ctx) fprintf(stderr, "liblinne_amd: %s: no usable HIP device (LINNE_AMD_DEVICE=%d); the prediction path has no CPU fallback\n", who, default_device());
    return ctx;
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
    struct LINNEAmdContext *ctx;        /* GPU resources: outside the work area, released by Destroy */
    double parcor_state;                /* oracle quirk Q2, carried from block to block */
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
    if (encoder->ctx) { LINNEAmd_ContextDestroy(encoder->ctx); encoder->ctx = NULL; }
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
    /* refinements outside the hot path this build provides */
    if (parameter->enable_learning != 0 || parameter->num_afmethod_iterations != 0) {
        fprintf(stderr, "liblinne_amd: -l (learning) and -a N (auxiliary-function iterations) are not offered by the MI355X path\n");
        return LINNE_APIRESULT_INVALID_FORMAT;
    }
    if (parameter->num_channels > LINNE_MAX_NUM_CHANNELS) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    memset(&h, 0, sizeof(h));
    h.num_channels = parameter->num_channels; h.sampling_rate = parameter->sampling_rate; h.bits_per_sample = parameter->bits_per_sample;
    h.num_samples_per_block = parameter->num_samples_per_block; h.preset = parameter->preset; h.ch_process_method = parameter->ch_process_method;
    encoder->header = h; encoder->shape = shape; encoder->layers = ly;
    encoder->set_parameter = 1;
    return LINNE_APIRESULT_OK;
}

static LINNEApiResult encoder_device(struct LINNEEncoder *enc)
{
    if (enc->ctx == NULL) enc->ctx = open_context("LINNEEncoder");
    return enc->ctx ? LINNE_APIRESULT_OK : LINNE_APIRESULT_NG;
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
    if (encoder_device(encoder) != LINNE_APIRESULT_OK) return LINNE_APIRESULT_NG;
    S = encoder->shape.num_samples_per_block; C = encoder->shape.num_channels;
    for (ch = 0; ch < C; ch++) {
        if (input[ch] == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
        memcpy(encoder->pcm + (size_t)ch * S, input[ch], sizeof(int32_t) * num_samples);
        if (num_samples < S) memset(encoder->pcm + (size_t)ch * S + num_samples, 0, sizeof(int32_t) * (S - num_samples));
    }
    ret = LINNEAmd_EncodeFramesHost(encoder->ctx, &encoder->shape, encoder->pcm, &num_samples, 1, encoder->residual, encoder->params, encoder->stats);
    if (ret != LNN_OK) { report(encoder->ctx, "EncodeFramesHost", ret); return (LINNEApiResult)ret; }
    ret = LINNEAmd_PackFrames(&encoder->shape, encoder->pcm, &num_samples, 1, encoder->residual, encoder->params, encoder->stats,
            data, data_size, output_size, &encoder->parcor_state, 1);
    return (LINNEApiResult)ret;
}

LINNEApiResult LINNEEncoder_EncodeWhole(struct LINNEEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
        uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    LINNEApiResult r;
    uint32_t S, C, F, f, ch, base;
    uint64_t off = LINNE_HEADER_SIZE, CS;
    int32_t *pcm = NULL, *res = NULL, *prm = NULL; double *st = NULL; uint32_t *nsm = NULL, *sizes = NULL;
    const uint32_t group = 2048;                    /* frames analysed per device batch */
    int ret = LNN_OK;
    if (encoder == NULL || input == NULL || data == NULL || output_size == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if (encoder->set_parameter != 1) return LINNE_APIRESULT_PARAMETER_NOT_SET;
    encoder->header.num_samples = num_samples;
    if ((r = LINNEEncoder_EncodeHeader(&encoder->header, data, data_size)) != LINNE_APIRESULT_OK) return r;
    if (encoder_device(encoder) != LINNE_APIRESULT_OK) return LINNE_APIRESULT_NG;
    S = encoder->shape.num_samples_per_block; C = encoder->shape.num_channels; CS = (uint64_t)C * S;
    F = (uint32_t)(((uint64_t)num_samples + S - 1) / S);
    {
        const uint32_t g = (F < group) ? F : group;
        pcm = malloc(sizeof(int32_t) * CS * g); res = malloc(sizeof(int32_t) * CS * g);
        prm = malloc(sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (size_t)C * g); st = malloc(sizeof(double) * LINNE_AMD_STAT_WORDS * (size_t)C * g);
        nsm = malloc(sizeof(uint32_t) * g); sizes = malloc(sizeof(uint32_t) * g);
        if (!pcm || !res || !prm || !st || !nsm || !sizes) { ret = LNN_NG; goto done; }
        if (g > 32) (void)LINNEAmd_ReserveScratch(encoder->ctx, 4ull << 30);
    }
    for (base = 0; base < F; base += group) {
        const uint32_t cnt = (F - base < group) ? (F - base) : group;
        for (f = 0; f < cnt; f++) {
            const uint64_t start = (uint64_t)(base + f) * S;
            const uint32_t n = (num_samples - start < S) ? (uint32_t)(num_samples - start) : S;
            nsm[f] = n;
            for (ch = 0; ch < C; ch++) {
                int32_t *dst = pcm + f * CS + (size_t)ch * S;
                memcpy(dst, input[ch] + start, sizeof(int32_t) * n);
                if (n < S) memset(dst + n, 0, sizeof(int32_t) * (S - n));
            }
        }
        ret = LINNEAmd_EncodeFramesHost(encoder->ctx, &encoder->shape, pcm, nsm, cnt, res, prm, st);
        if (ret != LNN_OK) { report(encoder->ctx, "EncodeFramesHost", ret); goto done; }
        ret = LINNEAmd_PackFrames(&encoder->shape, pcm, nsm, cnt, res, prm, st, data + off, data_size - off, sizes, &encoder->parcor_state, default_threads());
        if (ret != LNN_OK) goto done;
        for (f = 0; f < cnt; f++) off += sizes[f];
    }
    *output_size = (uint32_t)off;
done:
    free(pcm); free(res); free(prm); free(st); free(nsm); free(sizes);
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
    struct LINNEAmdContext *ctx;
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
    if (decoder->ctx) { LINNEAmd_ContextDestroy(decoder->ctx); decoder->ctx = NULL; }
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

static LINNEApiResult decoder_device(struct LINNEDecoder *dec)
{
    if (dec->ctx == NULL) dec->ctx = open_context("LINNEDecoder");
    return dec->ctx ? LINNE_APIRESULT_OK : LINNE_APIRESULT_NG;
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
        if (decoder_device(decoder) != LINNE_APIRESULT_OK) return LINNE_APIRESULT_NG;
        ret = LINNEAmd_DecodeFramesHost(decoder->ctx, &decoder->shape, decoder->samples, &n, 1, decoder->params);
        if (ret != LNN_OK) { report(decoder->ctx, "DecodeFramesHost", ret); return (LINNEApiResult)ret; }
    }
    for (ch = 0; ch < C; ch++) memcpy(buffer[ch], decoder->samples + (size_t)ch * S, sizeof(int32_t) * n);
    *decode_size = consumed;
    *num_decode_samples = n;
    return LINNE_APIRESULT_OK;
}

/* ---- whole stream: scan block boundaries, entropy-decode blocks on a thread pool, synthesise on the GPU ---- */
struct unpack_job {
    const struct LINNEDecoder *dec; const uint8_t *data; const uint64_t *offs; const uint64_t *avail; const uint32_t *room;
    int32_t *samples, *params; uint32_t *types, *ns; int *rets; uint32_t first, count;
};
static void *unpack_worker(void *arg)
{
    struct unpack_job *j = arg;
    const struct LINNEAmdShape *sh = &j->dec->shape;
    const uint64_t CS = (uint64_t)sh->num_channels * sh->num_samples_per_block;
    uint32_t f, consumed;
    for (f = j->first; f < j->first + j->count; f++)
        j->rets[f] = lnn_parse_block(sh, &j->dec->layers, j->data + j->offs[f], j->avail[f], j->dec->check_crc, j->room[f],
                &j->types[f], &j->ns[f], &consumed, j->samples + f * CS, j->params + (size_t)f * sh->num_channels * LINNE_AMD_PARAM_WORDS);
    return NULL;
}

LINNEApiResult LINNEDecoder_DecodeWhole(struct LINNEDecoder *decoder, const uint8_t *data, uint32_t data_size,
        int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples)
{
    LINNEApiResult r;
    struct LINNEHeader h;
    const struct LINNEHeader *hd;
    uint32_t C, S, group = 2048, nblk = 0, f, ch, progress = 0, t;
    uint64_t off, CS;
    uint64_t *offs = NULL, *avail = NULL; uint32_t *room = NULL, *types = NULL, *ns = NULL, *cn = NULL; int *rets = NULL;
    int32_t *samples = NULL, *params = NULL, *cbuf = NULL, *cprm = NULL;
    int ret = LNN_OK;
    if (decoder == NULL || data == NULL || buffer == NULL) return LINNE_APIRESULT_INVALID_ARGUMENT;
    if ((r = LINNEDecoder_DecodeHeader(data, data_size, &h)) != LINNE_APIRESULT_OK) return r;
    if ((r = LINNEDecoder_SetHeader(decoder, &h)) != LINNE_APIRESULT_OK) return r;
    hd = &decoder->header;
    if (buffer_num_channels < hd->num_channels || buffer_num_samples < hd->num_samples) return LINNE_APIRESULT_INSUFFICIENT_BUFFER;
    C = decoder->shape.num_channels; S = decoder->shape.num_samples_per_block; CS = (uint64_t)C * S;
    offs = malloc(sizeof(*offs) * group); avail = malloc(sizeof(*avail) * group); room = malloc(sizeof(*room) * group);
    types = malloc(sizeof(*types) * group); ns = malloc(sizeof(*ns) * group); cn = malloc(sizeof(*cn) * group); rets = malloc(sizeof(*rets) * group);
    samples = malloc(sizeof(int32_t) * CS * group); params = malloc(sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (size_t)C * group);
    cbuf = malloc(sizeof(int32_t) * CS * group); cprm = malloc(sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * (size_t)C * group);
    if (!offs || !avail || !room || !types || !ns || !cn || !rets || !samples || !params || !cbuf || !cprm) { ret = LNN_NG; goto done; }
    off = LINNE_HEADER_SIZE;
    while (progress < hd->num_samples && off < data_size && ret == LNN_OK) {
        /* scan up to `group` block boundaries from the size fields (linne_decoder.c:603-615) */
        uint32_t scan_progress = progress, nthreads = default_threads(), first = 0, ncomp = 0;
        pthread_t th[64];
        struct unpack_job jobs[64];
        nblk = 0;
        while (nblk < group && scan_progress < hd->num_samples && off < data_size) {
            const uint64_t rem = data_size - off;
            uint32_t bsize;
            offs[nblk] = off; avail[nblk] = rem; room[nblk] = buffer_num_samples - scan_progress;
            if (rem < 11 || get_be16(data + off) != 0xFFFF) { nblk++; break; }      /* the parser reports the error */
            bsize = get_be32(data + off + 2);
            if ((uint64_t)bsize + 6 > rem) { nblk++; break; }
            scan_progress += get_be16(data + off + 9);
            off += (uint64_t)bsize + 6;
            nblk++;
        }
        if (nthreads > nblk) nthreads = nblk ? nblk : 1;
        for (t = 0; t < nthreads; t++) {
            const uint32_t c = nblk / nthreads + ((t < nblk % nthreads) ? 1u : 0u);
            struct unpack_job *j = &jobs[t];
            j->dec = decoder; j->data = data; j->offs = offs; j->avail = avail; j->room = room; j->samples = samples; j->params = params;
            j->types = types; j->ns = ns; j->rets = rets; j->first = first; j->count = c; first += c;
            if (nthreads == 1) unpack_worker(j); else pthread_create(&th[t], NULL, unpack_worker, j);
        }
        if (nthreads > 1) for (t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        for (f = 0; f < nblk; f++) if (rets[f] != LNN_OK) { ret = rets[f]; nblk = f; break; }    /* first failing block wins */
        /* gather COMPRESS blocks, synthesise them as one batch */
        for (f = 0; f < nblk; f++) if (types[f] == LNN_BLOCK_COMPRESS) {
            memcpy(cbuf + ncomp * CS, samples + f * CS, sizeof(int32_t) * CS);
            memcpy(cprm + (size_t)ncomp * C * LINNE_AMD_PARAM_WORDS, params + (size_t)f * C * LINNE_AMD_PARAM_WORDS, sizeof(int32_t) * LINNE_AMD_PARAM_WORDS * C);
            cn[ncomp++] = ns[f];
        }
        if (ncomp) {
            int dret;
            if (decoder_device(decoder) != LINNE_APIRESULT_OK) { ret = LNN_NG; goto done; }
            dret = LINNEAmd_DecodeFramesHost(decoder->ctx, &decoder->shape, cbuf, cn, ncomp, cprm);
            if (dret != LNN_OK) { report(decoder->ctx, "DecodeFramesHost", dret); ret = dret; goto done; }
        }
        ncomp = 0;
        for (f = 0; f < nblk; f++) {
            const int32_t *src = (types[f] == LNN_BLOCK_COMPRESS) ? (cbuf + (ncomp++) * CS) : (samples + f * CS);
            for (ch = 0; ch < C; ch++) memcpy(buffer[ch] + progress, src + (size_t)ch * S, sizeof(int32_t) * ns[f]);
            progress += ns[f];
        }
    }
done:
    free(offs); free(avail); free(room); free(types); free(ns); free(cn); free(rets); free(samples); free(params); free(cbuf); free(cprm);
    return (LINNEApiResult)ret;
}

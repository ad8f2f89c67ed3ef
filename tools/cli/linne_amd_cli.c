/*
 * linne_amd_cli -- command line encoder/decoder on liblinne_amd.so.
 *
 * Same options and the same files as the reference's tools/linne_codec (linne_codec.c:15-33: -e -d -m -l -a -c -h -v,
 * INPUT OUTPUT), written from scratch with its own WAV reader/writer (wavio.c).  Differences, all additive:
 *   - whole streams go through LINNEEncoder_EncodeWhole / LINNEDecoder_DecodeWhole (pipelined over the GPU and the host
 *     threads; identical bytes); -B / --block-at-a-time walks LINNEEncoder_EncodeBlock like the reference tool does;
 *   - --batch DIR encodes (or decodes) every remaining argument into DIR with ONE handle, so the GPU context and the
 *     pinned staging slots are set up once (BASELINE config 4: many independent tracks);
 *   - -l and -a N are parsed and refused: the MI355X path does not offer them.
 */
#include "linne_decoder.h"
#include "linne_encoder.h"
#include "wavio.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct options {
    int encode, decode, no_crc, learning, block_at_a_time, help, version, quiet;
    long mode, af_iterations;
    const char *batch_dir;
    const char *files[4096];
    int num_files;
};

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void usage(const char *argv0)
{
    printf("Usage: %s [options] INPUT_FILE_NAME OUTPUT_FILE_NAME \n", argv0);
    printf("       %s [options] --batch OUTPUT_DIRECTORY INPUT_FILE_NAME... \n", argv0);
    printf("options: \n"
           "  -e, --encode                           Encode mode \n"
           "  -d, --decode                           Decode mode \n"
           "  -m, --mode N                           Specify compress mode: 0(fast), ..., 7(high compression) (default:0) \n"
           "  -l, --enable-learning                  (not offered by the MI355X path) \n"
           "  -a, --auxiliary-function-iteration N   (not offered by the MI355X path unless N is 0) \n"
           "  -c, --no-crc-check                     Whether to NOT check CRC16 at decoding (default:no) \n"
           "  -B, --block-at-a-time                  Encode block by block (LINNEEncoder_EncodeBlock), as the reference tool does \n"
           "      --batch DIR                        Process every input into DIR with one handle (names keep their stem) \n"
           "  -q, --quiet                            No progress or summary lines \n"
           "  -h, --help                             Show command help message \n"
           "  -v, --version                          Show version information \n");
}

/* option with a value: "-m 7", "-m7", "--mode 7", "--mode=7" */
static int take_value(int argc, char **argv, int *i, const char *shortopt, const char *longopt, const char **value)
{
    const char *a = argv[*i];
    const size_t ll = strlen(longopt);
    if (strcmp(a, shortopt) == 0 || strcmp(a, longopt) == 0) {
        if (*i + 1 >= argc) { fprintf(stderr, "%s needs a value. \n", a); exit(1); }
        *value = argv[++*i];
        return 1;
    }
    if (strncmp(a, longopt, ll) == 0 && a[ll] == '=') { *value = a + ll + 1; return 1; }
    if (shortopt[0] && strncmp(a, shortopt, 2) == 0 && a[1] != '-' && a[2] != '\0') { *value = a + 2; return 1; }
    return 0;
}

static void parse(int argc, char **argv, struct options *o)
{
    int i;
    memset(o, 0, sizeof(*o));
    for (i = 1; i < argc; i++) {
        const char *a = argv[i], *v = NULL;
        if (strcmp(a, "-e") == 0 || strcmp(a, "--encode") == 0) o->encode = 1;
        else if (strcmp(a, "-d") == 0 || strcmp(a, "--decode") == 0) o->decode = 1;
        else if (strcmp(a, "-c") == 0 || strcmp(a, "--no-crc-check") == 0) o->no_crc = 1;
        else if (strcmp(a, "-l") == 0 || strcmp(a, "--enable-learning") == 0) o->learning = 1;
        else if (strcmp(a, "-B") == 0 || strcmp(a, "--block-at-a-time") == 0) o->block_at_a_time = 1;
        else if (strcmp(a, "-q") == 0 || strcmp(a, "--quiet") == 0) o->quiet = 1;
        else if (strcmp(a, "-h") == 0 || strcmp(a, "--help") == 0) o->help = 1;
        else if (strcmp(a, "-v") == 0 || strcmp(a, "--version") == 0) o->version = 1;
        else if (take_value(argc, argv, &i, "-m", "--mode", &v)) {
            char *end; o->mode = strtol(v, &end, 10);
            if (*end != '\0' || o->mode < 0 || o->mode >= LINNE_NUM_PARAMETER_PRESETS) { fprintf(stderr, "%s: Encode preset number is out of range. \n", argv[0]); exit(1); }
        } else if (take_value(argc, argv, &i, "-a", "--auxiliary-function-iteration", &v)) {
            char *end; o->af_iterations = strtol(v, &end, 10);
            if (*end != '\0' || o->af_iterations < 0 || o->af_iterations > 255) { fprintf(stderr, "%s: auxiliary function iteration count is out of range. \n", argv[0]); exit(1); }
        } else if (take_value(argc, argv, &i, "", "--batch", &v)) o->batch_dir = v;
        else if (a[0] == '-' && a[1] != '\0') { fprintf(stderr, "%s: unknown option %s \n", argv[0], a); exit(1); }
        else if (o->num_files < (int)(sizeof(o->files) / sizeof(o->files[0]))) o->files[o->num_files++] = a;
        else { fprintf(stderr, "%s: too many files. \n", argv[0]); exit(1); }
    }
}

/* DIR/stem.ext for a batch item */
static void batch_name(char *dst, size_t cap, const char *dir, const char *input, const char *ext)
{
    const char *base = strrchr(input, '/');
    const char *dot;
    size_t stem;
    base = base ? base + 1 : input;
    dot = strrchr(base, '.');
    stem = dot ? (size_t)(dot - base) : strlen(base);
    snprintf(dst, cap, "%s/%.*s%s", dir, (int)stem, base, ext);
}

static int read_file(const char *path, uint8_t **data, uint32_t *size)
{
    FILE *fp = fopen(path, "rb");
    long n;
    if (!fp) return -1;
    if (fseek(fp, 0, SEEK_END) != 0 || (n = ftell(fp)) < 0 || fseek(fp, 0, SEEK_SET) != 0 || (unsigned long)n > 0xFFFFFFFFul) { fclose(fp); return -1; }
    if (!(*data = malloc(n ? (size_t)n : 1))) { fclose(fp); return -1; }
    if (fread(*data, 1, (size_t)n, fp) != (size_t)n) { fclose(fp); free(*data); return -1; }
    fclose(fp);
    *size = (uint32_t)n;
    return 0;
}

static int encode_one(struct LINNEEncoder *enc, const struct options *o, const char *in_path, const char *out_path)
{
    struct wav_pcm wav;
    struct LINNEEncodeParameter par;
    char err[256];
    uint8_t *buf = NULL;
    uint64_t cap;
    uint32_t out_size = 0;
    LINNEApiResult ret;
    FILE *fp;
    const double t0 = now_s();
    if (wav_read(in_path, &wav, err, sizeof(err)) != 0) { fprintf(stderr, "Failed to open %s. (%s) \n", in_path, err); return 1; }
    if (wav.num_channels > LINNE_MAX_NUM_CHANNELS) { fprintf(stderr, "%s: %u channels, at most %d are supported. \n", in_path, wav.num_channels, LINNE_MAX_NUM_CHANNELS); wav_free(&wav); return 1; }
    par.num_channels = (uint16_t)wav.num_channels; par.bits_per_sample = (uint16_t)wav.bits_per_sample; par.sampling_rate = wav.sampling_rate;
    par.num_samples_per_block = 5 * 2048;                          /* linne_codec.c:75-76 */
    par.ch_process_method = (wav.num_channels >= 2) ? LINNE_CH_PROCESS_METHOD_MS : LINNE_CH_PROCESS_METHOD_NONE;
    par.preset = (uint8_t)o->mode; par.enable_learning = (uint8_t)o->learning; par.num_afmethod_iterations = (uint8_t)o->af_iterations;
    if ((ret = LINNEEncoder_SetEncodeParameter(enc, &par)) != LINNE_APIRESULT_OK) { fprintf(stderr, "Failed to set encode parameter: %d \n", ret); wav_free(&wav); return 1; }
    cap = 2ull * ((uint64_t)wav.num_samples * wav.num_channels * (wav.bits_per_sample / 8) + 44) + 65536;     /* "twice the input", linne_codec.c:93-95 */
    if (cap > 0xFFFFFFFFull) cap = 0xFFFFFFFFull;
    if (!(buf = malloc((size_t)cap))) { fprintf(stderr, "out of memory \n"); wav_free(&wav); return 1; }
    if (!o->block_at_a_time) {
        ret = LINNEEncoder_EncodeWhole(enc, (const int32_t *const *)wav.plane, wav.num_samples, buf, (uint32_t)cap, &out_size);
        if (ret != LINNE_APIRESULT_OK) { fprintf(stderr, "Failed to encode! ret:%d \n", ret); free(buf); wav_free(&wav); return 1; }
    } else {
        struct LINNEHeader h;
        uint32_t progress = 0, ch;
        memset(&h, 0, sizeof(h));
        h.num_channels = par.num_channels; h.num_samples = wav.num_samples; h.sampling_rate = par.sampling_rate; h.bits_per_sample = par.bits_per_sample;
        h.num_samples_per_block = par.num_samples_per_block; h.preset = par.preset; h.ch_process_method = par.ch_process_method;
        if ((ret = LINNEEncoder_EncodeHeader(&h, buf, (uint32_t)cap)) != LINNE_APIRESULT_OK) { fprintf(stderr, "Failed to encode header! ret:%d \n", ret); free(buf); wav_free(&wav); return 1; }
        out_size = LINNE_HEADER_SIZE;
        while (progress < wav.num_samples) {
            const int32_t *ptr[LINNE_MAX_NUM_CHANNELS];
            const uint32_t n = (wav.num_samples - progress < par.num_samples_per_block) ? (wav.num_samples - progress) : par.num_samples_per_block;
            uint32_t wrote = 0;
            for (ch = 0; ch < wav.num_channels; ch++) ptr[ch] = wav.plane[ch] + progress;
            if ((ret = LINNEEncoder_EncodeBlock(enc, ptr, n, buf + out_size, (uint32_t)cap - out_size, &wrote)) != LINNE_APIRESULT_OK) {
                fprintf(stderr, "Failed to encode! ret:%d \n", ret); free(buf); wav_free(&wav); return 1;
            }
            out_size += wrote; progress += n;
            if (!o->quiet) { printf("progress... %5.2f%% \r", (progress * 100.0f) / wav.num_samples); fflush(stdout); }
        }
    }
    if (!(fp = fopen(out_path, "wb")) || fwrite(buf, 1, out_size, fp) != out_size) { fprintf(stderr, "File output error! %s \n", out_path); if (fp) fclose(fp); free(buf); wav_free(&wav); return 1; }
    fclose(fp);
    if (!o->quiet) {
        const uint64_t in_bytes = (uint64_t)wav.num_samples * wav.num_channels * (wav.bits_per_sample / 8) + 44;
        printf("finished: %llu -> %u (%6.2f %%) in %.3f s \n", (unsigned long long)in_bytes, out_size, 100.0 * (double)out_size / (double)in_bytes, now_s() - t0);
    }
    free(buf); wav_free(&wav);
    return 0;
}

static int decode_one(struct LINNEDecoder *dec, const struct options *o, const char *in_path, const char *out_path)
{
    uint8_t *buf = NULL;
    uint32_t size = 0;
    struct LINNEHeader h;
    struct wav_pcm wav;
    char err[256];
    LINNEApiResult ret;
    const double t0 = now_s();
    if (read_file(in_path, &buf, &size) != 0) { fprintf(stderr, "Failed to open %s. \n", in_path); return 1; }
    if ((ret = LINNEDecoder_DecodeHeader(buf, size, &h)) != LINNE_APIRESULT_OK) { fprintf(stderr, "Failed to get header information: %d \n", ret); free(buf); return 1; }
    memset(&wav, 0, sizeof(wav));
    wav.num_channels = h.num_channels; wav.sampling_rate = h.sampling_rate; wav.bits_per_sample = h.bits_per_sample; wav.num_samples = h.num_samples;
    if (wav_alloc(&wav) != 0) { fprintf(stderr, "Failed to create wav handle. \n"); free(buf); wav_free(&wav); return 1; }
    if ((ret = LINNEDecoder_DecodeWhole(dec, buf, size, wav.plane, wav.num_channels, wav.num_samples)) != LINNE_APIRESULT_OK) {
        fprintf(stderr, "Decoding error! %d \n", ret); free(buf); wav_free(&wav); return 1;
    }
    if (wav_write(out_path, &wav, err, sizeof(err)) != 0) { fprintf(stderr, "Failed to write wav file. (%s) \n", err); free(buf); wav_free(&wav); return 1; }
    if (!o->quiet) printf("finished: %u -> %llu in %.3f s \n", size, (unsigned long long)wav.num_samples * wav.num_channels * (wav.bits_per_sample / 8) + 44, now_s() - t0);
    free(buf); wav_free(&wav);
    return 0;
}

int main(int argc, char **argv)
{
    struct options o;
    int i, rc = 0;
    if (argc == 1) { usage(argv[0]); printf("Type `%s -h` to display command helps. \n", argv[0]); return 1; }
    parse(argc, argv, &o);
    if (o.help) { usage(argv[0]); return 0; }
    if (o.version) { printf("LINNE -- LInear-predictive Neural Net Encoder Version.%d (liblinne_amd, MI355X) \n", LINNE_CODEC_VERSION); return 0; }
    if (o.encode && o.decode) { fprintf(stderr, "%s: encode and decode mode cannot specify simultaneously. \n", argv[0]); return 1; }
    if (!o.encode && !o.decode) { fprintf(stderr, "%s: decode(-d) or encode(-e) option must be specified. \n", argv[0]); return 1; }
    if (o.batch_dir ? (o.num_files < 1) : (o.num_files != 2)) { fprintf(stderr, "%s: input and output file name must be specified. \n", argv[0]); return 1; }
    if (o.encode) {
        struct LINNEEncoderConfig cfg;
        struct LINNEEncoder *enc;
        cfg.max_num_channels = LINNE_MAX_NUM_CHANNELS; cfg.max_num_samples_per_block = 16 * 1024; cfg.max_num_layers = 5; cfg.max_num_parameters_per_layer = 128;
        if (!(enc = LINNEEncoder_Create(&cfg, NULL, 0))) { fprintf(stderr, "Failed to create encoder handle. \n"); return 1; }
        if (!o.batch_dir) rc = encode_one(enc, &o, o.files[0], o.files[1]);
        else for (i = 0; i < o.num_files && rc == 0; i++) { char out[4096]; batch_name(out, sizeof(out), o.batch_dir, o.files[i], ".lnn"); rc = encode_one(enc, &o, o.files[i], out); }
        LINNEEncoder_Destroy(enc);
    } else {
        struct LINNEDecoderConfig cfg;
        struct LINNEDecoder *dec;
        cfg.max_num_channels = LINNE_MAX_NUM_CHANNELS; cfg.max_num_layers = 5; cfg.max_num_parameters_per_layer = 128; cfg.check_crc = o.no_crc ? 0 : 1;
        if (!(dec = LINNEDecoder_Create(&cfg, NULL, 0))) { fprintf(stderr, "Failed to create decoder handle. \n"); return 1; }
        if (!o.batch_dir) rc = decode_one(dec, &o, o.files[0], o.files[1]);
        else for (i = 0; i < o.num_files && rc == 0; i++) { char out[4096]; batch_name(out, sizeof(out), o.batch_dir, o.files[i], ".wav"); rc = decode_one(dec, &o, o.files[i], out); }
        LINNEDecoder_Destroy(dec);
    }
    return rc;
}

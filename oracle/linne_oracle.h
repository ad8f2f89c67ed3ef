/*
 * linne_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the LINNE per-frame prediction path and of the host stages around it,
 * written from the reference's behaviour (citations are file:line under /root/reference).  It keeps the
 * reference's exact operation order -- including the redundant recomputations and the two stale-buffer
 * quirks -- so it doubles as the "port" CPU baseline.  It is pinned against the real reference
 * (oracle/_ref/liblinne_ref.so, built by oracle/Makefile from /root/reference) and against the golden
 * vectors under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (linne_amd/) never links, imports or executes it.
 */
#ifndef LINNE_ORACLE_H_INCLUDED
#define LINNE_ORACLE_H_INCLUDED

#include <stdint.h>

#define ORACLE_MAX_CHANNELS   8
#define ORACLE_MAX_LAYERS     3
#define ORACLE_MAX_PARAMS     128
#define ORACLE_NUM_PREEM      2
#define ORACLE_MAX_REGULARS   4

/* same numeric values as LINNEApiResult (include/linne.h:17-26) */
enum {
    ORACLE_OK = 0, ORACLE_INVALID_ARGUMENT, ORACLE_INVALID_FORMAT, ORACLE_INSUFFICIENT_BUFFER,
    ORACLE_INSUFFICIENT_DATA, ORACLE_PARAMETER_NOT_SET, ORACLE_DETECT_DATA_CORRUPTION, ORACLE_NG
};

/* block types (libs/linne_internal/include/linne_internal.h:47-52) */
enum { ORACLE_BLOCK_COMPRESS = 0, ORACLE_BLOCK_SILENT = 1, ORACLE_BLOCK_RAW = 2 };

struct OracleEncodeParameter {          /* mirrors struct LINNEEncodeParameter (include/linne_encoder.h:8-17) */
    uint32_t num_channels;
    uint32_t bits_per_sample;
    uint32_t sampling_rate;
    uint32_t num_samples_per_block;
    uint32_t preset;
    uint32_t ch_process_method;         /* 0 none, 1 mid/side */
};

/* hot-path result of one channel of one frame ("channel-frame") */
struct OracleChannelTap {
    int32_t  preem_prev[ORACLE_NUM_PREEM];
    int32_t  preem_coef[ORACLE_NUM_PREEM];
    uint32_t num_units[ORACLE_MAX_LAYERS];
    uint32_t rshift[ORACLE_MAX_LAYERS];
    int32_t  coef[ORACLE_MAX_LAYERS][ORACLE_MAX_PARAMS];
    double   coef_double[ORACLE_MAX_LAYERS][ORACLE_MAX_PARAMS];
    double   pass_loss[ORACLE_MAX_REGULARS];
    uint32_t best_pass;
    double   est_r0;                    /* SIN-window r[0] of the block-type decision          */
    double   est_parcor[ORACLE_MAX_PARAMS + 2]; /* parcor buffer as EstimateCodeLength saw it  */
    double   est_length;                /* estimated bits/sample of this channel               */
    double   parcor_tail;               /* parcor[P0] left behind by this channel's analysis   */
};

struct OracleFrameTap {
    uint32_t block_type;
    uint32_t num_samples;
    uint32_t num_analyze_samples;
    struct OracleChannelTap ch[ORACLE_MAX_CHANNELS];
};

struct OracleEncoder;
struct OracleDecoder;

#ifdef __cplusplus
extern "C" {
#endif

struct OracleEncoder *oracle_encoder_create(const struct OracleEncodeParameter *param);
void oracle_encoder_destroy(struct OracleEncoder *enc);
/* -a N: auxiliary-function iterations of the final pass (lpc.c:578-633, linne_network.c:605-630); default 0 */
void oracle_encoder_set_af_iterations(struct OracleEncoder *enc, uint32_t n);
/* -l: the momentum-SGD trainer after the analysis (linne_network.c:805-873, linne_encoder.c:669-675); default off */
void oracle_encoder_set_learning(struct OracleEncoder *enc, uint32_t on);

/* == LINNEEncoder_EncodeBlock (libs/linne_encoder/src/linne_encoder.c:774-862).
 * tap (optional) receives the hot-path intermediates; residual_out (optional) receives
 * residual[ch][num_samples] of a COMPRESS block (planar, row stride = num_samples_per_block). */
int oracle_encode_block(struct OracleEncoder *enc, const int32_t *const *input, uint32_t num_samples,
        uint8_t *data, uint32_t data_size, uint32_t *output_size,
        struct OracleFrameTap *tap, int32_t *residual_out);

/* == LINNEEncoder_EncodeWhole (linne_encoder.c:865-932) */
int oracle_encode_whole(struct OracleEncoder *enc, const int32_t *const *input, uint32_t num_samples,
        uint8_t *data, uint32_t data_size, uint32_t *output_size);

/* Hot path only (no entropy coding, always treated as a COMPRESS block): MS, pre-emphasis, analysis,
 * quantisation, int32 FIR cascade.  input/residual are planar [ch][stride]. */
int oracle_encode_frame_hotpath(struct OracleEncoder *enc, const int32_t *input, uint32_t stride,
        uint32_t num_samples, struct OracleFrameTap *tap, int32_t *residual);

/* == LINNEDecoder_DecodeWhole (libs/linne_decoder/src/linne_decoder.c:671-730); buffer is planar
 * [ch][stride]; header fields are returned in hdr[9] = {format, codec, channels, samples, rate, bits,
 * block, preset, ch_method}. */
int oracle_decode_whole(const uint8_t *data, uint32_t data_size, int32_t *buffer, uint32_t buffer_channels,
        uint32_t stride, uint32_t *hdr, int check_crc);

/* Decode hot path for one frame: int32 IIR synthesis cascade (reverse layer order), de-emphasis, MS->LR.
 * taps supply units/rshift/coef/preem; data is planar [ch][stride], in place. */
int oracle_decode_frame_hotpath(const struct OracleEncodeParameter *param, const struct OracleChannelTap *taps,
        int32_t *data, uint32_t stride, uint32_t num_samples);

/* host-stage helpers exposed for unit tests */
uint16_t oracle_crc16(const uint8_t *data, uint64_t size);
/* partitioned recursive Rice coder: returns bytes written (flushed to a byte boundary) */
uint32_t oracle_rice_encode(const int32_t *data, uint32_t num_samples, uint8_t *out, uint32_t out_size);
uint32_t oracle_rice_decode(const uint8_t *in, uint32_t in_size, int32_t *data, uint32_t num_samples);
/* static Huffman code of symbol sym: returns bit length, *code receives the code word */
uint32_t oracle_huffman_code(uint32_t sym, uint32_t *code);

/* multi-threaded throughput helper for bench.py's cpu_baseline leg ("port"): encodes num_frames frames
 * (planar [frame][ch][block]) with num_threads handles over disjoint frames; returns seconds. */
/* measurement tap (tools/search_margins.py): records of 5 doubles per trial of every unit-count search of THIS thread from now on */
void oracle_set_trial_tap(double *buf, uint32_t cap_records);
uint32_t oracle_trial_tap_count(void);
double oracle_bench_encode(const struct OracleEncodeParameter *param, const int32_t *frames, uint32_t num_frames,
        uint32_t num_threads, uint64_t *total_bytes);

#ifdef __cplusplus
}
#endif

#endif /* LINNE_ORACLE_H_INCLUDED */

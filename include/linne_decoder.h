/*
 * linne_decoder.h -- LINNE decoder API exported by liblinne_amd.so.
 *
 * Drop-in boundary for the reference's include/linne_decoder.h:8-53 (identical layouts and signatures).
 * Parsing, CRC and entropy decoding run on the host; the int32 synthesis cascade, de-emphasis and MS->LR
 * run as HIP kernels on gfx950.
 */
#ifndef LINNE_DECODER_H_INCLUDED
#define LINNE_DECODER_H_INCLUDED

#include "linne.h"
#include "linne_stdint.h"

struct LINNEDecoderConfig {
    uint32_t max_num_channels;
    uint32_t max_num_layers;
    uint32_t max_num_parameters_per_layer;
    uint8_t check_crc;                              /* 1: verify each block's CRC16 */
};

struct LINNEDecoder;

#ifdef __cplusplus
extern "C" {
#endif

/* reference: libs/linne_decoder/src/linne_decoder.c:60-131 */
LINNEApiResult LINNEDecoder_DecodeHeader(
    const uint8_t *data, uint32_t data_size, struct LINNEHeader *header);

/* reference: linne_decoder.c:187-216 */
int32_t LINNEDecoder_CalculateWorkSize(const struct LINNEDecoderConfig *config);

/* reference: linne_decoder.c:219-296 */
struct LINNEDecoder *LINNEDecoder_Create(const struct LINNEDecoderConfig *config, void *work, int32_t work_size);

/* reference: linne_decoder.c:299-306 */
void LINNEDecoder_Destroy(struct LINNEDecoder *decoder);

/* reference: linne_decoder.c:309-354 */
LINNEApiResult LINNEDecoder_SetHeader(
    struct LINNEDecoder *decoder, const struct LINNEHeader *header);

/* reference: linne_decoder.c:564-668 */
LINNEApiResult LINNEDecoder_DecodeBlock(
    struct LINNEDecoder *decoder,
    const uint8_t *data, uint32_t data_size,
    int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples,
    uint32_t *decode_size, uint32_t *num_decode_samples);

/* reference: linne_decoder.c:671-730; every block of the stream is synthesised on the GPU as one batch */
LINNEApiResult LINNEDecoder_DecodeWhole(
    struct LINNEDecoder *decoder,
    const uint8_t *data, uint32_t data_size,
    int32_t **buffer, uint32_t buffer_num_channels, uint32_t buffer_num_samples);

#ifdef __cplusplus
}
#endif

#endif /* LINNE_DECODER_H_INCLUDED */

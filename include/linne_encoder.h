/*
 * linne_encoder.h -- LINNE encoder API exported by liblinne_amd.so.
 *
 * Drop-in boundary for the reference's include/linne_encoder.h:8-64: identical struct layouts and
 * function signatures; same argument meaning, ownership and error conventions.
 * Behind it the per-frame prediction path (MS, pre-emphasis, LPC analysis, layer cascade, quantisation,
 * int32 FIR) runs as HIP kernels on gfx950; the entropy / bit-stream stage stays on the host.
 * A handle also owns GPU resources (device buffers, a stream); they live outside the caller's work area
 * and are released by LINNEEncoder_Destroy.
 */
#ifndef LINNE_ENCODER_H_INCLUDED
#define LINNE_ENCODER_H_INCLUDED

#include "linne.h"
#include "linne_stdint.h"

struct LINNEEncodeParameter {
    uint16_t num_channels;
    uint16_t bits_per_sample;
    uint32_t sampling_rate;
    uint16_t num_samples_per_block;
    uint8_t preset;                                 /* 0 .. LINNE_NUM_PARAMETER_PRESETS-1 */
    LINNEChannelProcessMethod ch_process_method;
    uint8_t enable_learning;                        /* -l : SGD refinement of the analysed parameters (linne_network.c:805-873; on the device: lnn_k_train.h) */
    uint8_t num_afmethod_iterations;                /* -a N: N auxiliary-function iterations in the final pass (lpc.c:578-633; on the device: lnn_k_af.h) */
};

struct LINNEEncoderConfig {
    uint32_t max_num_channels;
    uint32_t max_num_samples_per_block;
    uint32_t max_num_layers;
    uint32_t max_num_parameters_per_layer;
};

struct LINNEEncoder;

#ifdef __cplusplus
extern "C" {
#endif

/* reference: libs/linne_encoder/src/linne_encoder.c:53-138 */
LINNEApiResult LINNEEncoder_EncodeHeader(
    const struct LINNEHeader *header, uint8_t *data, uint32_t data_size);

/* reference: linne_encoder.c:201-265 (returns -1 for an invalid configuration) */
int32_t LINNEEncoder_CalculateWorkSize(const struct LINNEEncoderConfig *config);

/* reference: linne_encoder.c:268-396; (work == NULL && work_size == 0) => the library allocates */
struct LINNEEncoder *LINNEEncoder_Create(const struct LINNEEncoderConfig *config, void *work, int32_t work_size);

/* reference: linne_encoder.c:399-407 */
void LINNEEncoder_Destroy(struct LINNEEncoder *encoder);

/* reference: linne_encoder.c:410-477 */
LINNEApiResult LINNEEncoder_SetEncodeParameter(
    struct LINNEEncoder *encoder, const struct LINNEEncodeParameter *parameter);

/* reference: linne_encoder.c:774-862; one block, synchronous */
LINNEApiResult LINNEEncoder_EncodeBlock(
    struct LINNEEncoder *encoder,
    const int32_t *const *input, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size);

/* reference: linne_encoder.c:865-932; header + every block.  All blocks of the stream are analysed on the
 * GPU as one batch. */
LINNEApiResult LINNEEncoder_EncodeWhole(
    struct LINNEEncoder *encoder,
    const int32_t *const *input, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size);

#ifdef __cplusplus
}
#endif

#endif /* LINNE_ENCODER_H_INCLUDED */

/*
 * linne.h -- common types of the LINNE public API, as exported by liblinne_amd.so.
 *
 * Drop-in boundary: declarations are ABI-identical to the reference's include/linne.h:6-51 (same
 * constant values, enum order and struct member order), so tools/linne_codec/linne_codec.c compiles
 * and links against this library unchanged.
 */
#ifndef LINNE_H_INCLUDED
#define LINNE_H_INCLUDED

#include "linne_stdint.h"

#define LINNE_FORMAT_VERSION        1   /* .lnn container version                  */
#define LINNE_CODEC_VERSION         2   /* encoder version written to the header   */
#define LINNE_HEADER_SIZE           30  /* bytes                                   */
#define LINNE_MAX_NUM_CHANNELS      8
#define LINNE_NUM_PARAMETER_PRESETS 8   /* -m 0 .. -m 7                            */

typedef enum LINNEApiResultTag {
    LINNE_APIRESULT_OK = 0,
    LINNE_APIRESULT_INVALID_ARGUMENT,
    LINNE_APIRESULT_INVALID_FORMAT,
    LINNE_APIRESULT_INSUFFICIENT_BUFFER,
    LINNE_APIRESULT_INSUFFICIENT_DATA,
    LINNE_APIRESULT_PARAMETER_NOT_SET,
    LINNE_APIRESULT_DETECT_DATA_CORRUPTION,
    LINNE_APIRESULT_NG
} LINNEApiResult;

typedef enum LINNEChannelProcessMethodTag {
    LINNE_CH_PROCESS_METHOD_NONE = 0,
    LINNE_CH_PROCESS_METHOD_MS,         /* mid/side on channels 0 and 1 */
    LINNE_CH_PROCESS_METHOD_INVALID
} LINNEChannelProcessMethod;

struct LINNEHeader {
    uint32_t format_version;
    uint32_t codec_version;
    uint16_t num_channels;
    uint32_t num_samples;               /* per channel, whole stream */
    uint32_t sampling_rate;
    uint16_t bits_per_sample;
    uint32_t num_samples_per_block;
    uint8_t preset;
    LINNEChannelProcessMethod ch_process_method;
};

#endif /* LINNE_H_INCLUDED */

/* lnn_host.h -- internal declarations of the host C side of liblinne_amd.so */
#ifndef LNN_HOST_H_INCLUDED
#define LNN_HOST_H_INCLUDED

#include <stdint.h>
#include "linne_amd.h"
#include "lnn_common.h"

/* block types (libs/linne_internal/include/linne_internal.h:47-52) */
#define LNN_BLOCK_COMPRESS 0u
#define LNN_BLOCK_SILENT   1u
#define LNN_BLOCK_RAW      2u

struct lnn_layers {
    uint32_t num_layers, size[LINNE_AMD_MAX_LAYERS], offset[LINNE_AMD_MAX_LAYERS], total, max_size;
    uint32_t num_regs; double regs[4];
};

#define LNN_MAX_DEVICES 16
/* "0,1,2" -> device list (lnn_multi.c); returns the count, 0 when text is NULL or empty */
uint32_t lnn_parse_device_list(const char *text, int *devices, uint32_t max);

void lnn_tables_init(void);
uint16_t lnn_crc16(const uint8_t *data, uint64_t size);
int lnn_shape_layers(const struct LINNEAmdShape *shape, struct lnn_layers *out);
uint32_t lnn_decide_block_type(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, uint32_t n,
        const int32_t *pcm_frame, const double *stats_frame, double *state);
void lnn_parallel_for(uint32_t count, uint32_t num_threads, void (*fn)(void *arg, uint32_t first, uint32_t count), void *arg);
int lnn_parse_block(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, const uint8_t *data, uint64_t avail,
        int check_crc, uint32_t max_samples, uint32_t *type_out, uint32_t *n_out, uint32_t *consumed_out,
        int32_t *samples, int32_t *params);

int lnn_parse_block_head(const struct LINNEAmdShape *shape, const struct lnn_layers *ly, const uint8_t *data, uint64_t avail,
        int check_crc, uint32_t max_samples, uint32_t *type_out, uint32_t *n_out, uint32_t *consumed_out,
        int32_t *samples, int32_t *params, uint64_t *rice_bit_out);

#endif

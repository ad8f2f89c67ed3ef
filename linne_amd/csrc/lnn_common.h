/* lnn_common.h -- internal declarations shared by the host C code and the HIP translation unit. */
#ifndef LNN_COMMON_H_INCLUDED
#define LNN_COMMON_H_INCLUDED

#include <stdint.h>

/* numeric values of LINNEApiResult (include/linne.h) */
#define LNN_OK                      0
#define LNN_INVALID_ARGUMENT        1
#define LNN_INVALID_FORMAT          2
#define LNN_INSUFFICIENT_BUFFER     3
#define LNN_INSUFFICIENT_DATA       4
#define LNN_PARAMETER_NOT_SET       5
#define LNN_DETECT_DATA_CORRUPTION  6
#define LNN_NG                      7

#ifdef __cplusplus
extern "C" {
#endif

/* parameter presets (libs/linne_internal/src/linne_internal.c:16-41): layer sizes and ridge regularisers */
int lnn_preset_info(uint32_t preset, uint32_t *num_layers, uint32_t *layers, uint32_t *num_regs, double *regs);

/* steps of the Rice parameter as a function of the partition mean (lnn_entropy.c: located with the host libm): steps[k] is
 * the smallest mean whose parameter is k + 1; returns how many there are (at most 32) */
uint32_t lnn_rice_k2_steps(double *steps);
/* relative half-width of the band around a step inside which the libm expression itself must be evaluated */
#define LNN_RICE_GUARD 1e-9

#ifdef __cplusplus
}
#endif

#endif

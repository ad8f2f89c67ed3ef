/* TEST INFRASTRUCTURE.  Decodes one .lnn file with the REAL reference decoder (CRC check off) built with
 * AddressSanitizer + UBSan (oracle/Makefile target _ref/ref_decode_san): a damaged stream the reference only survives
 * through undefined behaviour -- reads past the end of the data, shifts by the type's width, int overflow in its own
 * bookkeeping -- aborts here, so that tests/golden/make_corrupt_golden.py can tell the damaged streams whose decoded PCM
 * is DEFINED (and must be matched bit for bit) from those where anything goes.  The input is copied into a malloc'ed
 * buffer of exactly its size, so a read past the end is a heap overflow ASan sees.  Own code against the public API
 * (include/linne_decoder.h:23-50); prints "ret <code>" and an FNV-1a-64 of the decoded planes. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "linne_decoder.h"

int main(int argc, char **argv)
{
    FILE *fp;
    long size;
    uint8_t *data;
    struct LINNEHeader header;
    struct LINNEDecoderConfig config;
    struct LINNEDecoder *dec;
    int32_t *planes[LINNE_MAX_NUM_CHANNELS];
    uint32_t ch, s;
    uint64_t h = 0xcbf29ce484222325ull;
    LINNEApiResult ret;
    if (argc < 2 || !(fp = fopen(argv[1], "rb"))) return 2;
    fseek(fp, 0, SEEK_END); size = ftell(fp); fseek(fp, 0, SEEK_SET);
    data = (uint8_t *)malloc((size_t)size);
    if (!data || fread(data, 1, (size_t)size, fp) != (size_t)size) return 2;
    fclose(fp);
    if (LINNEDecoder_DecodeHeader(data, (uint32_t)size, &header) != LINNE_APIRESULT_OK) { printf("ret -1\n"); return 0; }
    config.max_num_channels = header.num_channels; config.max_num_layers = 5; config.max_num_parameters_per_layer = 128; config.check_crc = 0;
    dec = LINNEDecoder_Create(&config, NULL, 0);
    if (!dec) return 2;
    for (ch = 0; ch < header.num_channels; ch++) planes[ch] = (int32_t *)calloc(header.num_samples, sizeof(int32_t));
    ret = LINNEDecoder_DecodeWhole(dec, data, (uint32_t)size, planes, header.num_channels, header.num_samples);
    for (ch = 0; ch < header.num_channels; ch++)
        for (s = 0; s < header.num_samples; s++) {
            uint32_t v = (uint32_t)planes[ch][s]; int b;
            for (b = 0; b < 4; b++) { h ^= (v >> (8 * b)) & 0xFF; h *= 0x100000001b3ull; }
        }
    printf("ret %d fnv %016llx\n", (int)ret, (unsigned long long)h);
    LINNEDecoder_Destroy(dec);
    for (ch = 0; ch < header.num_channels; ch++) free(planes[ch]);
    free(data);
    return 0;
}

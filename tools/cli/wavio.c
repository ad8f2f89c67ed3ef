#include "wavio.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
#define FAIL(...) do { if (err) snprintf(err, err_size, __VA_ARGS__); goto fail; } while (0)

int wav_alloc(struct wav_pcm *pcm)
{
    uint32_t ch;
    pcm->plane = calloc(pcm->num_channels ? pcm->num_channels : 1, sizeof(int32_t *));
    if (!pcm->plane) return -1;
    for (ch = 0; ch < pcm->num_channels; ch++)
        if (!(pcm->plane[ch] = malloc(sizeof(int32_t) * (pcm->num_samples ? pcm->num_samples : 1)))) return -1;
    return 0;
}

void wav_free(struct wav_pcm *pcm)
{
    uint32_t ch;
    if (pcm->plane) { for (ch = 0; ch < pcm->num_channels; ch++) free(pcm->plane[ch]); free(pcm->plane); }
    pcm->plane = NULL;
}

int wav_read(const char *path, struct wav_pcm *out, char *err, unsigned err_size)
{
    FILE *fp = fopen(path, "rb");
    uint8_t hdr[12], ck[8], fmt[40], *raw = NULL;
    int have_fmt = 0;
    uint32_t tag = 0, block_align = 0, data_bytes = 0, s, ch;
    memset(out, 0, sizeof(*out));
    if (!fp) FAIL("cannot open %s", path);
    if (fread(hdr, 1, 12, fp) != 12 || memcmp(hdr, "RIFF", 4) != 0 || memcmp(hdr + 8, "WAVE", 4) != 0) FAIL("%s: not a RIFF/WAVE file", path);
    for (;;) {
        uint32_t size;
        if (fread(ck, 1, 8, fp) != 8) FAIL("%s: no data chunk", path);
        size = le32(ck + 4);
        if (memcmp(ck, "fmt ", 4) == 0) {
            const uint32_t take = size < sizeof(fmt) ? size : (uint32_t)sizeof(fmt);
            if (size < 16 || fread(fmt, 1, take, fp) != take) FAIL("%s: bad fmt chunk", path);
            if (size > take && fseek(fp, (long)(size - take), SEEK_CUR) != 0) FAIL("%s: bad fmt chunk", path);
            if (size & 1u) (void)fseek(fp, 1, SEEK_CUR);
            tag = le16(fmt); out->num_channels = le16(fmt + 2); out->sampling_rate = le32(fmt + 4);
            block_align = le16(fmt + 12); out->bits_per_sample = le16(fmt + 14);
            if (tag == 0xFFFEu && size >= 26) tag = le16(fmt + 24);      /* WAVE_FORMAT_EXTENSIBLE: first word of the sub-format GUID */
            have_fmt = 1;
        } else if (memcmp(ck, "data", 4) == 0) {
            if (!have_fmt) FAIL("%s: data chunk before fmt chunk", path);
            data_bytes = size;
            break;
        } else {
            if (fseek(fp, (long)(size + (size & 1u)), SEEK_CUR) != 0) FAIL("%s: truncated chunk", path);
        }
    }
    if (tag != 1u) FAIL("%s: only linear PCM is supported (format tag %u)", path, tag);
    if (out->num_channels == 0 || (out->bits_per_sample != 8 && out->bits_per_sample != 16 && out->bits_per_sample != 24 && out->bits_per_sample != 32))
        FAIL("%s: unsupported PCM layout (%u channels, %u bits)", path, out->num_channels, out->bits_per_sample);
    if (block_align != out->num_channels * (out->bits_per_sample / 8)) FAIL("%s: block align %u does not match the format", path, block_align);
    out->num_samples = data_bytes / block_align;
    if (!(raw = malloc(data_bytes ? data_bytes : 1))) FAIL("out of memory");
    {
        const size_t got = fread(raw, 1, data_bytes, fp);
        if (got < data_bytes) out->num_samples = (uint32_t)(got / block_align);       /* truncated file: keep the whole frames */
    }
    if (wav_alloc(out) != 0) FAIL("out of memory");
    {
        const uint32_t bytes = out->bits_per_sample / 8;
        const uint8_t *q = raw;
        for (s = 0; s < out->num_samples; s++)
            for (ch = 0; ch < out->num_channels; ch++, q += bytes) {
                int32_t v;
                switch (bytes) {
                case 1: v = (int32_t)q[0] - 128; break;                                  /* 8-bit WAV is offset binary */
                case 2: v = (int16_t)le16(q); break;
                case 3: v = (int32_t)((le16(q) | ((uint32_t)q[2] << 16)) << 8) >> 8; break;
                default: v = (int32_t)le32(q); break;
                }
                out->plane[ch][s] = v;
            }
    }
    free(raw); fclose(fp);
    return 0;
fail:
    free(raw); if (fp) fclose(fp); wav_free(out);
    return -1;
}

int wav_write(const char *path, const struct wav_pcm *pcm, char *err, unsigned err_size)
{
    FILE *fp = fopen(path, "wb");
    const uint32_t bytes = pcm->bits_per_sample / 8, block_align = bytes * pcm->num_channels;
    const uint64_t data_bytes = (uint64_t)pcm->num_samples * block_align;
    uint8_t hdr[44], *raw = NULL, *q;
    uint32_t s, ch;
    if (!fp) FAIL("cannot create %s", path);
    if (data_bytes + 36 > 0xFFFFFFFFull) FAIL("%s: too large for a RIFF file", path);
    memcpy(hdr, "RIFF", 4); put32(hdr + 4, (uint32_t)data_bytes + 36); memcpy(hdr + 8, "WAVEfmt ", 8); put32(hdr + 16, 16);
    put16(hdr + 20, 1); put16(hdr + 22, pcm->num_channels); put32(hdr + 24, pcm->sampling_rate); put32(hdr + 28, pcm->sampling_rate * block_align);
    put16(hdr + 32, block_align); put16(hdr + 34, pcm->bits_per_sample); memcpy(hdr + 36, "data", 4); put32(hdr + 40, (uint32_t)data_bytes);
    if (!(raw = malloc(data_bytes ? (size_t)data_bytes : 1))) FAIL("out of memory");
    q = raw;
    for (s = 0; s < pcm->num_samples; s++)
        for (ch = 0; ch < pcm->num_channels; ch++, q += bytes) {
            const uint32_t v = (uint32_t)pcm->plane[ch][s];
            switch (bytes) {
            case 1: q[0] = (uint8_t)(v + 128u); break;
            case 2: put16(q, v); break;
            case 3: put16(q, v); q[2] = (uint8_t)(v >> 16); break;
            default: put32(q, v); break;
            }
        }
    if (fwrite(hdr, 1, 44, fp) != 44 || fwrite(raw, 1, (size_t)data_bytes, fp) != (size_t)data_bytes) FAIL("%s: write error", path);
    free(raw);
    if (fclose(fp) != 0) { fp = NULL; FAIL("%s: write error", path); }
    return 0;
fail:
    free(raw); if (fp) fclose(fp);
    return -1;
}

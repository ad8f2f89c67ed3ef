/* wavio.h -- minimal RIFF/WAVE PCM reader and writer for the command line tool (own implementation; the layout written
 * is the canonical 44-byte header the reference's libs/wav writes, wav.c:523-650, so decoded files compare equal). */
#ifndef WAVIO_H_INCLUDED
#define WAVIO_H_INCLUDED
#include <stdint.h>

struct wav_pcm {
    uint32_t num_channels, sampling_rate, bits_per_sample, num_samples;
    int32_t **plane;            /* [num_channels][num_samples], right-justified signed samples */
};

/* returns 0 on success; on failure writes a message to err (if not NULL) */
int wav_read(const char *path, struct wav_pcm *out, char *err, unsigned err_size);
int wav_write(const char *path, const struct wav_pcm *pcm, char *err, unsigned err_size);
int wav_alloc(struct wav_pcm *pcm);       /* allocates plane[][] from the format fields */
void wav_free(struct wav_pcm *pcm);

#endif

/*
 * lnn_multi.c -- one process, several GPUs: fans a batch of frames out over per-GPU contexts (SURVEY.md section 8e,
 * "direct per-GPU H2D/D2H" variant).
 *
 * The reference has no equivalent (its caller is a single-threaded block loop, tools/linne_codec/linne_codec.c:133-161);
 * frames are independent on the prediction path (libs/linne_encoder/src/linne_encoder.c:637: nothing is carried from block
 * to block), so a batch is cut into groups of frames, group g goes to device g mod G, and nothing is ever exchanged
 * between GPUs.  One host thread per device drives that device's staging slots (pinned host buffers, H2D on the GPU's own
 * PCIe link, kernels, D2H: lnn_device.hip LINNEAmd_Slot*), two or three groups in flight per device, so that all links and
 * all GPUs work at once.  Results land in the caller's arrays in the caller's frame order.
 *
 * Host code is C like the reference; the kernels are reached through the C-ABI of include/linne_amd.h only.
 */
#define _GNU_SOURCE
#include "lnn_host.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MULTI_SLOTS 3u

struct LINNEAmdMulti {
    uint32_t ndev;
    int device[LNN_MAX_DEVICES];
    struct LINNEAmdContext *ctx[LNN_MAX_DEVICES];
    struct LINNEAmdSlot *slot[LNN_MAX_DEVICES][MULTI_SLOTS];
    struct LINNEAmdShape slot_shape; uint32_t slot_frames; int slot_for_encode;
    char err[256];
};

/* "0,1,2" -> device list; returns the count (0 if the variable is unset or empty) */
uint32_t lnn_parse_device_list(const char *text, int *devices, uint32_t max)
{
    uint32_t n = 0;
    if (!text) return 0;
    while (*text && n < max) {
        char *end = NULL;
        const long v = strtol(text, &end, 10);
        if (end == text) break;
        devices[n++] = (int)v;
        text = end;
        while (*text == ',' || *text == ' ') text++;
    }
    return n;
}

struct LINNEAmdMulti *LINNEAmd_MultiCreate(const int *devices, uint32_t num_devices, uint64_t scratch_bytes_per_device)
{
    struct LINNEAmdMulti *m;
    int list[LNN_MAX_DEVICES];
    uint32_t i;
    if (devices == NULL || num_devices == 0) {              /* LINNE_AMD_DEVICES, else every visible device */
        num_devices = lnn_parse_device_list(getenv("LINNE_AMD_DEVICES"), list, LNN_MAX_DEVICES);
        if (num_devices == 0) {
            const int n = LINNEAmd_GetDeviceCount();
            for (i = 0; i < (uint32_t)n && i < LNN_MAX_DEVICES; i++) list[i] = (int)i;
            num_devices = i;
        }
        devices = list;
    }
    if (num_devices == 0 || num_devices > LNN_MAX_DEVICES) { fprintf(stderr, "liblinne_amd: MultiCreate: %u devices (1..%d)\n", num_devices, LNN_MAX_DEVICES); return NULL; }
    m = calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->ndev = num_devices;
    for (i = 0; i < num_devices; i++) {
        m->device[i] = devices[i];
        m->ctx[i] = LINNEAmd_ContextCreate(devices[i], scratch_bytes_per_device ? scratch_bytes_per_device : (2ull << 30));
        if (!m->ctx[i]) { LINNEAmd_MultiDestroy(m); return NULL; }        /* ContextCreate said why; there is no CPU fallback */
    }
    return m;
}

static void multi_drop_slots(struct LINNEAmdMulti *m)
{
    uint32_t d, i;
    for (d = 0; d < m->ndev; d++)
        for (i = 0; i < MULTI_SLOTS; i++) { if (m->slot[d][i]) LINNEAmd_SlotDestroy(m->slot[d][i]); m->slot[d][i] = NULL; }
    m->slot_frames = 0;
}

void LINNEAmd_MultiDestroy(struct LINNEAmdMulti *m)
{
    uint32_t d;
    if (!m) return;
    multi_drop_slots(m);
    for (d = 0; d < m->ndev; d++) if (m->ctx[d]) LINNEAmd_ContextDestroy(m->ctx[d]);
    free(m);
}

uint32_t LINNEAmd_MultiNumDevices(const struct LINNEAmdMulti *m) { return m ? m->ndev : 0; }
int LINNEAmd_MultiDevice(const struct LINNEAmdMulti *m, uint32_t index) { return (m && index < m->ndev) ? m->device[index] : -1; }
struct LINNEAmdContext *LINNEAmd_MultiContext(struct LINNEAmdMulti *m, uint32_t index) { return (m && index < m->ndev) ? m->ctx[index] : NULL; }
const char *LINNEAmd_MultiGetLastError(const struct LINNEAmdMulti *m) { return m ? m->err : "no handle"; }

/* group g of the batch: frames [g * group, ...) -> device g mod ndev (round-robin) */
struct multi_job {
    struct LINNEAmdMulti *m; uint32_t dev;
    const struct LINNEAmdShape *shape;
    const int32_t *pcm; int32_t *data; int32_t *params; double *stats; uint8_t *plan;
    const uint32_t *num_samples; uint32_t num_frames, group, ngroups; int for_encode;
    int ret;
    char err[256];                       /* this device's message; multi_run copies the first failing device's into m->err after the joins */
};

static void *multi_worker(void *arg)
{
    struct multi_job *j = arg;
    struct LINNEAmdMulti *m = j->m;
    const uint32_t d = j->dev, G = m->ndev, C = j->shape->num_channels, S = j->shape->num_samples_per_block;
    const uint64_t fb = sizeof(int32_t) * (uint64_t)C * S, pb = sizeof(int32_t) * (uint64_t)C * LINNE_AMD_PARAM_WORDS,
                   sb = sizeof(double) * (uint64_t)C * LINNE_AMD_STAT_WORDS, qb = (uint64_t)C * LINNE_AMD_RICE_PLAN_BYTES;
    uint32_t mine = 0, submitted = 0, done = 0, g;
    uint32_t *full = NULL;
    for (g = d; g < j->ngroups; g += G) mine++;
    if (!j->num_samples) {                                   /* every frame full */
        uint32_t f;
        if (!(full = malloc(sizeof(uint32_t) * j->group))) { j->ret = LNN_NG; snprintf(j->err, sizeof(j->err), "device %d: out of host memory", m->device[d]); return NULL; }
        for (f = 0; f < j->group; f++) full[f] = S;
    }
    j->ret = LNN_OK;
    while (done < mine && j->ret == LNN_OK) {
        while (submitted < mine && submitted - done < MULTI_SLOTS && j->ret == LNN_OK) {
            struct LINNEAmdSlot *sl = m->slot[d][submitted % MULTI_SLOTS];
            const uint32_t base = (d + submitted * G) * j->group, cnt = (j->num_frames - base < j->group) ? (j->num_frames - base) : j->group;
            const uint32_t *ns = j->num_samples ? j->num_samples + base : full;
            if (j->for_encode) {
                memcpy(LINNEAmd_SlotPcm(sl), j->pcm + (uint64_t)base * C * S, fb * cnt);
                j->ret = LINNEAmd_SlotEncodeSubmit(sl, ns, cnt);
            } else {
                memcpy(LINNEAmd_SlotData(sl), j->data + (uint64_t)base * C * S, fb * cnt);
                memcpy(LINNEAmd_SlotParams(sl), j->params + (uint64_t)base * C * LINNE_AMD_PARAM_WORDS, pb * cnt);
                j->ret = LINNEAmd_SlotDecodeSubmit(sl, ns, cnt);
            }
            if (j->ret != LNN_OK) snprintf(j->err, sizeof(j->err), "device %d: %s", m->device[d], LINNEAmd_GetLastError(m->ctx[d]));
            submitted++;
        }
        if (j->ret != LNN_OK) break;
        {
            struct LINNEAmdSlot *sl = m->slot[d][done % MULTI_SLOTS];
            const uint32_t base = (d + done * G) * j->group, cnt = (j->num_frames - base < j->group) ? (j->num_frames - base) : j->group;
            if ((j->ret = LINNEAmd_SlotWait(sl)) != LNN_OK) { snprintf(j->err, sizeof(j->err), "device %d: %s", m->device[d], LINNEAmd_GetLastError(m->ctx[d])); break; }
            memcpy(j->data + (uint64_t)base * C * S, LINNEAmd_SlotData(sl), fb * cnt);
            if (j->for_encode) {
                memcpy(j->params + (uint64_t)base * C * LINNE_AMD_PARAM_WORDS, LINNEAmd_SlotParams(sl), pb * cnt);
                memcpy(j->stats + (uint64_t)base * C * LINNE_AMD_STAT_WORDS, LINNEAmd_SlotStats(sl), sb * cnt);
                if (j->plan) memcpy(j->plan + (uint64_t)base * qb, LINNEAmd_SlotRicePlan(sl), qb * cnt);
            }
            done++;
        }
    }
    {   /* nothing of this call may stay in flight */
        uint32_t i;
        for (i = 0; i < MULTI_SLOTS; i++) if (m->slot[d][i]) (void)LINNEAmd_SlotWait(m->slot[d][i]);
    }
    free(full);
    return NULL;
}

static int multi_run(struct LINNEAmdMulti *m, const struct LINNEAmdShape *shape, const int32_t *pcm, int32_t *data, int32_t *params,
        double *stats, uint8_t *plan, const uint32_t *num_samples, uint32_t num_frames, uint32_t group, int for_encode)
{
    struct multi_job job[LNN_MAX_DEVICES];
    pthread_t th[LNN_MAX_DEVICES];
    uint32_t d, i, started = 0, ngroups;
    int ret = LNN_OK;
    if (!m) return LNN_INVALID_ARGUMENT;
    m->err[0] = 0;
    if (!shape || !data || !params || (for_encode && (!pcm || !stats))) { snprintf(m->err, sizeof(m->err), "null argument"); return LNN_INVALID_ARGUMENT; }
    if (num_frames == 0) return LNN_OK;
    if (group == 0) {                                       /* a few groups per device, each large enough to be throughput-bound */
        group = (num_frames + m->ndev * 4u - 1u) / (m->ndev * 4u);
        if (group < 256u) group = 256u;
        if (group > 2048u) group = 2048u;
    }
    if (group > num_frames) group = num_frames;
    ngroups = (num_frames + group - 1u) / group;
    if (memcmp(&m->slot_shape, shape, sizeof(*shape)) != 0 || m->slot_frames < group || m->slot_for_encode != for_encode) {
        multi_drop_slots(m);
        m->slot_shape = *shape; m->slot_frames = group; m->slot_for_encode = for_encode;
    }
    for (d = 0; d < m->ndev; d++) {
        uint32_t mine = 0, g;
        for (g = d; g < ngroups; g += m->ndev) mine++;
        for (i = 0; i < MULTI_SLOTS && i < mine; i++)
            if (!m->slot[d][i] && !(m->slot[d][i] = LINNEAmd_SlotCreate(m->ctx[d], shape, m->slot_frames, for_encode))) {
                snprintf(m->err, sizeof(m->err), "device %d: SlotCreate: %s", m->device[d], LINNEAmd_GetLastError(m->ctx[d]));
                multi_drop_slots(m);                        /* no half-created set stays behind: the next call starts from nothing */
                return LNN_NG;
            }
    }
    for (d = 0; d < m->ndev; d++) {
        struct multi_job *j = &job[d];
        memset(j, 0, sizeof(*j));
        j->m = m; j->dev = d; j->shape = shape; j->pcm = pcm; j->data = data; j->params = params; j->stats = stats; j->plan = plan;
        j->num_samples = num_samples; j->num_frames = num_frames; j->group = group; j->ngroups = ngroups; j->for_encode = for_encode;
        if (d + 1 == m->ndev) break;                        /* the calling thread serves the last device */
        if (pthread_create(&th[d], NULL, multi_worker, j) != 0) { j->ret = LNN_NG; ret = LNN_NG; snprintf(m->err, sizeof(m->err), "device %d: pthread_create failed", m->device[d]); break; }
        started++;
    }
    if (ret == LNN_OK) multi_worker(&job[m->ndev - 1]);
    for (d = 0; d < started; d++) pthread_join(th[d], NULL);
    for (d = 0; d < m->ndev && ret == LNN_OK; d++) if (job[d].ret != LNN_OK) { ret = job[d].ret; snprintf(m->err, sizeof(m->err), "%s", job[d].err); }
    return ret;
}

int LINNEAmd_MultiEncodeFramesHost(struct LINNEAmdMulti *m, const struct LINNEAmdShape *shape, const int32_t *pcm,
        const uint32_t *num_samples, uint32_t num_frames, int32_t *residual, int32_t *params, double *stats, uint8_t *rice_plan,
        uint32_t group_frames)
{
    return multi_run(m, shape, pcm, residual, params, stats, rice_plan, num_samples, num_frames, group_frames, 1);
}

int LINNEAmd_MultiDecodeFramesHost(struct LINNEAmdMulti *m, const struct LINNEAmdShape *shape, int32_t *data,
        const uint32_t *num_samples, uint32_t num_frames, const int32_t *params, uint32_t group_frames)
{
    return multi_run(m, shape, NULL, data, (int32_t *)params, NULL, NULL, num_samples, num_frames, group_frames, 0);
}

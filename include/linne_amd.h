/*
 * linne_amd.h -- C-ABI of the MI355X (gfx950) per-frame prediction path, batch form.
 *
 * The reference has no FFI for this path: its caller reaches it through LINNEEncoder_EncodeBlock /
 * LINNEDecoder_DecodeBlock (include/linne_encoder.h:49-54, include/linne_decoder.h:38-43), one block per
 * call.  liblinne_amd.so keeps those 13 public symbols (include/linne_encoder.h, include/linne_decoder.h
 * in this directory) and adds the entry points below, which are what those functions call internally and
 * what a maintainer would bind to feed many frames at once.  Plain pointers and sizes
 * only; no torch types.  Every function returns a LINNEApiResult value (0 = OK) unless noted.
 *
 * Units: a "frame" is one LINNE block (num_samples_per_block samples per channel); a "channel-frame" is
 * one channel of one frame -- the independent unit of work.
 *
 * Data layout (all device buffers are SoA, frame-major, planar):
 *   pcm / residual : int32_t [num_frames][num_channels][stride]        stride = num_samples_per_block
 *   params         : int32_t [num_frames][num_channels][LINNE_AMD_PARAM_WORDS]
 *   stats          : double  [num_frames][num_channels][LINNE_AMD_STAT_WORDS]
 */
#ifndef LINNE_AMD_H_INCLUDED
#define LINNE_AMD_H_INCLUDED

#include <stdint.h>

#define LINNE_AMD_MAX_LAYERS      3
#define LINNE_AMD_MAX_PARAMS      128
#define LINNE_AMD_PARAM_WORDS     160     /* per channel-frame, see offsets below */
#define LINNE_AMD_STAT_WORDS      8

/* params record (int32 words) of one channel-frame:
 *   [0..1]  pre-emphasis prev (first sample of each stage's input; linne_encoder.c:637,707-709)
 *   [2..3]  pre-emphasis coefficient, 0..15            (linne_utility.c:158-193)
 *   [4..6]  number of units per layer (power of two)   (linne_network.c:268-347)
 *   [7..9]  coefficient right shift per layer          (lpc.c:981-1040)
 *   [10..]  quantised coefficients, layers back to back (sum of the preset's layer sizes <= 148),
 *           in filter order (index 0 multiplies the oldest sample; linne_network.c:310-316) */
#define LINNE_AMD_PRM_PREV    0
#define LINNE_AMD_PRM_PCOEF   2
#define LINNE_AMD_PRM_UNITS   4
#define LINNE_AMD_PRM_RSHIFT  7
#define LINNE_AMD_PRM_COEF    10

/* stats record (doubles) of one channel-frame, inputs of the host-side block-type decision
 * (linne_encoder.c:480-529, lpc.c:810-865):
 *   [0] r0        SIN-window autocorrelation lag 0 of the raw channel
 *   [1..3] k1..k3 PARCOR coefficients 1..order-1 of that analysis (order = layer-0 size)
 *   [4] zero_path 1.0 if that Levinson call took the all-zero branch (lpc.c:271-276), else 0.0
 *   [5] tail      value the channel's analysis leaves in parcor[order] (oracle quirk Q2)
 *   [6] best_pass index of the winning regulariser
 *   [7] loss      its L1 loss */
#define LINNE_AMD_ST_R0     0
#define LINNE_AMD_ST_K1     1
#define LINNE_AMD_ST_ZERO   4
#define LINNE_AMD_ST_TAIL   5
#define LINNE_AMD_ST_BEST   6
#define LINNE_AMD_ST_LOSS   7

struct LINNEAmdShape {                  /* batch-wide stream parameters (struct LINNEEncodeParameter) */
    uint32_t num_channels;
    uint32_t bits_per_sample;
    uint32_t num_samples_per_block;     /* = row stride of pcm / residual */
    uint32_t preset;
    uint32_t ch_process_method;         /* 0 none, 1 mid/side */
};

struct LINNEAmdContext;

#ifdef __cplusplus
extern "C" {
#endif

/* number of visible HIP devices (0 when there is none); never fails */
int LINNEAmd_GetDeviceCount(void);

/* Creates a context on `device`: a scratch arena of about scratch_bytes (0 = default 6 GiB, grown on
 * demand) and the stream work is issued on.  Returns NULL when the HIP runtime or the device is
 * unavailable -- there is no CPU fallback. */
struct LINNEAmdContext *LINNEAmd_ContextCreate(int device, uint64_t scratch_bytes);
void LINNEAmd_ContextDestroy(struct LINNEAmdContext *ctx);
/* message of the last failure on this context ("" if none) */
const char *LINNEAmd_GetLastError(const struct LINNEAmdContext *ctx);
/* grows the scratch arena to at least `bytes` (bigger arena = more frames per launch) */
int LINNEAmd_ReserveScratch(struct LINNEAmdContext *ctx, uint64_t bytes);
/* scratch one frame of this shape needs inside EncodeFramesDevice (0 for an invalid shape): frames x this = the arena that holds a
 * batch in one launch chunk */
uint64_t LINNEAmd_ScratchBytesPerFrame(const struct LINNEAmdShape *shape);
/* issue all subsequent work on an existing hipStream_t (e.g. torch's current stream); NULL names the device's
 * default (null) stream.  Until this is called the context uses a stream of its own. */
int LINNEAmd_SetStream(struct LINNEAmdContext *ctx, void *hip_stream);

/* `-a N` (struct LINNEEncodeParameter.num_afmethod_iterations; lpc.c:578-633): the number of auxiliary-function iterations that
 * refine every layer's coefficients in the final pass of LINNENetwork_SetUnitsAndParameters (linne_network.c:605-630).  Applies to
 * the following EncodeFramesDevice / Host calls of this context; 0 (the default) = off.  With N > 0 a call synchronises the
 * host: every Cholesky pivot's pow(x, -0.5) is taken from the host's libm, whose bits no device routine can promise. */
int LINNEAmd_SetAfIterations(struct LINNEAmdContext *ctx, uint32_t iterations);
/* `-l` (struct LINNEEncodeParameter.enable_learning; linne_encoder.c:669-675): after the analysis every channel-frame's parameters go
 * through LINNENetworkTrainer_Train (linne_network.c:805-873: up to 2000 momentum-SGD steps on the L1 loss).  Synchronises the host. */
int LINNEAmd_SetLearning(struct LINNEAmdContext *ctx, uint32_t enable);

/* ENCODE hot path, device resident.  Replaces, for every frame of the batch, the numeric core of
 * LINNEEncoder_EncodeCompressData (linne_encoder.c:613-696) and the analysis half of
 * LINNEEncoder_DecideBlockDataType (linne_encoder.c:494-503):
 *   MS (linne_utility.c:120-132) -> 2x pre-emphasis (:158-212) -> per regulariser { per layer { Welch window,
 *   autocorrelation, Levinson-Durbin (lpc.c:176-366) for every unit count; residual L1 search
 *   (linne_network.c:268-347); forward (linne_network.c:165-210) } } -> best regulariser (linne_network.c:605-630)
 *   -> quantisation (lpc.c:981-1040) -> int32 FIR cascade (linne_lpc_predict.c:7-38).
 * d_pcm: right-justified signed PCM; h_num_samples[f] <= stride is frame f's valid length (host array,
 * NULL = all frames full).  Work is enqueued on the context's stream; the call returns without
 * synchronising unless it has to grow the arena. */
int LINNEAmd_EncodeFramesDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_pcm, const uint32_t *h_num_samples, uint32_t num_frames,
        int32_t *d_residual, int32_t *d_params, double *d_stats);

/* DECODE hot path, device resident, in place: d_data holds the entropy-decoded residual on entry and PCM on
 * return.  Replaces linne_decoder.c:503-522: per channel the int32 synthesis cascade in reverse layer order
 * (linne_lpc_synthesize.c:8-83), two-stage de-emphasis (linne_utility.c:215-241), then MS->LR
 * (linne_utility.c:135-147).  d_params' coefficients must lie in [-128, 127], the range the stream's 8-bit coefficient code can
 * express (lnn_parse_block delivers nothing else): the synthesis kernels multiply them on 8-bit / exact-FP64 paths. */
int LINNEAmd_DecodeFramesDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        int32_t *d_data, const uint32_t *h_num_samples, uint32_t num_frames, const int32_t *d_params);

/* Same two paths on host buffers (H2D, kernels, D2H, synchronous); what EncodeBlock / DecodeBlock use.  Where the parameter records
 * are in HOST memory -- here and in the staging slots' decode submits -- the [-128, 127] contract of the coefficients is CHECKED:
 * a record outside it is refused with LINNE_APIRESULT_INVALID_FORMAT's value (2), so that no result depends on which form of the
 * synthesis the batch size picks. */
int LINNEAmd_EncodeFramesHost(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *pcm, const uint32_t *num_samples, uint32_t num_frames,
        int32_t *residual, int32_t *params, double *stats);
int LINNEAmd_DecodeFramesHost(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        int32_t *data, const uint32_t *num_samples, uint32_t num_frames, const int32_t *params);

/* Rice planning on the device (linne_coder.c:217-279): for every channel-frame of a batch, the
 * partition means of the zig-zagged residual, the parameter of every partition at every partition order, the code length
 * of every order and the argmin -- everything of LINNECoder_EncodePartitionedRecursiveRice except writing the bits.
 * Plan record per channel-frame, LINNE_AMD_RICE_PLAN_BYTES bytes: [0] partition order, [1] 1 if some mean fell inside the
 * guard band of a parameter step (the host then runs its own search for this channel-frame; the libm expression decides),
 * [16 ..] the parameter of each partition of the chosen order.  Enqueues on the context's stream. */
#define LINNE_AMD_RICE_PLAN_BYTES  1040
#define LINNE_AMD_RICE_PLAN_NBITS  4       /* uint32 at this byte offset: length of the channel's whole Rice code in bits (0xFFFFFFFF when flagged) */
#define LINNE_AMD_RICE_PLAN_K2     16
int LINNEAmd_RicePlanDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_residual, const uint32_t *h_num_samples, uint32_t num_frames, uint8_t *d_plan);

/* Rice EMISSION on the device (beyond step 2): writes every channel-frame's partitioned recursive Rice code
 * (linne_coder.c:281-302 with the bit order of bit_stream.h:240-282) from the residual and the plan RicePlanDevice made for the
 * same batch (call it first).  d_offsets [F * C + 1] receives each code's byte offset in d_packed (8-byte aligned; 0xFFFFFFFF
 * for a channel-frame whose plan is flagged or whose code does not fit), the last entry the bytes used.  The host stage then
 * appends the codes at their bit positions instead of coding the residual (LINNEAmd_PackFramesEmitted). */
int LINNEAmd_RiceEmitDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const int32_t *d_residual, uint32_t num_frames, const uint8_t *d_plan,
        uint32_t *d_offsets, uint8_t *d_packed, uint64_t packed_capacity);

/* Rice DECODING on the device (linne_coder.c:306-327): lanes = frames, every lane walks its block's channels from d_bitpos[f] on.
 * d_stream must be 4-byte aligned and readable up to the next multiple of 8 behind stream_bytes.  d_endbit[f] = the bit position
 * behind the frame's last code, ~0 for a frame whose code holds something no encoder writes (decode it on the host). */
int LINNEAmd_RiceDecodeDevice(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        const uint8_t *d_stream, uint64_t stream_bytes, const uint64_t *d_bitpos, const uint32_t *h_num_samples, uint32_t num_frames,
        int32_t *d_residual, uint64_t *d_endbit);

/* Staging slots: what a whole-stream caller (LINNEEncoder_EncodeWhole / LINNEDecoder_DecodeWhole,
 * linne_encoder.c:865-932, linne_decoder.c:671-742) uses instead of the synchronous host forms.  A slot owns pinned
 * host buffers and device buffers for up to max_frames frames of one shape.  The caller fills SlotPcm (encode) or
 * SlotData + SlotParams (decode), submits, and later waits; Submit only enqueues (H2D on a copy stream, the kernels on
 * the context stream, D2H on a second copy stream, chained by events), so rotating over two or three slots overlaps
 * the host entropy stage, PCIe and the kernels.  After SlotWait: encode -> SlotData = residual, SlotParams, SlotStats;
 * decode -> SlotData = PCM.  A slot must be destroyed before its context. */
struct LINNEAmdSlot;
struct LINNEAmdSlot *LINNEAmd_SlotCreate(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        uint32_t max_frames, int for_encode);
/* Encode slots with less PCIe traffic (what LINNEEncoder_EncodeWhole uses):
 *   LINNE_AMD_SLOT_PCM16  the input is staged NARROW (SlotPcm16; LINNEAmd_SlotPcmWidth bytes per sample: int16 up to 16 bits per sample,
 *                         packed little-endian 3-byte samples up to 24, ignored above) and widened on the device;
 *   LINNE_AMD_SLOT_EMIT   the device also WRITES the residual's Rice code (linne_coder.c:281-302; LINNEAmd_RiceEmitDevice): after
 *                         SlotWait, SlotPacked holds the channels' codes back to back, SlotOffsets[cf] the byte offset of
 *                         channel-frame cf's code (0xFFFFFFFF: not emitted -- flagged plan or no room; the host then fetches that
 *                         frame's residual with SlotFetchResidual and codes it itself), SlotRicePlan the plans with each code's
 *                         bit length at LINNE_AMD_RICE_PLAN_NBITS.  The residual is not copied to the host (SlotData is NULL). */
#define LINNE_AMD_SLOT_PCM16  1u
#define LINNE_AMD_SLOT_EMIT   2u
#define LINNE_AMD_SLOT_STREAM 4u       /* decode slots: see LINNEAmd_SlotDecodeStreamSubmit */
struct LINNEAmdSlot *LINNEAmd_SlotCreateEx(struct LINNEAmdContext *ctx, const struct LINNEAmdShape *shape,
        uint32_t max_frames, int for_encode, uint32_t flags);
uint32_t  LINNEAmd_SlotFlags(const struct LINNEAmdSlot *slot);
int16_t  *LINNEAmd_SlotPcm16(struct LINNEAmdSlot *slot);                   /* [F][C][S] int16 (NULL unless LINNE_AMD_SLOT_PCM16 took effect; SlotPcm is NULL then) */
const uint8_t  *LINNEAmd_SlotPacked(struct LINNEAmdSlot *slot);
const uint32_t *LINNEAmd_SlotOffsets(struct LINNEAmdSlot *slot);           /* [F * C + 1], the last entry = bytes used */
int LINNEAmd_SlotFetchResidual(struct LINNEAmdSlot *slot, uint32_t frame, int32_t *dst /* [C][S] */);
/* Decode slots with less PCIe traffic and no Rice decoding on the host (what LINNEDecoder_DecodeWhole uses for streams whose CRCs
 * it checks): created with LINNE_AMD_SLOT_STREAM (| LINNE_AMD_SLOT_PCM16 for <= 16-bit audio).  The caller copies the bytes of a group
 * of blocks into SlotStream, sets SlotBitPos[f] = the bit offset in that buffer at which frame f's first channel's Rice code starts
 * (behind the parameter bits; lnn_parse_block_head), fills SlotParams, and submits: H2D, LINNEAmd_RiceDecodeDevice, the synthesis
 * kernels, D2H.  After SlotWait: SlotEndBits[f] = the bit position behind frame f's last code (from which the bytes the block
 * consumed follow, linne_decoder.c:495-499), or ~0 when the stream held something no encoder writes -- the host must then decode
 * the group itself (SlotDecodeSubmit); PCM in SlotData, or in SlotPcm16 when SlotPcm16Valid (every sample fitted). */
uint8_t  *LINNEAmd_SlotStream(struct LINNEAmdSlot *slot);
uint64_t  LINNEAmd_SlotStreamCapacity(const struct LINNEAmdSlot *slot);
uint64_t *LINNEAmd_SlotBitPos(struct LINNEAmdSlot *slot);
/* [max_frames] where every block ends (bit position in the slot's stream buffer): the device's decoder reads a block's codes no further */
uint64_t *LINNEAmd_SlotBitEnd(struct LINNEAmdSlot *slot);
const uint64_t *LINNEAmd_SlotEndBits(struct LINNEAmdSlot *slot);
int LINNEAmd_SlotPcm16Valid(const struct LINNEAmdSlot *slot);
/* bytes per staged PCM sample of this slot: 4 (int32), 2 (int16) or 3 (packed little-endian: a LINNE_AMD_SLOT_PCM16 slot of 17 .. 24-bit
 * audio; SlotPcm16 then points at bytes) */
uint32_t LINNEAmd_SlotPcmWidth(const struct LINNEAmdSlot *slot);
int LINNEAmd_SlotDecodeStreamSubmit(struct LINNEAmdSlot *slot, uint64_t stream_bytes, const uint32_t *num_samples, uint32_t num_frames);
int LINNEAmd_SlotFetchPcm32(struct LINNEAmdSlot *slot, uint32_t num_frames);
void      LINNEAmd_SlotDestroy(struct LINNEAmdSlot *slot);
int32_t  *LINNEAmd_SlotPcm(struct LINNEAmdSlot *slot);       /* [F][C][S] int32, encode input (NULL for a decode slot) */
int32_t  *LINNEAmd_SlotData(struct LINNEAmdSlot *slot);      /* [F][C][S] int32, residual (encode out, decode in) / PCM (decode out) */
int32_t  *LINNEAmd_SlotParams(struct LINNEAmdSlot *slot);    /* [F][C][LINNE_AMD_PARAM_WORDS] */
double   *LINNEAmd_SlotStats(struct LINNEAmdSlot *slot);     /* [F][C][LINNE_AMD_STAT_WORDS] (NULL for a decode slot) */
uint8_t  *LINNEAmd_SlotRicePlan(struct LINNEAmdSlot *slot);  /* [F][C][LINNE_AMD_RICE_PLAN_BYTES] (NULL for a decode slot) */
uint32_t  LINNEAmd_SlotCapacity(const struct LINNEAmdSlot *slot);
int LINNEAmd_SlotEncodeSubmit(struct LINNEAmdSlot *slot, const uint32_t *num_samples, uint32_t num_frames);
int LINNEAmd_SlotDecodeSubmit(struct LINNEAmdSlot *slot, const uint32_t *num_samples, uint32_t num_frames);
int LINNEAmd_SlotWait(struct LINNEAmdSlot *slot);

/* Several GPUs from ONE process (the "direct per-GPU H2D/D2H" transport).  The reference has nothing like
 * it: its caller loops over blocks in one thread (tools/linne_codec/linne_codec.c:133-161).  Frames are independent on this path
 * (linne_encoder.c:637), so a batch on host memory is cut into groups of group_frames frames (0 = a default that gives every
 * device a few throughput-sized groups), group g goes to device g mod G, one host thread per device drives that device's staging
 * slots -- H2D and D2H on each GPU's own PCIe link, two or three groups in flight per device -- and the results land in the
 * caller's arrays in the caller's frame order.  No data ever moves between GPUs.
 *   devices == NULL / num_devices == 0: LINNE_AMD_DEVICES="0,1,..." if set, else every visible device.  The same device may be
 *   listed more than once (two contexts on one GPU: what the one-GPU tests do).
 * The whole-stream API functions (LINNEEncoder_EncodeWhole / LINNEDecoder_DecodeWhole) fan out the same way when
 * LINNE_AMD_DEVICES names several devices; their .lnn bytes do not depend on it. */
struct LINNEAmdMulti;
struct LINNEAmdMulti *LINNEAmd_MultiCreate(const int *devices, uint32_t num_devices, uint64_t scratch_bytes_per_device);
void LINNEAmd_MultiDestroy(struct LINNEAmdMulti *multi);
uint32_t LINNEAmd_MultiNumDevices(const struct LINNEAmdMulti *multi);
int LINNEAmd_MultiDevice(const struct LINNEAmdMulti *multi, uint32_t index);                       /* HIP device id of member `index`, -1 if out of range */
struct LINNEAmdContext *LINNEAmd_MultiContext(struct LINNEAmdMulti *multi, uint32_t index);       /* the member's context (timing, telemetry) */
const char *LINNEAmd_MultiGetLastError(const struct LINNEAmdMulti *multi);
/* pcm / residual [F][C][S], params [F][C][LINNE_AMD_PARAM_WORDS], stats [F][C][LINNE_AMD_STAT_WORDS] on the host;
 * rice_plan (may be NULL) [F][C][LINNE_AMD_RICE_PLAN_BYTES].  Synchronous.  Returns LINNEApiResult. */
int LINNEAmd_MultiEncodeFramesHost(struct LINNEAmdMulti *multi, const struct LINNEAmdShape *shape, const int32_t *pcm,
        const uint32_t *num_samples, uint32_t num_frames, int32_t *residual, int32_t *params, double *stats, uint8_t *rice_plan,
        uint32_t group_frames);
/* in place: data holds the residual on entry, PCM on return */
int LINNEAmd_MultiDecodeFramesHost(struct LINNEAmdMulti *multi, const struct LINNEAmdShape *shape, int32_t *data,
        const uint32_t *num_samples, uint32_t num_frames, const int32_t *params, uint32_t group_frames);

/* Number of (job, layer) unit-count searches of the last EncodeFramesDevice call that the certified order-free
 * search could not decide and that therefore ran the exact ordered sums (synchronises; -1 on error). */
int64_t LINNEAmd_GetLastFallbackCount(struct LINNEAmdContext *ctx);

/* Telemetry of the certified search (linne_network.c:338-341): over all (job, layer) searches of the last EncodeFramesDevice
 * call that the certificate decided, the smallest gap between the winning trial's upper bound and the runner-up's lower bound,
 * relative to the winning mean (synchronises; a huge value when no search had two trials; -1 on error).  LINNE_AMD_EXACT=1 in
 * the environment of ContextCreate makes every search take the exact ordered chains instead (for diffing the two paths). */
double LINNEAmd_GetLastMinMargin(struct LINNEAmdContext *ctx);

/* blocks until everything enqueued on the context's stream has finished */
int LINNEAmd_Synchronize(struct LINNEAmdContext *ctx);

/* Per-kernel timing of the last Encode/DecodeFramesDevice call, measured with HIP events recorded on the
 * context's stream around each launch (only while timing is enabled).  GetLastTimingMs returns the summed
 * milliseconds of all launches of one kernel kind (negative if none was recorded), GetLastTimingLaunches their
 * count.  which: 0 whole call, 1 prep, 2 window, 3 autocorrelation, 4 levinson, 5 trial residual, 6 loss sum,
 * 7 select, 8 forward, 9 final loss, 10 finalize (quantise + FIR cascade), 11 synthesis, 12 MS->LR,
 * 13 block-type statistics (runs on a side stream beside the analysis), 14 autocorrelation of the short layers
 * (kind 3 is the long layer's kernel), 15 / 16 trial residual / forward of layer 0 (int32 input; kinds 5 / 8 are the
 * double-input instantiations used by the other layers; when those layers run without the fused one-unit forward --
 * the last layer, or LINNE_AMD_SPECULATE=0 -- they are different kernels and report as kinds 18 / 19), 20 the last layer's
 * forward pass fused with its loss (k_fwd_loss; replaces 19 + 9 for the frames it takes), 21 / 22 / 23 the long layer's
 * autocorrelation with lanes = jobs (k_autocorr_hist for the trials of order P and P/2, k_autocorr_sub for the shorter ones;
 * replace 3 for the frames they take), 24 Rice scan + emission, 25 the long layer's search in one window pass (k_search_long;
 * replaces 5 for the frames it takes), 26 the real final pass of -a N, 27 the -l trainer, 28 Rice decoding, 30 / 31 the synthesis of
 * the long layer (k_synth_big) / of the short layers and the de-emphasis (k_synth_small); 11 is then the one-launch form
 * (k_synthesize), 32 the pipelined latency form (k_synth_pipe: small batches), 33 the throughput form of a layer (k_synth_rows / k_synth_rows8: four
 * or eight channel-frames per wave, what large batches take), 34 the de-emphasis behind it (k_deemph_lr; it includes MS->LR when whole frames
 * lie in a block of 64 rows: no kind 12 then). */
double LINNEAmd_GetLastTimingMs(struct LINNEAmdContext *ctx, int which);
int LINNEAmd_GetLastTimingLaunches(struct LINNEAmdContext *ctx, int which);
int LINNEAmd_EnableTiming(struct LINNEAmdContext *ctx, int enable);

/* Host entropy stage, batch form (thread pool over frames): serialises analysed frames to .lnn blocks exactly
 * as linne_encoder.c:698-749,806-855 does.  blocks_out receives the blocks back to back; block_sizes[f] their
 * byte counts.  pcm is needed for RAW blocks.  parcor_state (in/out, may be NULL = 0.0) carries oracle quirk Q2
 * across calls.  Returns LINNEApiResult. */
int LINNEAmd_PackFrames(const struct LINNEAmdShape *shape, const int32_t *pcm, const uint32_t *num_samples,
        uint32_t num_frames, const int32_t *residual, const int32_t *params, const double *stats,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state,
        uint32_t num_threads);
/* The same with the device's Rice plan (LINNEAmd_RicePlanDevice; NULL = search on the host): the host then only writes bits. */
int LINNEAmd_PackFramesPlanned(const struct LINNEAmdShape *shape, const int32_t *pcm, const uint32_t *num_samples,
        uint32_t num_frames, const int32_t *residual, const int32_t *params, const double *stats, const uint8_t *rice_plan,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state,
        uint32_t num_threads);

/* How the calling thread's last LINNEDecoder_DecodeWhole ran.  Bit 0: it finished with the device decoding the Rice codes
 * (LINNEAmd_SlotDecodeStreamSubmit; the default for CRC-checked streams, LINNE_AMD_DECODE_STREAM=0 turns it off).  Bit 1: it had
 * started that way, met a block no encoder writes (or 16-bit PCM out of range) and went over the stream again with the host's Rice
 * decoder, which is the reference's decoder restated (linne_coder.c:304-345). */
uint32_t LINNEAmd_LastDecodeWholeMode(void);

/* The host stage when the device wrote the Rice codes (LINNEAmd_RiceEmitDevice, or an encode slot with LINNE_AMD_SLOT_EMIT): block
 * types in stream order, then per block the header, the parameter bits (linne_encoder.c:698-735), the channels' codes appended at
 * the running bit position, padding and CRC16 (:743-749, :848-855).  PCM is read in place from the caller's planes (frame f of
 * the batch starts at sample first_sample + f * num_samples_per_block of every plane; needed for the SILENT test and RAW blocks).
 * fetch(arg, frame, dst[C][S]) must deliver a frame's residual; it is called for the (rare) channel-frames without a device code
 * (offset 0xFFFFFFFF).  Same bytes as LINNEAmd_PackFrames. */
int LINNEAmd_PackFramesEmitted(const struct LINNEAmdShape *shape, const int32_t *const *planes, uint64_t first_sample,
        const uint32_t *num_samples, uint32_t num_frames, const int32_t *params, const double *stats, const uint8_t *rice_plan,
        const uint8_t *packed, const uint32_t *offsets, int (*fetch)(void *arg, uint32_t frame, int32_t *dst), void *fetch_arg,
        uint8_t *blocks_out, uint64_t blocks_capacity, uint32_t *block_sizes, double *parcor_state, uint32_t num_threads);

#ifdef __cplusplus
}
#endif

#endif /* LINNE_AMD_H_INCLUDED */

/* dcs_hip.h -- C ABI of libdcs_hip.so: batched DCS audio frame decode on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE hot path of mjrgh/DCSExplorer: DCSDecoderNative's per-frame
 * decode (bitstream unpack -> dequantise -> inverse transform -> overlap-add -> int16 PCM).  The
 * reference has no process/device boundary on this path (everything is one C++ object), so the ABI
 * below is new; every entry point names the reference interface it replaces.  `file:line` citations
 * are relative to the reference tree (DCSDecoder/...).
 *
 *   reference                                              this library
 *   ----------------------------------------------------   ----------------------------------------
 *   DCSDecoderNative::GetStreamInfo   (Native.cpp:1486)     dcs_index_stream        (host scan)
 *   SetMasterVolume + MainLoop scale + UpdateMixingLevels   dcs_volume_multiplier, dcs_mixing_multiplier,
 *     (Native.cpp:3250, :227-269, :3072-3121)                dcs_frame_scale, dcs_stream_params
 *   DecoderImpl{93,93a,94x}::DecompressFrame  (:1679-3032)  dcs_decode_batch / dcs_batch_run (HIP kernel,
 *   DecoderImpl{93,94x}::TransformFrame       (:397-813)     phase 1 = unpack, phase 2 = transform+overlap)
 *   DCSDecoder::GetNextSample pump  (DCSDecoder.cpp:1579)   DCSDecoderHIP (include/DCSDecoderHIP.h) on top
 *
 * Plain C types only; all memory is caller-owned unless a function says otherwise; every function
 * returns a DcsStatus (0 = ok, negative = error) and never throws.  A DcsCtx is bound to one GPU and
 * is not thread-safe (same rule as the reference decoder object, DCSDecoder.h:90-105); use one
 * context per thread / per GPU.  There is NO CPU fallback: if no gfx950 device is usable,
 * dcs_ctx_create fails with DCS_ERR_NO_DEVICE and nothing decodes.
 */
#ifndef DCS_HIP_H
#define DCS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCS_ABI_VERSION 9              /* 9: dcs_decode_batch_live, dcs_seq_decode_view, dcs_seq_plan_ahead, dcs_seq_stream_playing_at (round 6) */
#define DCS_FRAME_SAMPLES 240          /* PCM samples per frame (DCSDecoder.h:123: 7.68 ms at 31250 Hz) */
#define DCS_MAX_CHANNELS 8             /* DCSDecoderNative.h:305 */

typedef int32_t DcsStatus;
enum
{
    DCS_OK = 0,
    DCS_ERR_INVALID_ARG   = -1,
    DCS_ERR_NO_DEVICE     = -2,        /* no usable gfx950 GPU / HIP runtime failure at init */
    DCS_ERR_HIP           = -3,        /* a HIP call failed; see dcs_last_error() */
    DCS_ERR_NO_MEMORY     = -4,
    DCS_ERR_CAPACITY      = -5,        /* caller-provided output array too small */
    DCS_ERR_BAD_STREAM    = -6         /* stream unusable (e.g. zero frames, truncated header) */
};

/* OS generation of the ROM the stream comes from (DCSDecoder.h:846-900).  OS94 and OS95 share one
 * audio format; OS93a differs from OS93b in its Type-1 streams and its mixing base. */
typedef enum DcsOsVersion { DCS_OS93A = 0, DCS_OS93B = 1, DCS_OS94 = 2, DCS_OS95 = 3 } DcsOsVersion;

/* Unpack layout of a stream = (OS family, header type bit, 94x sub-type == 0?)
 * (DCSDecoderNative.cpp:3147-3160, :1707-1712, :2308, :2841). */
typedef enum DcsFormat
{
    DCS_FMT_93_T0     = 0,             /* OS93a/OS93b Type 0  (DecoderImpl93, fixed-width bands)        */
    DCS_FMT_93B_T1    = 1,             /* OS93b Type 1        (Huffman band types, differential)        */
    DCS_FMT_93A_T1    = 2,             /* OS93a Type 1        (1-byte header, pair-table VQ)            */
    DCS_FMT_94_T0     = 3,             /* OS94/95 Type 0                                                 */
    DCS_FMT_94_T1_S0  = 4,             /* OS94/95 Type 1, sub-type 0 pre-adjust map                      */
    DCS_FMT_94_T1_S3  = 5              /* OS94/95 Type 1, sub-type 1..3 pre-adjust map                   */
} DcsFormat;

/* per-frame error bits (dcs_decode_batch errOut / DcsFrameIndex.flags >> 4) */
#define DCS_FRAME_STOP   1u            /* the reference's channel.stop: corrupt band, zeroed (:1989, :2216) */
#define DCS_FRAME_FATAL  2u            /* malformed beyond what the reference defines (it has UB there):
                                          decode of the frame stops at that point; STOP is raised too    */
#define DCS_FRAME_TAIL_LOST 4u         /* never set since ABI 9.  (Rounds 2-5: the overlap tail of the frame's predecessor, decoded
                                          by another wavefront of the launch, had not arrived within a wait bound.  Tails between
                                          wavefronts are a rendezvous now: nobody waits and nothing can be lost.) */

/* ------------------------------------------------------------------------------------------------
 * Index pass: the carried state that makes a frame independently decodable.
 * Replaces the serial walk of GetStreamInfo (DCSDecoderNative.cpp:1486-1537); there is no frame
 * index or sync word in a DCS stream, a frame's bit offset is only known after decoding every
 * earlier frame (:1715, :2260), and the 1994+/1993b-Type-1 formats delta-code band types (:1833, :2428).
 */
/* Decoder state at the start of header band k+1, k = 0..14: lets up to 16 lanes unpack one frame in
 * parallel.  The kernel gives a frame 4, 8 or 16 lanes (64 / frames-per-wavefront); with bpl = ceil(nBands /
 * lanes), lane q of a 1993-layout frame takes bands [q * bpl, (q + 1) * bpl) and starts from split[q * bpl - 1];
 * in a 1994+ frame, whose bands hold 7, 8, 13 x 16 and 32 samples, every lane after the first starts one band
 * later (bands 0 and 1 count as one, band 15 as two).  The library's packers do the dealing; a caller only
 * supplies the records. */
typedef struct DcsSplit
{
    uint16_t bitDelta;                 /* bits from the frame's first bit to the band's first bit         */
    uint16_t prv;                      /* 1993 formats: prvInput      (:2317)
                                          1994+, split[14] only: bits from the frame's first bit to the
                                          first code of band 15 that starts with half of the band's samples
                                          done (0: none recorded; then one lane unpacks the whole band)     */
    uint16_t prvDelta;                 /* 1993 formats: prvInputDelta (:2318)
                                          1994+, split[14] only: output index at that code (bits 0..8) |
                                          0x200 when a two-zeros code carried one sample across the middle */
    uint16_t state;                    /* output index (bits 0..8) | band sub-type << 9 (2 bits) |
                                          "reuse type 0" flag << 11 (:2319, :2388)                         */
} DcsSplit;

#define DCS_IDX_SERIAL 1u              /* flags: do not split this frame (a band raised an error): one lane
                                          unpacks all of it                                                */

typedef struct DcsFrameIndex
{
    uint32_t bitOff;                   /* first bit of the frame, counted from the first payload byte   */
    uint16_t nBits;                    /* bits the frame occupies                                       */
    uint16_t hdrBits;                  /* 1994+: bits of the frame header (band-type deltas, :1780-1834);
                                          band 0 starts at bitOff + hdrBits.  0 for the 1993 formats     */
    uint8_t  bandType[16];             /* 1994+: band-type codes AFTER this frame's header deltas;
                                          OS93b Type 1: codes carried INTO the frame (AudioStream::
                                          bandTypeBuf); values above 255 are stored as 255;
                                          OS93a Type 1 (no band-type codes, up to 18 bands): two DcsSplit
                                          records, of bands 16 and 17                                    */
    uint16_t preAdj;                   /* 1994+ Type 1: scale pre-adjust of bands 0..2 (4 bits each),
                                          derived from the PREVIOUS frame's codes (:1771-1773)           */
    uint8_t  nBands;                   /* populated header bands (stream constant)                       */
    uint8_t  flags;                    /* DCS_IDX_SERIAL | DCS_FRAME_* error bits << 4                   */
    DcsSplit split[15];
} DcsFrameIndex;                       /* 148 bytes */

typedef struct DcsStreamInfo           /* DCSDecoderNative::StreamInfo (DCSDecoderNative.h:106-122)     */
{
    int32_t  nFrames;                  /* frame count from the stream's U16 prefix                      */
    int32_t  nBytes;                   /* bytes from the start of the stream to the bit reader's byte
                                          pointer after the last frame, look-ahead included (:1509)      */
    int32_t  formatType;               /* header[0] & 0x80                                              */
    int32_t  formatSubType;            /* as GetStreamInfo computes it (:1517)                          */
    uint8_t  header[16];
    int32_t  format;                   /* DcsFormat                                                     */
    int32_t  hdrLen;                   /* 16, or 1 for OS93a Type 1 (:1457)                             */
    int32_t  nValidFrames;             /* frames indexed before a STOP/FATAL ended the stream (the frame
                                          that raised it is included: the reference still plays it)      */
    uint32_t payloadBits;              /* exact bit length of the nValidFrames frames                   */
} DcsStreamInfo;

/* Scan one stream on the host.  `stream` points at the U16 big-endian frame count.  Bytes past
 * `len` read as zero.  Writes min(nValidFrames, cap) entries to `out` (may be NULL with cap 0 to
 * only fill `info`).  Returns DCS_ERR_CAPACITY if cap was too small (info is still valid). */
DcsStatus dcs_index_stream(DcsOsVersion os, const uint8_t *stream, size_t len,
                           DcsFrameIndex *out, uint32_t cap, DcsStreamInfo *info);

/* Diagnostic: the same scan with a reader that keeps the reference reader's byte pointer literally (Peek pulls whole
 * bytes while nBits <= n, DCSDecoderNative.h:271) and decodes one code per look.  dcs_index_stream computes that pointer
 * instead and takes several codes per look; both return the same records and the same nBytes (tested). */
DcsStatus dcs_index_stream_literal(DcsOsVersion os, const uint8_t *stream, size_t len,
                                   DcsFrameIndex *out, uint32_t cap, DcsStreamInfo *info);

/* ------------------------------------------------------------------------------------------------
 * Per-frame mixing parameters (host arithmetic; inputs to the hot path)
 */
/* SetMasterVolume (DCSDecoderNative.cpp:3250-3282): 0..255 -> 1.15 PCM multiplier */
uint16_t dcs_volume_multiplier(int volume);
/* UpdateMixingLevels multiplier (:3072-3121): sum of mixing levels (level byte << 6 each), channel
 * volume 0..255 (ignored for OS93a, which starts from 0x7FFF) */
uint16_t dcs_mixing_multiplier(DcsOsVersion os, int levelSum, int channelVolume);
/* MainLoop's shared scale (:227-269): given the (unscaled) multipliers of the channels that are
 * active this frame, computes volShift (0..8) and rescales each multiplier in place. */
int dcs_frame_scale(uint16_t volMult, uint16_t *mixMul, const uint8_t *active, int nch);
/* The sequence MainLoop produces for ONE stream loaded on channel 0 of a freshly constructed
 * decoder with LoadAudioStream(0, ptr, level) (:1387): frame 0 uses the constructor's 0x7FFF
 * (DCSDecoderNative.h:514), later frames the value UpdateMixingLevels left behind. */
DcsStatus dcs_stream_params(DcsOsVersion os, int volume, int level, int channelVolume,
                            uint32_t nFrames, uint16_t *mixMulScaled, uint8_t *volShift);
/* The same for a decoder that has played before: frame 0 is mixed with `firstMixMul`, the multiplier the
 * previous tick left in the channel (MainLoop uses it at :240/:267 before UpdateMixingLevels recomputes it
 * at :281); 0x7FFF for a freshly constructed decoder (DCSDecoderNative.h:514). */
DcsStatus dcs_stream_params_from(DcsOsVersion os, int volume, int level, int channelVolume,
                                 uint16_t firstMixMul, uint32_t nFrames,
                                 uint16_t *mixMulScaled, uint8_t *volShift);

/* ------------------------------------------------------------------------------------------------
 * Batch description
 */
typedef struct DcsSrcDesc              /* one channel's contribution to one output frame; 160 bytes    */
{
    uint64_t streamOff;                /* byte offset in the blob of the stream's U16 frame count       */
    uint16_t mixMul;                   /* Channel::mixingMultiplier after MainLoop's rescale (:264-269) */
    uint8_t  format;                   /* DcsFormat                                                     */
    uint8_t  hdrLen;                   /* 16 or 1                                                       */
    DcsFrameIndex idx;                 /* the index pass's record of the frame                          */
} DcsSrcDesc;

#define DCS_PREV_NONE 0xFFFFFFFFu      /* overlap tail is all zero (fresh decoder / after silence)      */
#define DCS_PREV_EXT  0x80000000u      /* | index into the tailsIn array given to the run call          */

#define DCS_XFORM_93  0                /* DecoderImpl93::TransformFrame  (:614-813)                     */
#define DCS_XFORM_94  1                /* DecoderImpl94x::TransformFrame (:397-576)                     */

typedef struct DcsFrameJob             /* one output frame = one MainLoop pass; 16 bytes                */
{
    uint32_t firstSrc;                 /* first of nSrc consecutive DcsSrcDesc, mixed in that order     */
    uint8_t  nSrc;                     /* 0..8; 0 = silent spectrum (taper frame after a stream ends)   */
    uint8_t  volShift;                 /* 0..8 (:253-260)                                               */
    uint8_t  xform;                    /* DCS_XFORM_*: a decoder object has ONE transform (:3147-3160)  */
    uint8_t  flags;                    /* reserved, 0                                                   */
    uint32_t prev;                     /* job whose last 16 samples overlap into this frame (:569-575,
                                          :810-812), or DCS_PREV_NONE, or DCS_PREV_EXT | k              */
    uint32_t reserved;
} DcsFrameJob;

/* ------------------------------------------------------------------------------------------------
 * Context and execution
 */
typedef struct DcsCtx DcsCtx;

DcsStatus dcs_ctx_create(int deviceId, DcsCtx **ctx);
void      dcs_ctx_destroy(DcsCtx *ctx);
const char *dcs_last_error(const DcsCtx *ctx);     /* ctx may be NULL: last error of ctx_create */
int       dcs_device_count(void);                  /* does not initialise the GPU runtime further than counting */
/* Runtime settings the pipelines want, made explicitly: GPU_MAX_HW_QUEUES=8 if the variable is not set (the HIP runtime reads it
 * when it initialises; its default of four hardware queues serialises the chains of a pipeline's dozen streams).  Returns 1 if it
 * set the variable, 0 if the variable was already there.  LOADING the library changes nothing in the process's environment; the
 * library's own first calls into HIP (dcs_device_count, dcs_ctx_create) make this call once unless DCS_NO_RUNTIME_DEFAULTS is in
 * the environment.  It has an effect only before HIP is initialised and it calls setenv: a host with threads that read the
 * environment concurrently sets the variable itself at start-up and exports DCS_NO_RUNTIME_DEFAULTS. */
int       dcs_runtime_defaults(void);

/* tuning: frames handled per wavefront in the kernel (4, 8 or 16); 0 = choose from batch size */
DcsStatus dcs_ctx_set_frames_per_wave(DcsCtx *ctx, int fpw);
/* diagnostic: frames a wavefront decodes, when fewer than the kernel variant has slots for (the lanes of the unused
 * slots idle): 1 = one wavefront per frame.  0 (default) = every slot is used.  Same PCM at every setting. */
DcsStatus dcs_ctx_set_frames_per_chunk(DcsCtx *ctx, int frames);
/* tuning: how a frame gets the 16-sample tail of a predecessor that lies in another wavefront's chunk.  1 (default):
 * the two wavefronts meet in a device buffer -- each exchanges one word per tail sample there, and whichever of them arrives second
 * finishes the successor's first sixteen samples; nobody waits (round 6); 0: the predecessor is decoded a second time next to the
 * successor (a "halo" slot).
 * Same PCM either way.  Applies to batches created afterwards. */
DcsStatus dcs_ctx_set_tail_handoff(DcsCtx *ctx, int enable);
/* tuning: how dcs_decode_streams takes a LARGE list (32 768 frames and more, 32 streams and more), which it cuts into
 * eight parts that go through a pipeline of the context's own.  mode 1: index walk, planner and packer of every part on the
 * device (DCS_PIPE_ALL_ON_DEVICE; 3.3 ms for 256 x 256 frames, ~2 CPU-ms).  mode 2 (default, round 5): the same, with the host's
 * worker pool walking the list's first parts while the index kernel walks the last ones -- the kernel takes as long as its longest
 * stream however few streams it has, the pool delivers a part every third of a millisecond, and the early parts' PCM comes
 * down under the kernel; the records are the same bytes either way.  How many parts the pool takes follows the finish times
 * measured in the previous call.  Costs the pool's threads for the length of the call (~20 CPU-ms).  mode 0: the whole index
 * pass on the pool with the parts following it (4.0-4.4 ms, ~45 CPU-ms on 16 threads).  Same PCM in every mode. */
DcsStatus dcs_ctx_set_large_list_path(DcsCtx *ctx, int mode);
/* For a caller that runs SEVERAL resident batches on one GPU at once (batches on different streams, or several processes on one
 * card).  enable = 1: batches created afterwards keep their chunks in chain order and are launched in XCD ranges -- workgroup i runs on
 * XCD i % 8, XCD j decodes a contiguous range of chunks in order.  Rounds 4 and 5 NEEDED this when launches ran side by side (a
 * wavefront waited for the tail another wavefront published, and two launches could fill each other's places); since round 6 no
 * wavefront waits for another and the setting is a placement choice only (neighbouring chunks of a stream share an XCD's L2).
 * dcs_pipeline and dcs_node work this way.  Same PCM either way. */
DcsStatus dcs_ctx_set_concurrent_batches(DcsCtx *ctx, int enable);
/* Which frames' 16-sample tails a RESIDENT batch (dcs_batch_create) stores for dcs_batch_download's tailsOut.  0 (default): the last
 * frame of every chain of the batch -- a frame no other frame of the batch names as its predecessor -- which is what a caller needs
 * to carry a stream into its next batch (DCS_PREV_EXT); the other rows read as zero.  1: every frame (32 bytes of HBM writes per
 * frame more), for a caller that may resume behind ANY frame of the batch.  Applies to batches created afterwards.  dcs_decode_batch
 * stores every frame's tail whenever it is given a tailsOut array (DCSDecoderHIP's sequencer rewinds to any tick of its look-ahead). */
DcsStatus dcs_ctx_set_batch_tails(DcsCtx *ctx, int allFrames);
/* The context keeps device and pinned-host buffers of finished batches and lists for the next ones (hipMalloc / hipFree
 * cost as much as decoding thousands of frames, and hipFree waits for the whole device).  What it may keep is bounded:
 * by default min(32 GB, an eighth of the card's free memory at dcs_ctx_create) of device memory and min(8 GB, a
 * sixteenth of the host's RAM) of pinned memory (environment: DCS_CACHE_DEV_MB / DCS_CACHE_PIN_MB).  An allocation that
 * fails gives cached buffers back (largest first) and is tried again before DCS_ERR_NO_MEMORY reaches the caller.  A buffer
 * that comes back to a full cache stays and the OLDEST cached buffers go (the next list is most likely of the size of the last).
 * dcs_ctx_trim_cache releases everything the cache holds now (e.g. between jobs of very different list sizes);
 * dcs_ctx_cache_bytes reports what it holds and the limits.  Output pointers may be NULL. */
DcsStatus dcs_ctx_set_cache_limits(DcsCtx *ctx, uint64_t deviceBytes, uint64_t pinnedBytes);
DcsStatus dcs_ctx_trim_cache(DcsCtx *ctx, uint64_t *deviceBytesReleased, uint64_t *pinnedBytesReleased);
DcsStatus dcs_ctx_cache_bytes(DcsCtx *ctx, uint64_t *deviceBytes, uint64_t *pinnedBytes, uint64_t *deviceLimit, uint64_t *pinnedLimit);

/* One-shot convenience: host buffers in, host buffers out (H2D, kernel, D2H on the context's
 * stream, synchronous).  pcmOut = nJobs x 240 int16; errOut (optional) = nJobs x uint32 DCS_FRAME_*.
 * tailsIn (optional) = 16-sample overlap tails referenced by DCS_PREV_EXT; tailsOut (optional) =
 * nJobs x 16 samples, the tail each frame leaves for its successor (every frame's). */
DcsStatus dcs_decode_batch(DcsCtx *ctx,
                           const uint8_t *blob, size_t blobLen,
                           const DcsSrcDesc *srcs, uint32_t nSrcs,
                           const DcsFrameJob *jobs, uint32_t nJobs,
                           const int16_t *tailsIn, uint32_t nTailsIn,
                           int16_t *pcmOut, uint32_t *errOut, int16_t *tailsOut);

/* The same for a caller that comes back every few frames (DCSDecoderHIP's sample pump, DCSDecoder.cpp:1579-1690; the sequencer):
 * the context's LIVE decoder.  Nothing is allocated, created or cleared per call: the packages are built in a pinned arena the
 * context keeps, small batches are read and written by the kernel straight over the link (one launch, one wait), larger ones get
 * one copy up and one copy down (PCM, error words and EVERY frame's tail are one block).  The results are handed out as pointers
 * into the context's pinned arena (any of the three may be NULL), valid until the context's next decode call.
 * blobId names the blob: a caller whose blob only ever GROWS under one non-zero name (bytes already passed never change) has the
 * streams that multi-channel frames read on the device uploaded once, when they first appear; 0 = no such promise, the blob is
 * uploaded per call when some frame has more than one source (frames with one source never read it: their bytes travel in the
 * packages).  At most 131 072 frames a call (DCS_ERR_CAPACITY beyond).  dcs_decode_batch itself takes this way up to that size. */
DcsStatus dcs_decode_batch_live(DcsCtx *ctx,
                                const uint8_t *blob, size_t blobLen, uint64_t blobId,
                                const DcsSrcDesc *srcs, uint32_t nSrcs,
                                const DcsFrameJob *jobs, uint32_t nJobs,
                                const int16_t *tailsIn, uint32_t nTailsIn,
                                const int16_t **pcmOut, const uint32_t **errOut, const int16_t **tailsOut);

/* Resident batches: upload once, run many times, download when wanted.  This is the path bench.py
 * times (inputs already in HBM when the clock starts).
 * Lifetime and ordering: a batch belongs to its context and must be destroyed before it.  A run may be enqueued on
 * any stream of the context's device; the batch remembers the completion of its last run call (an event on that
 * stream), and dcs_batch_sync / _download / _download_view / _destroy wait for it, so a caller's own non-blocking
 * stream needs no extra synchronisation.  Every entry point makes the context's device current. */
typedef struct DcsBatch DcsBatch;

DcsStatus dcs_batch_create(DcsCtx *ctx,
                           const uint8_t *blob, size_t blobLen,
                           const DcsSrcDesc *srcs, uint32_t nSrcs,
                           const DcsFrameJob *jobs, uint32_t nJobs,
                           const int16_t *tailsIn, uint32_t nTailsIn,
                           DcsBatch **batch);
void      dcs_batch_destroy(DcsBatch *batch);
/* Enqueue the decode on `hipStream` (a hipStream_t passed as void*; NULL = the context's stream).
 * Asynchronous: returns after the launch.  No wavefront of the launch waits for another (a frame whose predecessor is decoded by
 * another wavefront meets it in the hand-off buffer, and the later of the two finishes the frame's first samples), so a launch
 * depends on nothing but itself, whatever else runs on the chip. */
DcsStatus dcs_batch_run(DcsBatch *batch, void *hipStream);
/* The same `count` times back to back (one call from the host language for many launches). */
DcsStatus dcs_batch_run_many(DcsBatch *batch, void *hipStream, int count);
/* Run `iters` times bracketed by HIP events on the same stream and return the average kernel
 * duration in milliseconds (what bench.py's roofline block divides by). */
DcsStatus dcs_batch_time(DcsBatch *batch, void *hipStream, int iters, float *avgMs);
/* The same with the `iters` launches dealt round-robin to `n` resident batches of one context: when their packages and PCM
 * together exceed the 256 MB Infinity Cache, no launch finds its inputs (or the lines of its outputs) there -- the kernel
 * time on COLD inputs (bench.py --rotate, roofline_cold). */
DcsStatus dcs_batch_time_rotating(DcsBatch *const *batches, uint32_t n, void *hipStream, int iters, float *avgMs);
DcsStatus dcs_batch_sync(DcsBatch *batch);
/* (tailsOut: nJobs x 16 samples; which rows are filled is the context's dcs_ctx_set_batch_tails setting at dcs_batch_create) */
DcsStatus dcs_batch_download(DcsBatch *batch, int16_t *pcmOut, uint32_t *errOut, int16_t *tailsOut);
/* bytes of one chunk package of this batch: the kernel's input is dcs_batch_num_chunks() packages of this size.  A batch planned
 * on the host sizes the packages' image of the LDS bit pool by its fullest chunk (a multiple of 128 bytes). */
uint32_t dcs_batch_package_bytes(const DcsBatch *batch);
/* the same without the copy into caller memory: PCM (and error words) in pinned host memory owned by the batch,
 * valid until the batch is run again or destroyed; the device-to-host copy then runs at link speed */
DcsStatus dcs_batch_download_view(DcsBatch *b, const int16_t **pcmOut, const uint32_t **errOut);
/* device pointers, for callers that keep the PCM on the GPU (int16 [nJobs][240]) */
void     *dcs_batch_device_pcm(DcsBatch *batch);
/* bytes the kernel reads + writes per launch of this batch, by the definition of SURVEY section 8(d): exact compressed
 * payload + the header (and U16 count) of every stream it draws on + a 56-byte frame descriptor per source + 480 bytes
 * of PCM per output frame.  dcs_batch_abi_bytes counts the descriptors as this ABI has them (160-byte DcsSrcDesc,
 * 16-byte DcsFrameJob) instead of 56 bytes. */
uint64_t  dcs_batch_algorithmic_bytes(const DcsBatch *batch);
uint64_t  dcs_batch_abi_bytes(const DcsBatch *batch);
uint32_t  dcs_batch_num_jobs(const DcsBatch *batch);
uint32_t  dcs_batch_num_chunks(const DcsBatch *batch);          /* wavefronts of one launch */
int       dcs_batch_frames_per_wave(const DcsBatch *batch);     /* the kernel variant chosen for it */
/* shader clock (MHz) the chip holds under an integer VALU load on every SIMD (a probe kernel of a few hundred
 * microseconds; not part of the decode path): turns a kernel duration into cycles */
DcsStatus dcs_ctx_clock_mhz(DcsCtx *ctx, float *mhzOut);
/* device memory -> pinned host memory in GB/s, measured now (five 64 MB copies on the context's stream): what the link allows
 * a sustained end-to-end rate (480 bytes of PCM per frame cross it).  Not part of the decode path. */
DcsStatus dcs_ctx_link_rate(DcsCtx *ctx, float *gbpsOut);
/* what a synchronous call on this box cannot get under, to hold the one-shot calls against: an empty kernel launched on the context's
 * stream and waited for (launchWaitUs), and the same with a copy of nFrames x 516 bytes (PCM, error word, tail) into pinned memory
 * behind it (launchCopyWaitUs); medians of `iters` rounds, microseconds of host time.  Not part of the decode path. */
DcsStatus dcs_ctx_call_floor(DcsCtx *ctx, uint32_t nFrames, int iters, float *launchWaitUs, float *launchCopyWaitUs);
/* test hooks: chunkOrderSeed != 0 -- batches planned on the host get their chunks (the units wavefronts decode) in a seeded random
 * order, so that frames are decoded long before or long after the frames whose tails they take; noXcdRanges != 0 -- no batch of the
 * context is launched in XCD ranges.  Same PCM either way: tails between chunks are a rendezvous, nobody waits (csrc/dcs_kernels.hip.h) */
DcsStatus dcs_ctx_set_test_hooks(DcsCtx *ctx, uint32_t chunkOrderSeed, int noXcdRanges);

/* ------------------------------------------------------------------------------------------------
 * Whole-stream convenience (the reference's --extract-streams shape, DCSExplorer.cpp:1628-1907):
 * index + parameters + decode of nStreams independent streams, each played alone from a fresh
 * decoder at (volume, level); extraFrames taper frames are appended per stream (ExtractToWAV plays
 * nFrames+2, :1670-1721).  pcmOut receives the streams back to back; frameOffsets (nStreams+1,
 * optional) the first output frame of each stream.
 */
typedef struct DcsStreamRef
{
    const uint8_t *data;               /* U16 frame count, header, payload                              */
    size_t         len;
    int32_t        os;                 /* DcsOsVersion                                                  */
    int32_t        volume;             /* master volume 0..255                                          */
    int32_t        level;              /* mixing level byte (track opcodes 07-0C), e.g. 0x64            */
    int32_t        channelVolume;      /* 0..255, normally 255                                          */
} DcsStreamRef;

DcsStatus dcs_decode_streams(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams,
                             uint32_t extraFrames, int16_t *pcmOut, size_t pcmCapFrames,
                             uint32_t *frameOffsets, uint32_t *errOut);
/* number of output frames dcs_decode_streams will produce (host-only; runs the index pass) */
DcsStatus dcs_count_stream_frames(const DcsStreamRef *streams, uint32_t nStreams,
                                  uint32_t extraFrames, uint64_t *nFramesOut);

/* Several GPUs of one node (the reference decodes on one thread, DCSExplorer.cpp:1742-1907).
 * Streams are the independent units of the path, so the work is cut by plain range partition and no device needs
 * anything from another (no collective).
 * dcs_partition_streams: cut a list of nStreams streams with the given frame counts into nParts contiguous ranges
 * balanced by total frame count; range r is [firstStreamOut[r], firstStreamOut[r+1]) (nParts + 1 entries; a range
 * may be empty).  Every range's frame total lies within one (longest) stream of total / nParts.
 * dcs_decode_streams_sharded: dcs_decode_streams over the devices `deviceIds` -- range d decoded on device d into its
 * own part of pcmOut / errOut, one host thread per device (bound to the CPUs of that GPU's NUMA node).  The contexts
 * live in a node-level object (below) that the library keeps per device list from the first call on, so a call creates
 * no context, stream or pipeline; dcs_node_cache_release() destroys them (call it before unloading the library, or
 * never).  Same output layout and the same PCM as dcs_decode_streams on one device.  firstStreamOfDevice (optional,
 * nDevices + 1) receives the cut. */
DcsStatus dcs_partition_streams(const uint32_t *frameCounts, uint32_t nStreams, uint32_t nParts,
                                uint32_t *firstStreamOut);
DcsStatus dcs_decode_streams_sharded(const int *deviceIds, uint32_t nDevices,
                                     const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames,
                                     int16_t *pcmOut, size_t pcmCapFrames, uint32_t *frameOffsets,
                                     uint32_t *errOut, uint32_t *firstStreamOfDevice);

void      dcs_node_cache_release(void);

/* Several GPUs behind one object (the loop being spread out is DCSExplorer.cpp:1628-1907).  N persistent
 * contexts -- deviceIds may name a device more than once -- with one dcs_pipeline each (`depth` lists in flight per
 * device, DCS_PIPE_* flags; created with the first list).  dcs_node_submit deals a list to the device with the fewest
 * FRAMES in flight among those with room and blocks while none has room; dcs_node_collect returns the results in
 * SUBMISSION order whatever device decoded them (pointers into that device's pinned memory, valid until the next
 * collect; deviceIndexOut, optional, says which entry of deviceIds it was; DCS_ERR_INVALID_ARG when no submitted list is
 * outstanding).  Streams must stay valid until collected; submit and collect may run on different threads (a list counts as
 * submitted once dcs_node_submit has returned; concurrent submits are taken one at a time).  No data moves between the devices: no collective, no peer copies.
 * Placement: every context, its pipeline's worker and indexer threads and the pinned buffers they allocate are created
 * from a thread bound to the CPUs of the GPU's NUMA node (dcs_device_numa_node: /sys/bus/pci/devices/<addr>/numa_node; the
 * default local memory policy then places the buffers there).  Where the node is unknown (-1) nothing is bound. */
typedef struct DcsNode DcsNode;
struct DcsPipelineResult;
int         dcs_device_numa_node(int deviceId);
DcsStatus   dcs_node_create(const int *deviceIds, uint32_t nDevices, int depth, uint32_t flags, DcsNode **out);
void        dcs_node_destroy(DcsNode *node);
DcsStatus   dcs_node_submit(DcsNode *node, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames);
DcsStatus   dcs_node_collect(DcsNode *node, struct DcsPipelineResult *out, int *deviceIndexOut);
uint32_t    dcs_node_num_devices(const DcsNode *node);
DcsStatus   dcs_node_device_info(const DcsNode *node, uint32_t index, int *deviceId, int *numaNode, uint64_t *listsDone);
const char *dcs_node_last_error(const DcsNode *node);

/* Batches in flight.  A caller with many lists of streams to decode (an archive, a ROM corpus) submits them and
 * collects their PCM in submission order; `depth` host worker threads each take a list through index pass, mixing
 * parameters, chunk plan, packing, upload, kernel and download on a HIP stream of their own, so the host preparation
 * of one list runs while the GPU decodes another and a third comes back over PCIe into pinned memory.  Same PCM as
 * dcs_decode_streams list by list.  The streams (and the bytes they point at) must stay valid until their list has
 * been collected.  submit blocks while `depth` lists are between submit and collect; collect blocks until the OLDEST
 * submitted list is finished.  The result's pointers are pinned memory of the pipeline, valid until the next
 * collect / destroy.  The pipeline owns the context's decode work while it exists (do not decode on the same context
 * from another thread meanwhile), and must be destroyed before the context. */
typedef struct DcsPipeline DcsPipeline;
typedef struct DcsPipelineResult
{
    const int16_t  *pcm;               /* nFrames x 240                                                    */
    const uint32_t *err;               /* nFrames x DCS_FRAME_*                                            */
    const uint32_t *frameOffsets;      /* nStreams + 1: first output frame of each stream                  */
    uint32_t        nFrames, nStreams;
    DcsStatus       status;
    float           hostMs, deviceMs;  /* the worker's wall time in host preparation / upload + kernel + download */
    uint32_t        path;              /* DCS_PIPE_*: the stages of THIS list that ran on the device (a list the device
                                          stages cannot serve -- see the flags -- takes the host's, same PCM)          */
} DcsPipelineResult;
/* flags: DCS_PIPE_INDEX_ON_DEVICE -- the index pass of every list runs on the GPU (one wavefront per stream, the walk of
 * dcs_index_streams_gpu) instead of on the host pool.  One list takes longer that way, many lists in flight much less:
 * the walks of different lists overlap on the GPU, and the host cores, which the index pass otherwise keeps busy most of
 * the time, are left with parameters, planner and packer.  Worth it from about 8 lists in flight.  Same PCM either way.
 * DCS_PIPE_PACK_ON_DEVICE (implies the former) -- the chunk packages are assembled on the device as well, from the
 * index records and streams that are already resident there; the host plans from an 8-byte-per-frame digest and
 * neither receives the records nor builds or uploads packages. */
#define DCS_PIPE_INDEX_ON_DEVICE 1u
#define DCS_PIPE_PACK_ON_DEVICE  2u
/* DCS_PIPE_PLAN_ON_DEVICE (implies the two above) -- the chunk plan is made on the device as well (dcsPlanKernel): a list of
 * whole streams has a regular job list, so its plan is arithmetic, one thread per chunk.  Nothing of the index results
 * comes back to the host; a worker lays the list's streams out and uploads them, and once the walk is done queues planner,
 * packer, decode kernel and the copy down on one stream and sleeps until the PCM is there.  A list the arithmetic plan
 * cannot serve (a stream that runs past its buffer; a chunk whose compressed bytes overflow the kernel's bit pool, where the
 * host planner closes the chunk early -- such a list is first planned again on the device with three quarters, then half of
 * the frames per chunk) is decoded by the host-planned path instead, same PCM (DcsPipelineResult.path tells). */
#define DCS_PIPE_PLAN_ON_DEVICE  4u
#define DCS_PIPE_ALL_ON_DEVICE   7u      /* the three together: what a caller with many lists in flight wants */
DcsStatus dcs_pipeline_create(DcsCtx *ctx, int depth /* 1..64 lists in flight */, uint32_t flags, DcsPipeline **out);
void      dcs_pipeline_destroy(DcsPipeline *p);
DcsStatus dcs_pipeline_submit(DcsPipeline *p, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames);
DcsStatus dcs_pipeline_collect(DcsPipeline *p, DcsPipelineResult *out);

/* The whole device path of one list of whole streams, RESIDENT: stream bytes in HBM -> PCM in HBM, nothing over PCIe.
 * What the reference does per stream -- the GetStreamInfo walk that finds every frame (DCSDecoderNative.cpp:1486-1537), then
 * DecompressFrame + TransformFrame per frame (:1546-1589, :272-278) -- as four kernels on one HIP stream: index walk (one
 * wavefront per stream), chunk planner, packer, decode.  create uploads the streams once and runs one pass (same PCM as
 * dcs_decode_streams; DCS_ERR_BAD_STREAM if the device planner cannot serve the list, see DCS_PIPE_PLAN_ON_DEVICE);
 * run queues `iters` passes back to back and reports the average time of a pass, and of each kernel from HIP events around
 * it (five further passes); download copies the last pass's PCM (nFrames x 240), error words and the first output frame of
 * every stream (nStreams + 1) to the host.  A measurement and test entry (bench.py device_full_path); lists in flight with
 * host buffers on both ends are dcs_pipeline's job. */
typedef struct DcsDevicePath DcsDevicePath;
typedef struct DcsDevicePathTimes
{
    float    passMs;                   /* index + plan + pack + decode, average over the back-to-back passes            */
    float    indexMs, planMs, packMs, decodeMs;     /* per kernel (plan includes clearing error and hand-off words) */
    uint32_t planFlags;                /* DCS_PLAN_*: 0 = the device planner served the list                            */
    uint32_t nStreams, nFrames, framesPerWave;
    uint64_t algorithmicBytes;         /* SURVEY 8(d) bytes of one pass (as dcs_batch_algorithmic_bytes)                */
} DcsDevicePathTimes;
DcsStatus dcs_device_path_create(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, DcsDevicePath **out);
DcsStatus dcs_device_path_run(DcsDevicePath *path, int iters, DcsDevicePathTimes *times);
DcsStatus dcs_device_path_run_many(DcsDevicePath *path, int iters);    /* `iters` passes back to back, a wait, nothing timed */
DcsStatus dcs_device_path_download(DcsDevicePath *path, int16_t *pcmOut, uint32_t *errOut, uint32_t *frameOffsets);
void      dcs_device_path_destroy(DcsDevicePath *path);

/* The stream loop of `DCSExplorer --extract-streams` exactly (DCSExplorer.cpp:1628-1907): ONE decoder
 * object plays the streams one after the other -- LoadAudioStream(0, ptr, level), nFrames + extraFrames
 * frames, ClearTracks() during the last two (ExtractToWAV :1670-1721) -- so frame 0 of every stream but the
 * first is mixed with the multiplier the previous stream's level left behind (level 0 if that stream was
 * stopped by an error, :95-116), and only the very first stream sees the constructor's 0x7FFF.  All streams
 * must name the same os / volume / channelVolume (they are properties of the one decoder); extraFrames
 * must be at least 2, as it is there, so that every stream starts from a silent decoder.  Output layout as
 * dcs_decode_streams.  Still one kernel launch for everything. */
DcsStatus dcs_decode_stream_sequence(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams,
                                     uint32_t extraFrames, int16_t *pcmOut, size_t pcmCapFrames,
                                     uint32_t *frameOffsets, uint32_t *errOut);

/* ------------------------------------------------------------------------------------------------
 * Index many streams at once: on `nThreads` host threads (0 = all hardware threads), or on the GPU with
 * one wavefront per stream (dcs_index_streams_gpu; the streams are given as offsets into one blob, which is
 * uploaded, walked by the index kernel and the records downloaded).  Stream k's records go to
 * out + firstRecord(k), at most nFrames(k) of them (nFrames = the stream's U16 prefix); infos[k] receives
 * its summary.  Both run the same walker (csrc/dcs_scan.h) and return identical records; they replace
 * the serial GetStreamInfo walk (DCSDecoderNative.cpp:1486-1537) in front of the decode kernel.
 */
/* host threads this process can run at once: CPUs of its affinity mask, limited by a cgroup CPU quota if any
 * (what "0 = all" means for nThreads below and for the packer's worker count) */
int       dcs_host_threads(void);
DcsStatus dcs_index_streams(const DcsStreamRef *streams, uint32_t nStreams, int nThreads,
                            DcsFrameIndex *out, const uint64_t *firstRecord, DcsStreamInfo *infos);

typedef struct DcsStreamLoc
{
    uint64_t off;                      /* byte offset of the stream (its U16 frame count) in the blob     */
    uint32_t len;                      /* bytes that belong to the stream (bytes past it read as zero)    */
    int32_t  os;                       /* DcsOsVersion                                                  */
    uint64_t firstRecord;              /* where the stream's records go in `out`                        */
} DcsStreamLoc;

DcsStatus dcs_index_streams_gpu(DcsCtx *ctx, const uint8_t *blob, size_t blobLen,
                                const DcsStreamLoc *streams, uint32_t nStreams,
                                DcsFrameIndex *out, uint64_t outCap, DcsStreamInfo *infos);
/* average milliseconds of the index kernel alone over `iters` launches of the last
 * dcs_index_streams_gpu call's inputs (kept resident in the context until the next call) */
DcsStatus dcs_index_streams_gpu_time(DcsCtx *ctx, int iters, float *avgMs);


/* ------------------------------------------------------------------------------------------------
 * ROM ingestion: from sound ROM images (U2..U9, or a PinMame .zip of them) to the streams the decoder is
 * fed with.  Same results as the reference's DCSDecoder::AddROM / CheckROMs / MakeROMPointer /
 * GetTrackInfo / DecompileTrackProgram / ListStreams (DCSDecoder.cpp:26-76, :207-495, :672-1293) and
 * LoadROMFromZipFile (DCSDecoderZipLoader.cpp:60-207) on the same images; every read is bounds-checked
 * (a pointer that leaves its image reads 0xFF, like unpopulated ROM space on the board).
 */
typedef struct DcsRomSet DcsRomSet;

enum { DCS_HW_UNKNOWN = 0, DCS_HW_INVALID = 1, DCS_HW_DCS93 = 2, DCS_HW_DCS95 = 3 };    /* DCSDecoder.h:816-822 */

typedef struct DcsRomCheck
{
    int32_t  status;                   /* CheckROMs: 1 = all good, else the Ux number of the first bad ROM   */
    int32_t  hw;                       /* DCS_HW_*                                                          */
    int32_t  os;                       /* DcsOsVersion, -1 = not detected                                   */
    uint32_t nominalVersion;           /* 0x0103.. when the software carries one, else 0                    */
    uint32_t catalogOffset;            /* 0x3000 / 0x4000 / 0x6000, 0 = none                                */
    uint32_t nTracks;
    char     signature[128];           /* U2 signature text, empty if U2 does not look like one             */
} DcsRomCheck;

typedef struct DcsTrackInfo            /* DCSDecoder::TrackInfo (DCSDecoder.h:375-430)                      */
{
    uint32_t address;                  /* 24-bit linear ROM address of the track                            */
    int32_t  channel;
    int32_t  type;                     /* 1 byte-code program, 2 deferred, 3 deferred indirect              */
    int32_t  deferCode;
    uint32_t time;                     /* running time in frames                                            */
    int32_t  looping;
} DcsTrackInfo;

typedef struct DcsTrackOp              /* DCSDecoder::Opcode without the text (DCSDecoder.h:432-478)        */
{
    int32_t  offset;                   /* byte offset of the step in the program                            */
    int32_t  nestingLevel;
    int32_t  loopParent;               /* -1 at top level                                                   */
    uint16_t delayCount;
    uint8_t  opcode;
    uint8_t  nOperandBytes;
    uint8_t  operandBytes[8];
} DcsTrackOp;

typedef struct DcsExtractItem          /* one stream of the --extract-streams loop                          */
{
    uint32_t track;                    /* track whose program plays it first                                */
    uint32_t streamNum;                /* 1-based count within that track (the file name's _%02X_)          */
    uint32_t address;                  /* linear ROM address of the stream                                  */
    int32_t  level;                    /* mixing level the loop plays it at                                 */
} DcsExtractItem;

DcsRomSet  *dcs_romset_create(void);
void        dcs_romset_destroy(DcsRomSet *rs);
const char *dcs_romset_last_error(const DcsRomSet *rs);
DcsStatus   dcs_romset_add_rom(DcsRomSet *rs, int chip /* 2..9 */, const uint8_t *data, size_t size);   /* copies */
DcsStatus   dcs_romset_load_zip(DcsRomSet *rs, const char *path, const char *explicitU2 /* may be NULL */);
DcsStatus   dcs_romset_load_zip_memory(DcsRomSet *rs, const uint8_t *zip, size_t len, const char *zipBaseName,
                                       const char *explicitU2);
DcsStatus   dcs_romset_check(DcsRomSet *rs, DcsRomCheck *out);           /* also adopts the detected versions    */
DcsStatus   dcs_romset_set_version(DcsRomSet *rs, int hw, int os);       /* explicit override (no ADSP code in U2) */
uint32_t    dcs_romset_num_tracks(const DcsRomSet *rs);
DcsStatus   dcs_romset_pointer(const DcsRomSet *rs, uint32_t linear, const uint8_t **p, size_t *avail, int *chip);
/* bytes from p to the end of the ROM image of this set that contains p; 0 when p points into none of them */
size_t      dcs_romset_bytes_behind(const DcsRomSet *rs, const uint8_t *p);
DcsStatus   dcs_romset_track_info(const DcsRomSet *rs, uint32_t track, DcsTrackInfo *ti);   /* BAD_STREAM: no such track */
DcsStatus   dcs_romset_decompile(const DcsRomSet *rs, uint32_t track, DcsTrackOp *ops, uint32_t cap, uint32_t *nOut);
DcsStatus   dcs_romset_list_streams(const DcsRomSet *rs, uint32_t *addrs, uint32_t cap, uint32_t *nOut);
/* which streams `DCSExplorer --extract-streams` extracts, in its order and at its mixing levels
 * (DCSExplorer.cpp:1742-1810), and the same as input for dcs_decode_stream_sequence */
DcsStatus   dcs_romset_extract_plan(const DcsRomSet *rs, DcsExtractItem *items, uint32_t cap, uint32_t *nOut);
DcsStatus   dcs_romset_stream_refs(const DcsRomSet *rs, const DcsExtractItem *items, uint32_t n, int volume,
                                   DcsStreamRef *refs);
/* which tracks `DCSExplorer --extract-tracks` extracts (DCSExplorer.cpp:1735-1925): every type-1 track whose program
 * holds at least one Play opcode, and the frames ExtractToWAV writes for it -- the track's running time as a
 * uint16_t plus two, in uint16_t arithmetic as there (:1667-1672) */
typedef struct DcsExtractTrack
{
    uint32_t track;
    uint32_t nFrames;                  /* frames of the track's WAV file                                    */
} DcsExtractTrack;
DcsStatus   dcs_romset_extract_tracks_plan(const DcsRomSet *rs, DcsExtractTrack *items, uint32_t cap, uint32_t *nOut);

/* ------------------------------------------------------------------------------------------------
 * Track-program sequencer: everything DCSDecoderNative::MainLoop does per 7.68 ms tick except decompress
 * and transform (command queue, ExecTrack opcodes 00-12, loops, deferred tracks, mixing levels and fades,
 * looping streams, host event timers, the data-port protocol; DCSDecoderNative.cpp:89-306, :826-1371,
 * :1387-1463, :1546-1589, :3042-3135, :3297-3437), run on the host for any number of ticks ahead.  Each
 * tick appends one frame job to a pending plan; dcs_seq_decode turns the whole plan into PCM with ONE
 * kernel launch.  This is what `--extract-tracks` / `--validate --autoplay` style callers need: up to 8
 * channels mixed in the frequency domain before one transform, with the per-frame multipliers and shared
 * scale the sequencer derives.  The ROM set must outlive the sequencer and have its versions set
 * (dcs_romset_check or dcs_romset_set_version).
 */
typedef struct DcsSequencer DcsSequencer;
typedef struct DcsHostByte { uint32_t tick; uint32_t byte; } DcsHostByte;      /* a byte the decoder sent to the host */

DcsSequencer *dcs_seq_create(const DcsRomSet *rs);                   /* NULL: no U2 / versions unknown          */
DcsSequencer *dcs_seq_create_standalone(DcsOsVersion os);            /* no ROMs: streams from caller memory     */
void        dcs_seq_destroy(DcsSequencer *seq);
const char *dcs_seq_last_error(const DcsSequencer *seq);
DcsStatus   dcs_seq_set_master_volume(DcsSequencer *seq, int volume);             /* SetMasterVolume           */
DcsStatus   dcs_seq_set_reported_version(DcsSequencer *seq, uint16_t version);    /* answer to 55C2/55C3       */
DcsStatus   dcs_seq_write_data_port(DcsSequencer *seq, uint8_t byte);             /* WriteDataPort             */
DcsStatus   dcs_seq_add_track_command(DcsSequencer *seq, uint16_t track);         /* AddTrackCommand (:1475)   */
DcsStatus   dcs_seq_clear_tracks(DcsSequencer *seq);                              /* ClearTracks (:1466)       */
DcsStatus   dcs_seq_load_audio_stream(DcsSequencer *seq, int channel, uint32_t linearAddress, int mixingLevel);
/* LoadAudioStream for a stream in caller memory (copied), the ROM-less recipe of DCSEncoder.cpp:522-571 */
DcsStatus   dcs_seq_load_audio_stream_mem(DcsSequencer *seq, int channel, const uint8_t *data, size_t len, int mixingLevel);
DcsStatus   dcs_seq_plan(DcsSequencer *seq, uint32_t nTicks);        /* run nTicks ticks, extend the pending plan  */
/* the same for a caller that decodes ahead of what it has been asked for: at most maxTicks, at least one, and no further than
 * idleTicks ticks into silence (nothing playing, no program, no timer, nothing queued: digital silence until the next command);
 * *plannedOut = ticks run */
DcsStatus   dcs_seq_plan_ahead(DcsSequencer *seq, uint32_t maxTicks, uint32_t idleTicks, uint32_t *plannedOut);
/* go back inside the current batch (pending, or just decoded): state, overlap tail and host bytes as they were
 * after its first keepTicks ticks; how a caller that decodes ahead stays exact when a command arrives */
DcsStatus   dcs_seq_rewind(DcsSequencer *seq, uint32_t keepTicks);
/* make dcs_seq_rewind possible (off by default): a snapshot of the machine every 64 ticks; going back runs the ticks between the
 * snapshot and the tick asked for again (the machine is deterministic; bytes for the host are not sent twice) */
DcsStatus   dcs_seq_set_rewindable(DcsSequencer *seq, int on);
uint32_t    dcs_seq_pending_ticks(const DcsSequencer *seq);
int         dcs_seq_is_fatal(const DcsSequencer *seq);               /* DecoderFatalError after 4 failed passes    */
uint64_t    dcs_seq_tick(const DcsSequencer *seq);                   /* ticks run so far                           */
uint64_t    dcs_seq_fatal_tick(const DcsSequencer *seq);             /* first silent tick of a fatal error, or ~0  */
int         dcs_seq_stream_playing(const DcsSequencer *seq, int channel);   /* IsStreamPlaying (:101)              */
/* ... as it was after the first `ticks` ticks of the current batch, without going back there (a caller that has handed out
 * `ticks` frames of its look-ahead answers IsStreamPlaying from this and keeps the rest) */
int         dcs_seq_stream_playing_at(const DcsSequencer *seq, uint32_t ticks, int channel);
/* whether any channel had a track program or a stream after the first `ticks` ticks of the current batch, i.e. whether ClearTracks
 * (DCSDecoderNative.cpp:1466-1473) would change anything there; 0 = no (a caller that decodes ahead then need not go back) */
int         dcs_seq_tracks_active_at(const DcsSequencer *seq, uint32_t ticks);
/* bytes sent to the host since the last successful call; returns their number (call with out = NULL to size) */
uint32_t    dcs_seq_host_bytes(DcsSequencer *seq, DcsHostByte *out, uint32_t cap);
/* decode the pending plan (pcmOut = pending ticks x 240 samples) in one launch and clear it; the overlap
 * tail carries into the next plan */
DcsStatus   dcs_seq_decode(DcsCtx *ctx, DcsSequencer *seq, int16_t *pcmOut, size_t pcmCapFrames, uint32_t *errOut);
/* the same without the copy: *pcmOut (and *errOut, optional) point into the context's pinned memory (dcs_decode_batch_live),
 * *nFramesOut frames, valid until the context decodes again.  The streams the sequencer has loaded stay resident on the device:
 * each is uploaded once.  At most 131 072 pending ticks. */
DcsStatus   dcs_seq_decode_view(DcsCtx *ctx, DcsSequencer *seq, const int16_t **pcmOut, uint32_t *nFramesOut, const uint32_t **errOut);

/* The track loop of `DCSExplorer --extract-tracks` exactly (DCSExplorer.cpp:1628-1721, :1905-1925): ONE decoder --
 * SoftBoot, SetMasterVolume(255) -- plays the tracks one after the other: ClearTracks(), AddTrackCommand(track), nFrames
 * frames, ClearTracks() after each of the last two.  Whatever a track leaves behind (channel state, mixing levels, a
 * deferred track, the overlap tail) is what the next one starts from, as there.  The sequencer runs all of it ahead on
 * the host and the frames of every track are decoded in ONE kernel launch.  pcmOut receives the tracks back to back,
 * frameOffsets (n + 1, optional) the first frame of each.  (What the decoder sends to the host meanwhile is not returned
 * here: a caller that wants it drives a DcsSequencer itself and reads dcs_seq_host_bytes.) */
DcsStatus   dcs_extract_tracks(DcsCtx *ctx, const DcsRomSet *rs, const DcsExtractTrack *items, uint32_t n,
                               int16_t *pcmOut, size_t pcmCapFrames, uint32_t *frameOffsets, uint32_t *errOut);

/* ------------------------------------------------------------------------------------------------
 * Output formats of the reference's extraction and validation modes.
 */
/* the 44-byte WAV header of ExtractToWAV (DCSExplorer.cpp:1686-1699): mono, 16 bit, 31 250 Hz, nFrames x 240 samples */
void      dcs_wav_header(uint32_t nFrames, uint8_t out[44]);
/* the 36-byte "DCSa" raw-stream container header (DCSExplorer.cpp:1831-1866; README.md:274-289); the data section
 * that follows is the stream's first GetStreamInfo().nBytes bytes */
DcsStatus dcs_dcsa_header(DcsOsVersion os, uint32_t nBytes, uint8_t out[36]);
/* its reader, with the acceptance test of DCSEncoder::IsDCSFile (DCSEncoder.cpp:358-400) */
DcsStatus dcs_dcsa_parse(const uint8_t *file, size_t len, DcsOsVersion *osOut,
                         const uint8_t **streamOut, uint32_t *nBytesOut);
DcsStatus dcs_write_wav(const char *path, const int16_t *pcm, uint32_t nFrames);
DcsStatus dcs_write_dcsa(const char *path, DcsOsVersion os, const uint8_t *stream, uint32_t nBytes);
/* one frame of the --validate log (DCSExplorer.cpp:1358-1447): returns the number of differing samples and
 * formats the reference's log block into `text` (optional) */
int       dcs_frame_diff(uint64_t frameNo, const int16_t *mine, const int16_t *theirs,
                         char *text, size_t textCap, size_t *textLen);

/* ------------------------------------------------------------------------------------------------
 * Synthetic stream writer (seeded, integer-only).  Produces VALID streams
 * of every unpack layout for tests and benchmarks -- the reference ships no audio (Tests/.gitignore).
 */
typedef struct DcsSynthParams
{
    uint64_t seed;
    int32_t  format;                   /* DcsFormat                                                     */
    int32_t  nFrames;                  /* 1..65535                                                      */
    int32_t  nBands;                   /* populated header bands: 1..16 (93a Type 1: 1..18)             */
    int32_t  strideFromBand;           /* first band carrying the half-density 0x40 bit; >= 16 = none   */
    int32_t  profile;                  /* 0 = default mix, 1 = dense (wide codes), 2 = sparse (many zero
                                          bands), 3 = adversarial edge cases (max widths, deep codes),
                                          4 = saturated (every band at its widest code in every frame:
                                          the largest frames the formats can express, ~500 bytes),
                                          5 = SURVEY config 3 (1994+ layouts: band-type deltas 0 / +-1 / +-2 /
                                          other at 70 / 20 / 8 / 2 %, a quarter of the Huffman-coded values
                                          zero, ~120 bytes a frame with 12 bands; other layouts as 0),
                                          6 = SURVEY config 2 (1993 band layouts: scale codes uniform in
                                          0x20..0x34, band-type codes 0 / 1-3 / 4-6 / 7-9 at 15 / 35 / 40 /
                                          10 %, sub-type changes with probability 0.2, samples uniform in
                                          the signed range of their width; other layouts as 0)              */
    int32_t  reserved;
} DcsSynthParams;

/* Returns the stream length in bytes via *lenOut; writes at most cap bytes (DCS_ERR_CAPACITY if the
 * stream did not fit; call with out = NULL, cap = 0 to size). */
DcsStatus dcs_synth_stream(const DcsSynthParams *params, uint8_t *out, size_t cap, size_t *lenOut);

/* Diagnostic: the chunk plan the kernel launch would use for `jobs` at `fpw` frames per wavefront.
 * Each slot is returned as job | prevSlot<<32 | flags<<40 (flags: 1 = halo, 2 = external tail, 4 = publishes its
 * tail for a later chunk, 8 = takes its predecessor's tail from an earlier chunk, 0x80 = padding).  One wavefront
 * decodes one chunk of fpw slots.  A frame whose overlap predecessor lies in another chunk takes the tail from the
 * hand-off buffer when the predecessor is the last frame of an earlier chunk (and `handoff` is non-zero); otherwise
 * the predecessor is decoded again as a halo slot.  `srcs` (may be NULL) lets the planner also respect the kernel's
 * LDS budget for staged compressed bytes.  dcs_plan_chunks = dcs_plan_chunks2 with handoff = 1. */
DcsStatus dcs_plan_chunks(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw,
                          uint64_t *slotsOut, size_t cap, uint32_t *nChunksOut);
DcsStatus dcs_plan_chunks2(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw, int handoff,
                           uint64_t *slotsOut, size_t cap, uint32_t *nChunksOut);

/* Diagnostic: the chunk packages dcs_batch_create uploads for `jobs` at `fpw` frames per wavefront (4, 8 or 16): per
 * chunk, at a fixed stride of *packageBytesOut bytes, 80 bytes per slot (the slot's first 16 bytes, the first 40 bytes of
 * its first DcsSrcDesc, its pool offset and bands per lane, its 16 stream-header bytes), one split record per lane (8 bytes;
 * 4 when every source is a 1994+ frame) and -- from the next 128-byte boundary -- the image of the kernel's bit pool (the
 * chunk's compressed dwords in bit order), as long as the plan's fullest chunk needs.  out = NULL to size. */
DcsStatus dcs_pack_chunks(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs,
                          const uint8_t *blob, size_t blobLen, int fpw,
                          uint8_t *out, size_t cap, uint32_t *nChunksOut, uint32_t *packageBytesOut);

/* Diagnostic: the same packages as the DEVICE packer assembles them (what a DCS_PIPE_PACK_ON_DEVICE pipeline does: plan on
 * the host from 8-byte source digests, pack kernel on the device from resident records and streams).  Byte for byte what
 * dcs_pack_chunks returns. */
DcsStatus dcs_pack_chunks_device(DcsCtx *ctx, const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, uint32_t nSrcs,
                                 const uint8_t *blob, size_t blobLen, int fpw,
                                 uint8_t *out, size_t cap, uint32_t *nChunksOut, uint32_t *packageBytesOut);

uint32_t dcs_abi_version(void);
/* a digest of the sources and compiler flags this library was built from (16 hex digits).  Counter profiles under
 * profiles/ carry the id of the library they were taken with; bench.py reports them only for a library with that id. */
const char *dcs_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* DCS_HIP_H */

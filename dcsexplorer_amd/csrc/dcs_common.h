// dcs_common.h -- internal declarations shared by the host side and the HIP kernels of libdcs_hip.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/dcs_hip.h"

// ---------------------------------------------------------------------------------------------
// Decode tables as the kernels consume them.  Built once on the host from the canonical code lists
// in dcs_tables.h (dcs_tables.cpp), uploaded at context creation, and staged into LDS by every
// workgroup (the `lds` part) or read through L1/L2 (the rest).
//
// Variable-length codes are decoded with a 256-entry first-level table on the next 8 bits plus a
// binary trie for the (rare) longer codes:
//   fast[peek8]  : bit15 set  -> leaf:  bits 11..8 = code length (1..8), bits 7..0 = payload
//                  bit15 clear -> index of the trie node reached after those 8 bits
//   trie[n]      : bit15 set  -> leaf payload (bits 7..0); clear -> index of the '0' child, the
//                  '1' child is the next entry
// Payload for the 1994+ band-type deltas is delta+16 (0..30); for the 1993b band types it is the
// raw 6-bit leaf value of the format (<0x1E: keep sub-type, value-0x0F; else toggle, value-0x2E).
// ---------------------------------------------------------------------------------------------
#define DCS_CB94_TOTAL   940            // 4+8+32+128+256+512 direct-lookup entries, codebooks 1..6
#define DCS_TRIE94_MAX   64
#define DCS_TRIE93_MAX   128

struct DcsLdsTables
{
    // ---- what the decode kernel stages (the first DCS_LDS_DECODE_BYTES) -----------------------------------------------
    uint16_t cb94[DCS_CB94_TOTAL];      // entry = sample (signed byte, 0 for the two-zeros code) | nBits<<8 | step<<13
                                        // (step = samples the code stands for: 1, or 2 for two zeros; .cpp:2046-2175)
    uint16_t fast93[256];
    uint16_t trie93[DCS_TRIE93_MAX];
    uint16_t bandBits93a[64];           // bandBits (0xFF = end of frame) | prefixBits<<8 (:2878-2902)
    uint16_t scaleCb93a[80];            // value (0xFF = escape) | nBits<<8 | subTable<<12 (:2938-2959)
    uint8_t  inputs93a[24];             // inputs per band, 18 used (:2865)
    uint16_t raw94[20];                 // two-entry "codebooks" of the fixed-width sample codes 7..16: width<<8 | 1<<13
    // Everything the 1994+ band set-up derives from a band-type code (:1886-2005), resolved once (dcs_tables.cpp) so that
    // the kernel's set-up is one look-up: [0..50] Type 1, [band class 0..2][min(code, 16)]; [51..68] Type 0, [min(code, 17)].
    // Entry: the codebook's offset in this block, in half-words (bits 0..10) | 32 - look-ahead width (11..15) | shift
    // that turns the next 32 bits into the codebook index (16..20) | DCS_B94_RAW | _ZERO | _STOP | _FATAL | scale
    // adjustment << 25.
    uint32_t band94[72];
    uint16_t scale64[64];               // scale factor of a band by the low six bits of its scale code (:1978-1979, :2342)
    uint8_t  padDecode_[8];
    // ---- the index walk only (dcs_scan.h; the index kernel stages the whole block) ---------------------------------------
    uint16_t cbInfo[8];                 // per sample code 1..6: maxBits | (base offset into cb94)<<4
    uint16_t xlat94[48];                // [band class 0..2][code] = typeCode | scalingAdj<<8 (:1926-1953)
    uint8_t  preAdj94[32];              // [0..15] sub-type 0 map, [16..31] sub-type 1..3 map (:1744-1749)
    uint16_t scaleMant[4];              // 0x8000, 0x9838, 0xB505, 0xD745 (:1978)
    uint8_t  pad_[8];
};
// (gfx950 hands LDS out in 1 280-byte pieces: with 16 frames per wavefront a workgroup's 4 x 12 576 bytes leave 3 456
// for the tables if three workgroups are to share a CU)
#define DCS_LDS_DECODE_BYTES 3424
#define DCS_B94_RAW      (1u << 21)     // fixed-width samples (sample codes 7..16)
#define DCS_B94_ZERO     (1u << 22)     // nothing coded (:1886)
#define DCS_B94_STOP     (1u << 23)     // sample code 0 behind a non-zero band-type code (:1985-1991)
#define DCS_B94_FATAL    (1u << 24)     // no such code (:1914, :1999)
#define DCS_B94_TYPE0    51             // first Type-0 entry

// per-lane constants of the transform passes (dcs_kernels.hip.h): twiddles and overlap-window entries that
// depend only on the lane number, precomputed on the host so that a wavefront fetches them with six
// 16-byte loads per lane instead of ~40 scattered table reads.  One 24-dword record per transform:
//   1994+ (lane94[l], l = lane & 7 matters):  [0..7]  pre-twiddle of pair i = l + 8j: c0 | c1 << 16 (.cpp:428-429)
//                                             [8..21] layout-B stages d=4 [8..9], d=2 [10..13], d=1 [14..21]; cos | sin << 16
//                                             [22]    overlap window of pair m = bitrev3(l): co[2m] | co[2m+1] << 16
//                                             [23]    co[15-2m] | co[14-2m] << 16
//   1993  (lane93[l], l = lane & 15 matters): [0..14] layout-B stages d=8 [0], d=4 [1..2], d=2 [3..6], d=1 [7..14]
//                                             [15]    overlap window of sample i = bitrev4(l): co[i] | co[15-i] << 16
#define DCS_LANE_CONSTS 24
#define DCS_K94_PRE   0
#define DCS_K94_TWB   8
#define DCS_K94_OVLA  22
#define DCS_K94_OVLB  23
#define DCS_K93_TWB   0
#define DCS_K93_OVL   15

struct DcsDevTables
{
    DcsLdsTables lds;                   // copied to LDS by each workgroup
    uint32_t lane94[64][DCS_LANE_CONSTS];
    uint32_t lane93[64][DCS_LANE_CONSTS];
    uint16_t fast94[256];               // 1994+ band-type delta code: only the host index pass reads it
    uint16_t trie94[DCS_TRIE94_MAX];
    uint16_t pair93a[2048];             // OS93a Type-1 sample pair table (:2698-2827); read via L1/L2
    uint16_t fftCoef[256];              // sin block 0..0x7F, cos block 0x80..0xFF, bit-reversed order (:366)
    uint16_t ovlCoef[16];               // overlap window (:314)
    int32_t  twA[8][4];                 // twiddles 0..7 as the in-lane (layout A) butterflies take them: 2 cos, 2 sin, -2 sin, 0
    // The device index pass (dcs_index_wave.hip.h) takes several 1994+ sample codes per step: per codebook 1..6 and for the
    // next DCS_IDX_MULTI_BITS bits, the codes that lie entirely inside them, as long as they stand for at most
    // DCS_IDX_MULTI_SAMPLES samples together (at least one code): total length | samples << 4
    uint8_t  multi94[6][1 << 10];
};
#define DCS_IDX_MULTI_BITS 10
#define DCS_IDX_MULTI_SAMPLES 4

// host-side view (same structure; one process-wide immutable instance)
const DcsDevTables &dcsTables();

// MainLoop's shared fixed-point scale with a per-channel master multiplier (dcs_params.cpp)
int dcsFrameScaleV(const uint16_t *vol, uint16_t *mixMul, const uint8_t *counted, int nch);

// ---------------------------------------------------------------------------------------------
// Kernel work list.  The planner (dcs_plan.cpp) cuts the job list into chunks of at most FPW slots;
// one wavefront decodes one chunk.  A slot is a job to decode; HALO slots are decoded only for the
// 16-sample tail they hand to a later slot of the same chunk (the predecessor of the chunk's first
// frame lives in another chunk).
// ---------------------------------------------------------------------------------------------
// LDS bit pool: the compressed bytes of the frames one wavefront unpacks in one round are staged
// there.  The planner closes a chunk before the pool would overflow.
#define DCS_POOL_DW_PER_FRAME 56        // 224 bytes per frame slot on average (typical frame: ~125-150 bytes)
#define DCS_POOL_DW_MIN       320       // but never less than two maximal frames (a halo and its successor)
#define DCS_MAX_FRAME_BITS    4480      // 16 band headers + 255 x 16-bit samples, rounded up

#ifdef __cplusplus
// dwords of pool one source occupies: whole dwords covering the frame + 3 dwords of window look-ahead
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t dcsPoolDwords(uint64_t streamOff, uint32_t hdrLen, uint32_t bitOff, uint32_t nBits)
{
    const uint32_t inDword = static_cast<uint32_t>(((streamOff + 2 + hdrLen) * 8 + bitOff) & 31);
    return (inDword + nBits + 31) / 32 + 3;
}
#endif

#ifdef __cplusplus
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr uint32_t dcsPoolCapacity(int fpw)
{
    return static_cast<uint32_t>(fpw * DCS_POOL_DW_PER_FRAME > DCS_POOL_DW_MIN ? fpw * DCS_POOL_DW_PER_FRAME : DCS_POOL_DW_MIN);
}
#endif

#define DCS_SLOT_HALO      0x01u        // do not write PCM / err for this slot
#define DCS_SLOT_EXT_TAIL  0x02u        // overlap tail comes from tailsIn[job.prev & 0x7FFFFFFF]
#define DCS_SLOT_EXPORT    0x04u        // this frame's tail meets a frame of another chunk (job nextJob) at handoff[this chunk]
#define DCS_SLOT_IMPORT    0x08u        // overlap tail meets this frame at handoff[prevJob] (prevJob = the chunk of its predecessor)
#define DCS_SLOT_KEEP_TAIL 0x10u        // store this frame's tail in tailsOut: the last frame of its chain in the batch (what a caller
                                        // needs to carry a stream into its next batch), or every frame when the batch keeps all tails
#define DCS_SLOT_EMPTY     0x80u        // padding
#define DCS_NO_PREV_SLOT   0xFFu

struct DcsSlot                          // 32 bytes: everything the kernel needs to know about a job, so
{                                       // that the job list itself is never read on the device
    uint32_t job;                       // output index (PCM row, err entry)
    uint8_t  prevSlot;                  // slot index inside the chunk whose tail overlaps into this one
    uint8_t  flags;
    uint8_t  nSrc;
    uint8_t  shiftXform;                // volShift | xform << 4
    uint32_t firstSrc;
    uint32_t prevJob;                   // DcsFrameJob.prev (external-tail index when DCS_SLOT_EXT_TAIL)
    // Unpack round 0 (the FIRST source of every job), worked out by the planner so that nothing of it waits for
    // the descriptor.  The compressed bytes of a chunk's frames mostly lie back to back in the blob (consecutive
    // frames of one stream), so they are staged as RUNS of dwords, 16 bytes per lane: slot k of a chunk carries
    // run k (runNDw == 0: no further run), which has nothing to do with slot k's own frame.
    uint32_t runStartDw;                // first blob dword of run k
    uint16_t runNDw;                    // its length in dwords (whole frames + 3 dwords of window look-ahead)
    uint16_t poolOff;                   // THIS slot's frame: pool dword that holds its first bit
    uint32_t nextJob;                   // DCS_SLOT_EXPORT: the job whose first 16 samples this frame's tail overlaps into (round 6; the
                                        // field held the blob dword of the stream header, which both packers take from the source)
    uint8_t  pad_;
    uint8_t  bpl;                       // header bands per unpack lane, ceil(min(nBands, 16) / (64 / fpw)); 0: one lane
                                        // unpacks the whole frame (DCS_IDX_SERIAL)
    uint16_t runPoolOff;                // pool dword where run k goes (a multiple of 4)
};

// Which header bands the q-th unpack lane of a frame takes: lane q starts at dcsLaneFirstBand(q) and ends where lane
// q + 1 starts.  The 1993 layouts (sixteen bands of sixteen samples) get bpl consecutive bands per lane.  The bands of a
// 1994+ frame hold 7, 8, 13 x 16 and 32 samples: there bands 0 and 1 count as one and band 15 as two, which with eight
// lanes gives {0, 1, 2} {3, 4} ... {13, 14} {15}, 31 or 32 samples for every lane (the symbol loop works through them in
// rounds of 7, 9 and 16 samples, unpack94 in dcs_kernels.hip.h).  With sixteen lanes it is {0, 1} {2} ... {14} {15} and
// the last lane is left for the second half of band 15, which the packers give it when the index pass recorded where
// that half starts (dcsMid15: split[14].prv / .prvDelta, dcs_scan.h).
#ifdef __cplusplus
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr int dcsLaneFirstBand(int format, int q, int bpl, int nbEnd)
{
    // OS93a Type 1: eighteen bands of 2, 2, 2, 2, 3, 4, 5, 6, 5, 6, 7, 9, 11, 14, 12, 12, 12, 13 sample pairs
    // (DCSDecoderNative.cpp:2865).  The lanes of a wavefront walk their k-th bands together, so what counts is the
    // longest k-th band: {0,1,2} {3,4,5} {6,7} {8,9} ... {16,17} with eight lanes (12 + 14 + 4 pairs; two bands per
    // lane in order cost 12 + 14 + 12 + 13), {0,1} {2,3} {4} {5} ... {17} with sixteen, {0..6} {7..10} {11..13} {14..17}
    // with four.  (nbEnd: 18 or the stream's own band count.)
    if (format == DCS_FMT_93A_T1)
    {
        const int b = bpl == 1 ? (q < 2 ? 2 * q : q + 2) : bpl == 2 ? (q < 2 ? 3 * q : 2 * q + 2) : (q == 0 ? 0 : q == 1 ? 7 : q == 2 ? 11 : 14);
        return b < nbEnd ? b : nbEnd;
    }
    const int b = q * bpl + ((format >= DCS_FMT_94_T0 && q != 0) ? 1 : 0);
    return b < nbEnd ? b : nbEnd;
}
// the band where the lanes' dealing ends: sixteen header bands, eighteen for OS93a Type 1
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr int dcsDealEnd(int format, int nBands)
{
    return format == DCS_FMT_93A_T1 ? (nBands < 18 ? nBands : 18) : (nBands < 16 ? nBands : 16);
}
// OS93a Type 1: a lane's first band can be 16 or 17; then this bit of its state word is set and bits 12..15 hold band - 16
// (the record itself comes from the frame record's bandType bytes, dcs_scan.h)
#define DCS_SPLIT_BASE16 0x200u
// state word of a lane that starts in the middle of band 15: output index | DCS_MID15_STRADDLE (bit 9) | this flag
#define DCS_SPLIT_MID15 0x800u
#define DCS_MID15_STRADDLE 0x200u
// ... and whether the frame's last lane does: one band per lane, all sixteen bands, a recorded middle
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr bool dcsMid15(int format, int bpl, int nb16, uint32_t midBits)
{
    return format >= DCS_FMT_94_T0 && bpl == 1 && nb16 == 16 && midBits != 0;
}
#endif

// Chunk packages.  Everything unpack round 0 of a chunk needs, gathered once per batch by the host packer
// (dcsBuildPackages, dcs_plan.cpp) or the device packer (dcsPackKernel) into one block at a fixed stride, so that a wavefront
// requests ALL of it at its first instruction (no load depends on another load).  Round 5 layout (a wavefront reads its whole
// package, so every byte of it counts as HBM traffic):
//   [0, fpw x 80)   per slot five 16-byte pieces: the slot (DcsSlot bytes 0..15: job, prevSlot | flags | nSrc | shiftXform,
//                   firstSrc, prevJob) | descriptor head bytes 0..15 | 16..31 | 32..39 followed by poolOff (u16), bpl (u8),
//                   a spare byte and nextJob (u32) | the stream header (16 B, a 1-byte header zero-extended)
//   [fpw x 80, ..)  the split record of every lane [64]: 8 bytes (zero for a frame's first lane; the lane's first band in bits
//                   12..15 of its state word, bit 15 of bitDelta: no bands) -- or, when every source of the batch is a 1994+
//                   frame, 4 bytes: bitDelta | state << 16 (those layouts carry nothing in prv / prvDelta but band 15's middle,
//                   which the packers fold into the two halves)
//   [dcsPkgOffPool, + imgDw x 4)  the image of the bit pool (runs placed, dwords in bit order), as long as the batch's fullest
//                   chunk needs, a multiple of 128 bytes (a HOST-planned batch; one planned on the device has the pool's capacity:
//                   its stride would have to come out of device memory, a dependent load in front of the package loads)
// The layout word: image dwords | DCS_PKG_SPLIT4; it travels to the kernel in bits 16..31 of its flags.
#ifdef __cplusplus
#define DCS_PKG_SLOT_BYTES 80u
#define DCS_PKG_SPLIT4     0x8000u
#ifdef __HIPCC__
#define DCS_HD __host__ __device__
#else
#define DCS_HD
#endif
static inline DCS_HD constexpr uint32_t dcsPkgImgDw(uint32_t layout) { return layout & 0x7FFFu; }
static inline DCS_HD constexpr uint32_t dcsPkgSplitBytes(uint32_t layout) { return (layout & DCS_PKG_SPLIT4) ? 4u : 8u; }
static inline DCS_HD constexpr uint32_t dcsPkgOffSplit(int fpw) { return static_cast<uint32_t>(fpw) * DCS_PKG_SLOT_BYTES; }
static inline DCS_HD constexpr uint32_t dcsPkgOffPool(int fpw, uint32_t layout)
{
    return (static_cast<uint32_t>(fpw) * DCS_PKG_SLOT_BYTES + 64u * dcsPkgSplitBytes(layout) + 127u) & ~127u;
}
static inline DCS_HD constexpr uint32_t dcsPkgStride(int fpw, uint32_t layout) { return dcsPkgOffPool(fpw, layout) + dcsPkgImgDw(layout) * 4u; }
#endif

struct DcsKernelArgs
{
    const uint8_t      *blob;
    uint64_t            blobLen;        // bytes that may be read (allocation is padded beyond this)
    const DcsSrcDesc   *srcs;
    uint8_t            *packages;       // nChunks x dcsPkgStride(fpw, layout), see above
    uint32_t            nChunks;
    uint32_t            nJobs;
    int16_t            *pcm;            // nJobs x 240
    uint32_t           *err;            // nJobs
    const int16_t      *tailsIn;        // k x 16 (may be null)
    int16_t            *tailsOut;       // nJobs x 16 (may be null)
    const DcsDevTables *tables;
    unsigned long long *debug;          // diagnostic builds only (DCS_STAMPS); null otherwise
    // tails between chunks (the rendezvous, dcs_kernels.hip.h): nChunks x 16 words of epoch << 33 | who << 32 | payload; a word
    // belongs to this launch when its epoch equals `epoch` (the launch counter, 1 .. 2^31 - 1), so the buffer is never cleared
    unsigned long long *handoff;
    uint32_t            epoch;
    uint32_t            flags;          // DCS_BATCH_*
};
#define DCS_EPOCH_MAX 0x7FFFFFFFu
#define DCS_BATCH_HAS_93A_T1 1u         // some source is an OS93a Type-1 frame: workgroups stage the pair table in LDS
// The chunks are in CHAIN order (a chunk takes its tail from the chunk before it) and the launch maps them to workgroups in XCD
// RANGES: workgroup i of a launch runs on XCD i % 8 (measured: tools/xcd_map.hip), so with logical workgroup
// L = (i % 8) * R + i / 8, R = workgroups / 8, XCD j decodes logical workgroups [j R, (j + 1) R) in order, and the producer of a tail
// sits in the same workgroup or in the one dispatched just before it ON THE SAME XCD.  Every wait is then for a wavefront that is
// resident or through whatever else runs on the chip -- other decode kernels included (dcs_pipeline.hip.h: why that matters); only
// the first workgroup of a range may wait for the last one of the range before, i.e. until that XCD is through.
#define DCS_BATCH_XCD_RANGES 2u
// set by the launch, not by the batch: more than one wavefront per SIMD, so the wavefronts arrange their priorities (s_setprio,
// dcs_kernels.hip.h); bits 8..15 then hold CUs / 8 (CUs x 4 workgroups are resident at once)
#define DCS_BATCH_PACED 4u
#define DCS_BATCH_CUS8_SHIFT 8
#define DCS_BATCH_IMG_SHIFT 16          // bits 16..31: the packages' layout word (image dwords | DCS_PKG_SPLIT4; dcsPkgStride)
#define DCS_BATCH_IMG_MASK  0xFFFFu

// A source as the planner and the DEVICE packer need it when the index records stay on the device (the pipeline's
// device path): 24 bytes instead of the 160 of DcsSrcDesc.  `record` = index of the frame's DcsFrameIndex in the
// device-resident record array.
struct DcsPlanSrc
{
    uint64_t streamOff;
    uint32_t bitOff;
    uint16_t nBits;
    uint8_t  hdrLen, nBands, flags, format;
    uint16_t mixMul;
    uint32_t record;
};
// What the DEVICE planner (dcsPlanKernel, dcs_runtime.hip) is told about a stream of a list of whole streams -- everything
// the host knows without walking the stream: where it lies, its frame count, layout and mixing parameters.  40 bytes.
struct DcsPlanStream
{
    uint64_t streamOff;                 // offset of the stream in the list's blob
    uint32_t len;                       // bytes that belong to it
    uint32_t firstRecord;               // its records in the list's record array
    uint32_t firstJob;                  // its first output frame
    uint32_t nFrames;                   // the stream's U16 frame count (output frames: nFrames + extraFrames)
    uint16_t mixMul0, mixMulN;          // rescaled mixing multiplier of frame 0 / of every later frame (dcs_stream_params_from)
    uint8_t  volShift0, volShiftN;
    uint8_t  xform, hdrLen, format, pad_[3];
};
static_assert(sizeof(DcsPlanStream) == 40, "DcsPlanStream layout (uploaded by copy kernel: whole dwords)");
#define DCS_PLAN_POOL_OVERFLOW 1u       // flag word of the device planner: some chunk's compressed bytes do not fit the bit pool
#define DCS_PLAN_TRUNCATED     2u       // ... some stream's frames run past its buffer
// what the index kernel writes per frame next to the full record: all the host needs for planning (8 bytes)
struct DcsFrameDigest
{
    uint32_t bitOff;
    uint16_t nBits;
    uint8_t  nBands, flags;
};

// planner: returns the number of chunks; slots is resized to nChunks * fpw
#ifdef __cplusplus
#include <vector>
// whole streams -> batch description (dcs_streams.cpp): index pass on the host pool, per-frame mixing parameters, the
// streams laid out back to back in one blob, one job per output frame
struct DcsBuiltStreams
{
    std::vector<uint8_t> blob;
    std::vector<DcsSrcDesc> srcs;
    std::vector<DcsFrameJob> jobs;
    std::vector<uint32_t> firstJob;     // per stream, plus a final total
};
// index records made elsewhere (the device index pass) for streams already laid out in a blob of the caller's: the build
// then neither walks the streams nor copies them (B.blob stays empty; sources point into the caller's blob)
struct DcsPreIndexed
{
    const DcsFrameIndex *records;       // stream k's records at records + firstRecord[k]
    const uint64_t *firstRecord;
    const DcsStreamInfo *infos;
    const uint64_t *streamOff;          // stream k's offset in the caller's blob; NULL: the build lays the streams out itself
};
DcsStatus dcsBuildStreams(const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, DcsBuiltStreams &B,
                          bool countOnly, bool sequence, const DcsPreIndexed *pre = nullptr);
// The same batch description in its light form, for packing on the device: jobs as above, sources as 24-byte
// digests (DcsPlanSrc) that name their index record by position (`recordBase` + the stream's first record + frame).
struct DcsDigested
{
    const DcsFrameDigest *digest;       // stream k's frames at digest + firstRecord[k]
    const uint64_t *firstRecord;
    const DcsStreamInfo *infos;
    const uint64_t *streamOff;          // stream k's offset in the uploaded blob
    uint32_t recordBase;                // where this list's records start in the device-resident record array
};
#include <functional>
DcsStatus dcsIndexStreamsNotify(const DcsStreamRef *streams, uint32_t nStreams, int nThreads,
                                DcsFrameIndex *out, const uint64_t *firstRecord, DcsStreamInfo *infos,
                                const std::function<void(uint32_t)> *done);
bool dcsIndexPoolBusy();
// the host walk with its records handed over frame by frame, and the container part of it alone (dcs_index.cpp)
DcsStatus dcsIndexStreamProgressive(DcsOsVersion os, const uint8_t *stream, size_t len, DcsStreamInfo *info,
                                    const std::function<void(uint32_t, const DcsFrameIndex &)> &onFrame);
DcsStatus dcsStreamContainer(DcsOsVersion os, const uint8_t *stream, size_t len, DcsStreamInfo *info);
// large lists through the context's own pipeline, in parts (dcs_pipeline.hip.h); *handled = false: take the direct path
DcsStatus dcsDecodeStreamsInParts(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames,
                                  int16_t *pcmOut, size_t pcmCapFrames, uint32_t *frameOffsets, uint32_t *errOut, bool *handled);
struct DcsBuiltPlan
{
    std::vector<DcsFrameJob> jobs;
    std::vector<DcsPlanSrc> srcs;
    std::vector<uint32_t> firstJob;
};
DcsStatus dcsBuildPlanFromDigest(const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, const DcsDigested &in,
                                 DcsBuiltPlan &P);
// (depthOrder = false: the chunks stay in chain order, for launches in XCD ranges -- DCS_BATCH_XCD_RANGES)
// dwords of pool image the packages of a plan need: the fullest chunk's runs, rounded up to 32 dwords, at most the pool's capacity
uint32_t dcsImageDwords(const DcsSlot *slots, uint32_t nChunks, int fpw);
bool dcsAllSources94(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs);
// (keepAllTails: every frame's slot gets DCS_SLOT_KEEP_TAIL, else only the last frame of every chain)
// (shuffleSeed != 0, a test hook: the chunks in a seeded random order -- the rendezvous between chunks must not care)
uint32_t dcsPlanChunks(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff = true,
                       int framesPerChunk = 0, bool depthOrder = true, bool keepAllTails = false, uint32_t shuffleSeed = 0);
void dcsShuffleChunks(std::vector<DcsSlot> &slots, uint32_t nChunks, int fpw, uint32_t seed);
// (places: wavefronts of the decode kernel the chip runs at a time -- CUs x 16 -- or 0; the diagnostic entries assume an MI355X)
#define DCS_MI355X_WAVE_PLACES 4096u
uint32_t dcsPlanChunksCapped(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                             int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t *imgDwOut, uint32_t places);
uint32_t dcsPlanChunksCappedLite(const DcsFrameJob *jobs, uint32_t nJobs, const DcsPlanSrc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                                 int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t *imgDwOut, uint32_t places);
uint32_t dcsPlanChunksLite(const DcsFrameJob *jobs, uint32_t nJobs, const DcsPlanSrc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff = true,
                           int framesPerChunk = 0, bool depthOrder = true, bool keepAllTails = false);
// packer: out = nChunks x dcsPkgStride(fpw, layout) bytes (the chunk packages described above); layout = image dwords | DCS_PKG_SPLIT4
void dcsBuildPackages(const DcsSlot *slots, uint32_t nChunks, int fpw, const DcsSrcDesc *srcs,
                      const uint8_t *blob, size_t blobLen, uint8_t *out, uint32_t layout);
#endif

/* oracle/dcs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's DCS frame-decode hot path
 * (DCSDecoderNative: bitstream unpack -> dequantise -> inverse transform ->
 * overlap-add -> int16 PCM).  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * shipped library (dcsexplorer_amd/) never includes, links or calls anything
 * in this directory.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py compares every function
 * below against the unmodified reference compiled into oracle/_ref/libdcsref.so
 * (same entry-point signatures as oracle/ref_driver.cpp, so the two libraries
 * are interchangeable in the tests), and tests/test_oracle_golden.py compares
 * it against the committed fixtures in tests/golden/ that were produced by
 * that reference build (tests/golden/make_golden.py).
 *
 * All `file:line` citations are relative to /root/reference/DCSDecoder/.
 */
#ifndef DCS_ORACLE_H
#define DCS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* OS versions, same numbering as oracle/ref_driver.cpp */
enum { ORC_OS93A = 0, ORC_OS93B = 1, ORC_OS94 = 2, ORC_OS95 = 3 };

/* per-frame error flags */
#define ORC_ERR_STOP   1   /* the reference's `channel.stop` (DCSDecoderNative.cpp:1989, :2216) */
#define ORC_ERR_FATAL  2   /* malformed beyond what the reference defines (UB there): band type code
                              out of table range, sample width > 16 bits, > 18 bands in a 93a-T1
                              frame.  Our defined behaviour: stop decoding the frame at that point,
                              keep what was accumulated, raise STOP as well. */

/* identical layout to RefProbe in ref_driver.cpp */
typedef struct OrcProbe
{
    int32_t  active;
    int32_t  bitOff;
    uint16_t mixMul;
    uint16_t volMult;
    uint16_t bandType[16];
} OrcProbe;

/* Whole-decoder player: nch streams loaded on channels 0..nch-1 of a freshly
 * constructed decoder, nFramesOut x 240 samples pulled.  Follows
 * DCSDecoderNative::MainLoop (:89-306) minus the track-program VM. */
int orc_decode(int os, int volume, int nch,
    const uint8_t *const *streams, const size_t *lens, const int *levels,
    int nFramesOut, int16_t *pcm, OrcProbe *probes);

/* The --extract-streams loop on ONE decoder object (DCSExplorer.cpp:1628-1907): per stream
 * LoadAudioStream(0, ptr, level), nFrames + extraFrames frames, ClearTracks after each of the last two.
 * pcm = all frames back to back. */
int orc_decode_sequence(int os, int volume, int n, const uint8_t *const *streams, const size_t *lens,
    const int *levels, int extraFrames, int16_t *pcm);

/* DCSDecoderNative::GetStreamInfo (:1486-1537) */
int orc_stream_info(int os, const uint8_t *stream, size_t len,
    int *nFrames, int *nBytes, int *formatType, int *formatSubType, uint8_t *header16);

/* DecoderImpl*::TransformFrame on a caller-supplied 512-word frame buffer (:397-813) */
int orc_transform(int os, uint16_t *frameBuf512, int volShift,
    uint16_t *overlap16, int16_t *pcm240);

/* DecoderImpl*::DecompressFrame x nFrames, each into a fresh zeroed buffer (:1679-3032) */
int orc_decompress(int os, const uint8_t *stream, size_t len, uint16_t mixMul,
    int nFrames, uint16_t *out, int32_t *bitOffs, uint16_t *bandTypes, int32_t *stopFlags);

/* SetMasterVolume (:3250-3282) and UpdateMixingLevels (:3072-3121) arithmetic */
uint16_t orc_volume_multiplier(int volume);
uint16_t orc_mixing_multiplier(int os, int levelSum, int channelVolume);

/* oracle-only extras (no reference twin): the per-frame parameters MainLoop derives
 * (:227-269) for a single stream on channel 0 of a fresh decoder.  mixMulScaled[f]
 * and volShift[f] are what DecompressFrame / TransformFrame receive for frame f. */
int orc_frame_params(int os, int volume, int level, int nFrames,
    uint16_t *mixMulScaled, uint8_t *volShift);

/* CPU-baseline helper: n single-channel streams, each from a fresh decoder, `repeat` times on nThreads
 * threads (range partition).  Returns frames decoded.  Same contract as ref_decode_many. */
long long orc_decode_many(const int *os, const int *volume, const int *level,
    const uint8_t *const *streams, const size_t *lens, int n, int repeat, int nThreads);

/* FNV-1a 64 over bytes (offset 0xcbf29ce484222325, prime 0x100000001b3) */
uint64_t orc_fnv1a64(const void *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif

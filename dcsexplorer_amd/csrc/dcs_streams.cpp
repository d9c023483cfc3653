// dcs_streams.cpp -- whole-stream convenience on top of the batch ABI: the shape of the reference's
// only batch decode, `DCSExplorer --extract-streams` (DCSExplorer.cpp:1628-1907): every stream is
// played alone through LoadAudioStream(0, ptr, level) (DCSDecoderNative.cpp:1387) and pulled for
// nFrames (+ taper) frames of 240 samples (ExtractToWAV, DCSExplorer.cpp:1670-1721).  Here each
// stream starts from a freshly constructed decoder (zero overlap tail, first-frame multiplier 0x7FFF).
#include "dcs_common.h"
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>

typedef DcsBuiltStreams Built;

// sequence: the streams are played one after the other by ONE decoder (dcs_decode_stream_sequence)
DcsStatus dcsBuildStreams(const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, DcsBuiltStreams &B,
                          bool countOnly, bool sequence, const DcsPreIndexed *pre)
{
    std::vector<uint16_t> mm;
    std::vector<uint8_t> vs;
    uint64_t total = 0, totalSrc = 0;
    std::vector<uint64_t> firstRecord(nStreams);
    // (B may be a caller's scratch that has served earlier lists: its vectors keep their memory, which matters when
    // many lists are prepared at once -- fresh multi-megabyte vectors per list mean page faults by the thousand)
    B.firstJob.clear();
    B.blob.clear();
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        const DcsStreamRef &sr = streams[k];
        if (sr.data == nullptr || sr.len < 3 || sr.os < DCS_OS93A || sr.os > DCS_OS95)
            return DCS_ERR_INVALID_ARG;
        const uint32_t nFrames = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
        if (nFrames == 0)
            return DCS_ERR_BAD_STREAM;
        B.firstJob.push_back(static_cast<uint32_t>(total));
        firstRecord[k] = totalSrc;
        total += nFrames + extraFrames;
        totalSrc += nFrames;
    }
    B.firstJob.push_back(static_cast<uint32_t>(total));
    if (countOnly)
        return DCS_OK;

    // the index pass over all streams, on the host worker pool (dcs_index.cpp) -- unless the caller brings the records
    std::vector<DcsFrameIndex> ownIdx;
    std::vector<DcsStreamInfo> ownInfos;
    DcsStatus st = DCS_OK;
    if (pre == nullptr)
    {
        ownIdx.resize(totalSrc);
        ownInfos.resize(nStreams);
        st = dcs_index_streams(streams, nStreams, 0, ownIdx.data(), firstRecord.data(), ownInfos.data());
        if (st != DCS_OK)
            return st;
    }
    const DcsFrameIndex *allIdx = pre ? pre->records : ownIdx.data();
    const DcsStreamInfo *infos = pre ? pre->infos : ownInfos.data();
    if (pre != nullptr)
        for (uint32_t k = 0 ; k < nStreams ; ++k)
            firstRecord[k] = pre->firstRecord[k];
    if (total > 0xFFFFFFFFull)
        return DCS_ERR_CAPACITY;
    // every element is written in full below, so elements left over from an earlier list need no clearing
    if (B.jobs.size() != total) B.jobs.resize(total);
    if (B.srcs.size() != totalSrc) B.srcs.resize(totalSrc);
    uint32_t nJobsOut = 0, nSrcsOut = 0;

    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        const DcsStreamRef &sr = streams[k];
        const DcsOsVersion os = static_cast<DcsOsVersion>(sr.os);
        const uint32_t nFrames = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
        const DcsFrameIndex *idx = allIdx + firstRecord[k];
        const DcsStreamInfo &info = infos[k];
        mm.resize(nFrames); vs.resize(nFrames);
        uint16_t firstMul = 0x7FFF;
        if (sequence && k != 0)
        {
            // what UpdateMixingLevels left behind after the previous stream's last tick: its level, or 0 when
            // the forced-stop sweep reset the mixer after an error (:95-116)
            const DcsStreamRef &pr = streams[k - 1];
            const DcsStreamInfo &pi = infos[k - 1];
            const bool stopped = pi.nValidFrames < pi.nFrames;      // cut short; an error in the very last frame
                                                                    // finds the channel already idle (:100-113)
            firstMul = dcs_mixing_multiplier(os, stopped ? 0 : pr.level * 64, pr.channelVolume);
        }
        st = dcs_stream_params_from(os, sr.volume, sr.level, sr.channelVolume, firstMul, nFrames, mm.data(), vs.data());
        if (st != DCS_OK)
            return st;

        uint64_t streamOff;
        if (pre != nullptr && pre->streamOff != nullptr)
            streamOff = pre->streamOff[k];
        else
        {
            // streams are laid out back to back, each starting on a 4-byte boundary
            while (B.blob.size() & 3)
                B.blob.push_back(0);
            streamOff = B.blob.size();
            // only the bytes the stream uses (the caller's buffer may be the whole rest of a ROM image) ...
            const size_t used = static_cast<size_t>(info.nBytes) < sr.len ? static_cast<size_t>(info.nBytes) : sr.len;
            B.blob.insert(B.blob.end(), sr.data, sr.data + used);
            // ... and a damaged or truncated stream may run past its buffer: bytes past the end read as zero (that
            // is what the index pass assumed), not as the start of the next stream
            if (static_cast<size_t>(info.nBytes) > sr.len)
                B.blob.insert(B.blob.end(), static_cast<size_t>(info.nBytes) - sr.len, 0);
        }

        const uint8_t xform = (os == DCS_OS93A || os == DCS_OS93B) ? DCS_XFORM_93 : DCS_XFORM_94;
        const uint32_t nValid = static_cast<uint32_t>(info.nValidFrames);
        for (uint32_t f = 0 ; f < nFrames + extraFrames ; ++f)
        {
            DcsFrameJob &jb = B.jobs[nJobsOut];
            jb.firstSrc = 0; jb.flags = 0; jb.reserved = 0;
            jb.xform = xform;
            jb.prev = (f == 0 && !(sequence && k != 0)) ? DCS_PREV_NONE : nJobsOut - 1;
            if (f < nValid)
            {
                DcsSrcDesc &sd = B.srcs[nSrcsOut];
                static_assert(sizeof(DcsSrcDesc) == 12 + sizeof(DcsFrameIndex), "DcsSrcDesc has no padding to clear");
                sd.streamOff = streamOff;
                sd.mixMul = mm[f];
                sd.format = static_cast<uint8_t>(info.format);
                sd.hdrLen = static_cast<uint8_t>(info.hdrLen);
                sd.idx = idx[f];
                jb.firstSrc = nSrcsOut++;
                jb.nSrc = 1;
                jb.volShift = vs[f];
            }
            else
            {
                // no active channel: zero spectrum, MainLoop's shift clamps to 8 (:253-260); the frame
                // still carries the predecessor's overlap tail (the "taper" frame)
                jb.nSrc = 0;
                jb.volShift = 8;
            }
            ++nJobsOut;
        }
    }
    B.srcs.resize(nSrcsOut);            // (streams cut short by an error have fewer sources than frames)
    return DCS_OK;
}

// dcsBuildStreams for independent streams when the index records stay on the device: same jobs, 24-byte source digests
DcsStatus dcsBuildPlanFromDigest(const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, const DcsDigested &in,
                                 DcsBuiltPlan &P)
{
    uint64_t total = 0, totalSrc = 0;
    P.firstJob.clear();
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        const DcsStreamRef &sr = streams[k];
        if (sr.data == nullptr || sr.len < 3 || sr.os < DCS_OS93A || sr.os > DCS_OS95)
            return DCS_ERR_INVALID_ARG;
        const uint32_t nFrames = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
        if (nFrames == 0)
            return DCS_ERR_BAD_STREAM;
        P.firstJob.push_back(static_cast<uint32_t>(total));
        total += nFrames + extraFrames;
        totalSrc += nFrames;
    }
    P.firstJob.push_back(static_cast<uint32_t>(total));
    if (total > 0xFFFFFFFFull)
        return DCS_ERR_CAPACITY;
    if (P.jobs.size() != total) P.jobs.resize(total);
    if (P.srcs.size() != totalSrc) P.srcs.resize(totalSrc);
    std::vector<uint16_t> mm;
    std::vector<uint8_t> vs;
    uint32_t nJobsOut = 0, nSrcsOut = 0;
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        const DcsStreamRef &sr = streams[k];
        const DcsOsVersion os = static_cast<DcsOsVersion>(sr.os);
        const uint32_t nFrames = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
        const DcsStreamInfo &info = in.infos[k];
        const DcsFrameDigest *dg = in.digest + in.firstRecord[k];
        mm.resize(nFrames); vs.resize(nFrames);
        const DcsStatus st = dcs_stream_params_from(os, sr.volume, sr.level, sr.channelVolume, 0x7FFF, nFrames, mm.data(), vs.data());
        if (st != DCS_OK)
            return st;
        const uint8_t xform = (os == DCS_OS93A || os == DCS_OS93B) ? DCS_XFORM_93 : DCS_XFORM_94;
        const uint32_t nValid = static_cast<uint32_t>(info.nValidFrames);
        for (uint32_t f = 0 ; f < nFrames + extraFrames ; ++f)
        {
            DcsFrameJob &jb = P.jobs[nJobsOut];
            jb.firstSrc = 0; jb.flags = 0; jb.reserved = 0;
            jb.xform = xform;
            jb.prev = f == 0 ? DCS_PREV_NONE : nJobsOut - 1;
            if (f < nValid)
            {
                DcsPlanSrc &sd = P.srcs[nSrcsOut];
                sd.streamOff = in.streamOff[k];
                sd.bitOff = dg[f].bitOff; sd.nBits = dg[f].nBits; sd.nBands = dg[f].nBands; sd.flags = dg[f].flags;
                sd.hdrLen = static_cast<uint8_t>(info.hdrLen);
                sd.format = static_cast<uint8_t>(info.format);
                sd.mixMul = mm[f];
                sd.record = in.recordBase + static_cast<uint32_t>(in.firstRecord[k]) + f;
                jb.firstSrc = nSrcsOut++;
                jb.nSrc = 1;
                jb.volShift = vs[f];
            }
            else
            {
                jb.nSrc = 0;
                jb.volShift = 8;
            }
            ++nJobsOut;
        }
    }
    P.srcs.resize(nSrcsOut);
    return DCS_OK;
}

static DcsStatus buildStreams(const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, Built &B, bool countOnly,
                              bool sequence = false)
{
    return dcsBuildStreams(streams, nStreams, extraFrames, B, countOnly, sequence);
}

extern "C" DcsStatus dcs_count_stream_frames(const DcsStreamRef *streams, uint32_t nStreams,
                                             uint32_t extraFrames, uint64_t *nFramesOut)
{
    if (streams == nullptr || nFramesOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    Built B;
    DcsStatus st = buildStreams(streams, nStreams, extraFrames, B, true);
    if (st == DCS_OK)
        *nFramesOut = B.firstJob.back();
    return st;
}

extern "C" DcsStatus dcs_decode_streams(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams,
                                        uint32_t extraFrames, int16_t *pcmOut, size_t pcmCapFrames,
                                        uint32_t *frameOffsets, uint32_t *errOut)
{
    if (ctx == nullptr || streams == nullptr || nStreams == 0 || pcmOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    // a large list goes through the context's pipeline in parts (host stages of one part overlap device stages of another)
    bool handled = false;
    DcsStatus st = dcsDecodeStreamsInParts(ctx, streams, nStreams, extraFrames, pcmOut, pcmCapFrames, frameOffsets, errOut, &handled);
    if (handled || st != DCS_OK)
        return st;
    Built B;
    st = buildStreams(streams, nStreams, extraFrames, B, false);
    if (st != DCS_OK)
        return st;
    if (B.jobs.size() > pcmCapFrames)
        return DCS_ERR_CAPACITY;
    if (frameOffsets != nullptr)
        memcpy(frameOffsets, B.firstJob.data(), sizeof(uint32_t) * B.firstJob.size());
    return dcs_decode_batch(ctx, B.blob.data(), B.blob.size(), B.srcs.data(), static_cast<uint32_t>(B.srcs.size()),
                            B.jobs.data(), static_cast<uint32_t>(B.jobs.size()), nullptr, 0, pcmOut, errOut, nullptr);
}

extern "C" DcsStatus dcs_decode_stream_sequence(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams,
                                                uint32_t extraFrames, int16_t *pcmOut, size_t pcmCapFrames,
                                                uint32_t *frameOffsets, uint32_t *errOut)
{
    if (ctx == nullptr || streams == nullptr || nStreams == 0 || pcmOut == nullptr || extraFrames < 2)
        return DCS_ERR_INVALID_ARG;
    for (uint32_t k = 1 ; k < nStreams ; ++k)
        if (streams[k].os != streams[0].os || streams[k].volume != streams[0].volume
            || streams[k].channelVolume != streams[0].channelVolume)
            return DCS_ERR_INVALID_ARG;
    Built B;
    DcsStatus st = buildStreams(streams, nStreams, extraFrames, B, false, true);
    if (st != DCS_OK)
        return st;
    if (B.jobs.size() > pcmCapFrames)
        return DCS_ERR_CAPACITY;
    if (frameOffsets != nullptr)
        memcpy(frameOffsets, B.firstJob.data(), sizeof(uint32_t) * B.firstJob.size());
    return dcs_decode_batch(ctx, B.blob.data(), B.blob.size(), B.srcs.data(), static_cast<uint32_t>(B.srcs.size()),
                            B.jobs.data(), static_cast<uint32_t>(B.jobs.size()), nullptr, 0, pcmOut, errOut, nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// Several GPUs: streams are the independent units of the path (a frame needs its stream's earlier frames for
// its bit position and band types, DCSDecoderNative.cpp:1715, :1833), so a list of streams is cut into contiguous
// ranges, one per device, balanced by total frame count; no device needs anything from another.
// ---------------------------------------------------------------------------------------------------------
extern "C" DcsStatus dcs_partition_streams(const uint32_t *frameCounts, uint32_t nStreams, uint32_t nParts,
                                           uint32_t *firstStreamOut)
{
    if (nParts == 0 || firstStreamOut == nullptr || (nStreams != 0 && frameCounts == nullptr))
        return DCS_ERR_INVALID_ARG;
    std::vector<uint64_t> prefix(static_cast<size_t>(nStreams) + 1, 0);
    for (uint32_t k = 0 ; k < nStreams ; ++k)
        prefix[k + 1] = prefix[k] + frameCounts[k];
    const uint64_t total = prefix[nStreams];
    firstStreamOut[0] = 0;
    firstStreamOut[nParts] = nStreams;
    for (uint32_t r = 1 ; r < nParts ; ++r)
    {
        // the cut whose prefix sum lies nearest to r/nParts of the frames, never left of the previous cut
        const uint64_t target = static_cast<uint64_t>((static_cast<unsigned __int128>(total) * r) / nParts);
        size_t i = static_cast<size_t>(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
        if (i > 0 && target - prefix[i - 1] <= prefix[i] - target)
            --i;
        firstStreamOut[r] = std::max(static_cast<uint32_t>(i), firstStreamOut[r - 1]);
    }
    return DCS_OK;
}

// (dcs_decode_streams_sharded: csrc/dcs_node.hip.h -- on the persistent contexts of a node-level object)

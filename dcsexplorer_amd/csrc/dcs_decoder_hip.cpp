// dcs_decoder_hip.cpp -- DCSDecoderHIP: the reference's decoder class surface on top of the C ABI.
// See include/DCSDecoderHIP.h.  The channel bookkeeping below restates what DCSDecoderNative::MainLoop
// and DecodeStream do for streams loaded with LoadAudioStream (DCSDecoderNative.cpp:89-306, :1387-1431,
// :1546-1589); the frame decode itself is dcs_decode_batch -> HIP kernels.
#include "../../include/DCSDecoderHIP.h"
#include <string.h>

namespace dcship {

// ---- DCSDecoder base mirror ------------------------------------------------------------------------------
static std::map<std::string, const DCSDecoder::Registration &> &regMap()
{
    static std::map<std::string, const DCSDecoder::Registration &> m;       // function-local static, like DCSDecoder.cpp:1955-1974
    return m;
}

DCSDecoder::Registration::Registration(const char *name, const char *desc, FactoryFunc factory)
    : name(name), desc(desc), factory(factory)
{
    regMap().emplace(this->name, *this);
}

const std::map<std::string, const DCSDecoder::Registration &> &DCSDecoder::GetRegistrationMap() { return regMap(); }

void DCSDecoder::SoftBoot()
{
    host->BootTimerControl(false);
    sampleCounter = 30000;                                      // force a refill (DCSDecoder.cpp:1527)
    state = Initialize() ? State::Running : State::InitializationError;
}

int16_t DCSDecoder::GetNextSample()
{
    if (state != State::Running)
        return 0;                                               // halted or not booted: silence (DCSDecoder.cpp:1680-1689)
    if (sampleCounter >= DCS_FRAME_SAMPLES)
    {
        MainLoop();
        if (state != State::Running)
            return 0;
        sampleCounter = 0;
    }
    return outputBuffer[sampleCounter++];
}

// the same plug-in seam the reference uses for "native" / "emulator-strict" (DCSDecoderNative.cpp:18)
static DCSDecoder::Registration registration("hip", "MI355X HIP batch decoder",
    [](DCSDecoder::Host *host) -> DCSDecoder * { return new DCSDecoderHIP(host); });

// ---- DCSDecoderHIP -----------------------------------------------------------------------------------------
DCSDecoderHIP::DCSDecoderHIP(Host *host, int deviceId) : DCSDecoder(host), deviceId(deviceId) { }

DCSDecoderHIP::~DCSDecoderHIP()
{
    if (ctx != nullptr)
        dcs_ctx_destroy(ctx);
}

void DCSDecoderHIP::InitStandalone(OSVersion v) { osVersion = v; }

DcsOsVersion DCSDecoderHIP::AbiOs() const
{
    switch (osVersion)
    {
    case OSVersion::OS93a: return DCS_OS93A;
    case OSVersion::OS93b: return DCS_OS93B;
    case OSVersion::OS95:  return DCS_OS95;
    default:               return DCS_OS94;                     // Initialize() picks the 94x codec for everything else (:3157)
    }
}

bool DCSDecoderHIP::Initialize()
{
    if (ctx == nullptr)
    {
        DcsStatus st = dcs_ctx_create(deviceId, &ctx);
        if (st != DCS_OK)
        {
            errorMessage = std::string("HIP decoder unavailable: ") + dcs_last_error(nullptr);
            return false;                                       // InitializationError: GetNextSample returns silence, IsOK() is false
        }
    }
    SetMasterVolume(defaultVolume);                             // DCSDecoderNative.cpp:3206
    return true;
}

void DCSDecoderHIP::SetMasterVolume(int vol)
{
    volumeMultiplier = dcs_volume_multiplier(vol);
    Invalidate();
}

DCSDecoderHIP::StreamInfo DCSDecoderHIP::GetStreamInfo(const ROMPointer &p, size_t maxLen)
{
    StreamInfo out;
    memset(&out, 0, sizeof(out));
    DcsStreamInfo info;
    if (!p.IsNull() && dcs_index_stream(AbiOs(), p.p, maxLen, nullptr, 0, &info) == DCS_OK)
    {
        out.nFrames = info.nFrames;
        out.nBytes = info.nBytes;
        out.formatType = info.formatType;
        out.formatSubType = info.formatSubType;
        memcpy(out.header, info.header, 16);
    }
    return out;
}

void DCSDecoderHIP::LoadAudioStream(int ch, const ROMPointer &p, int mixingLevel, size_t maxLen)
{
    if (ch < 0 || ch >= DCS_MAX_CHANNELS || p.IsNull())         // :1390
        return;
    Invalidate();
    Channel &c = channel[ch];
    const uint32_t nFrames = (static_cast<uint32_t>(p.p[0]) << 8) | p.p[1];
    c.active = false;
    c.stopPending = false;
    c.level = mixingLevel << 6;                                 // :1404
    if (nFrames == 0)
        return;                                                 // nothing to play (:1414)
    c.index.resize(nFrames);
    if (dcs_index_stream(AbiOs(), p.p, maxLen, c.index.data(), nFrames, &c.info) != DCS_OK)
        return;
    c.index.resize(static_cast<size_t>(c.info.nValidFrames));
    // private copy of exactly the bytes the stream uses (+ the bit reader's look-ahead)
    size_t used = static_cast<size_t>(c.info.nBytes) + 8;
    if (used > maxLen)
        used = maxLen;
    c.bytes.assign(p.p, p.p + used);
    c.bytes.resize((used > static_cast<size_t>(c.info.nBytes) ? used : static_cast<size_t>(c.info.nBytes)) + 16, 0);
    c.pos = 0;
    c.active = !c.index.empty();
}

bool DCSDecoderHIP::IsStreamPlaying(int ch)
{
    return ch >= 0 && ch < DCS_MAX_CHANNELS && channel[ch].active;
}

void DCSDecoderHIP::ClearTracks()
{
    Invalidate();
    for (Channel &c : channel)
        c.active = false;
}

// drop the frames decoded ahead and rewind to the state after the last frame handed out
void DCSDecoderHIP::Invalidate()
{
    if (ready.empty())
        return;
    // `rewind` is the decoder state after the last frame that was handed out
    for (int i = 0 ; i < DCS_MAX_CHANNELS ; ++i)
    {
        channel[i].pos = rewind.pos[i];
        channel[i].active = rewind.active[i];
        channel[i].mixMul = rewind.mixMul[i];
        channel[i].level = rewind.level[i];
        channel[i].stopPending = rewind.stopPending[i];
    }
    memcpy(tail, rewind.tail, sizeof(tail));
    ready.clear();
    after.clear();
}

void DCSDecoderHIP::MainLoop()
{
    if (ready.empty())
        PlanAndDecode();
    if (ready.empty())
    {
        state = State::DecoderFatalError;
        errorMessage = std::string("HIP decode failed: ") + (ctx ? dcs_last_error(ctx) : "no context");
        return;
    }
    memcpy(outputBuffer, ready.front().data(), sizeof(outputBuffer));
    rewind = after.front();                                     // committed: this frame has been handed out
    ready.pop_front();
    after.pop_front();
}

// Plan `lookahead` MainLoop ticks from the current channel state and decode them in one launch.
void DCSDecoderHIP::PlanAndDecode()
{
    // state before the first planned tick (for Invalidate)
    for (int i = 0 ; i < DCS_MAX_CHANNELS ; ++i)
    {
        rewind.pos[i] = channel[i].pos;
        rewind.active[i] = channel[i].active;
        rewind.mixMul[i] = channel[i].mixMul;
        rewind.level[i] = channel[i].level;
        rewind.stopPending[i] = channel[i].stopPending;
    }
    memcpy(rewind.tail, tail, sizeof(tail));

    // blob: the loaded streams back to back
    std::vector<uint8_t> blob;
    uint64_t off[DCS_MAX_CHANNELS] = { 0 };
    for (int i = 0 ; i < DCS_MAX_CHANNELS ; ++i)
    {
        if (!channel[i].active)
            continue;
        while (blob.size() & 3) blob.push_back(0);
        off[i] = blob.size();
        blob.insert(blob.end(), channel[i].bytes.begin(), channel[i].bytes.end());
    }

    const DcsOsVersion os = AbiOs();
    const uint8_t xform = (os == DCS_OS93A || os == DCS_OS93B) ? DCS_XFORM_93 : DCS_XFORM_94;
    std::vector<DcsSrcDesc> srcs;
    std::vector<DcsFrameJob> jobs;
    std::vector<Snapshot> snaps;
    for (int t = 0 ; t < lookahead ; ++t)
    {
        // forced-stop sweep (:95-116): a stream that raised an error on the previous tick is gone now, and
        // its mixer level with it
        for (Channel &c : channel)
            if (c.stopPending)
            {
                c.stopPending = false;
                c.level = 0;
            }

        // MainLoop's shared scale over the active channels (:227-269)
        uint16_t mm[DCS_MAX_CHANNELS];
        uint8_t act[DCS_MAX_CHANNELS];
        for (int i = 0 ; i < DCS_MAX_CHANNELS ; ++i)
        {
            mm[i] = channel[i].mixMul;
            act[i] = channel[i].active ? 1 : 0;
        }
        const int volShift = dcs_frame_scale(volumeMultiplier, mm, act, DCS_MAX_CHANNELS);

        DcsFrameJob jb;
        memset(&jb, 0, sizeof(jb));
        jb.firstSrc = static_cast<uint32_t>(srcs.size());
        jb.volShift = static_cast<uint8_t>(volShift);
        jb.xform = xform;
        jb.prev = (t == 0) ? (DCS_PREV_EXT | 0u) : static_cast<uint32_t>(jobs.size() - 1);
        for (int i = 0 ; i < DCS_MAX_CHANNELS ; ++i)           // DecodeStream per channel, in channel order (:272-273)
        {
            Channel &c = channel[i];
            if (!c.active)
                continue;
            DcsSrcDesc sd;
            memset(&sd, 0, sizeof(sd));
            sd.streamOff = off[i];
            sd.mixMul = mm[i];
            sd.format = static_cast<uint8_t>(c.info.format);
            sd.hdrLen = static_cast<uint8_t>(c.info.hdrLen);
            sd.idx = c.index[c.pos];
            srcs.push_back(sd);
            ++jb.nSrc;
            if (++c.pos >= c.index.size())
            {
                c.active = false;                               // end of stream, loop count 1 (:1565-1588); an
                                                                // error frame also ends it (:95-116)
                c.stopPending = c.index.size() < static_cast<size_t>(c.info.nFrames);
            }
        }
        jobs.push_back(jb);

        // UpdateMixingLevels: next tick's multiplier from the channel's level (:3072-3121)
        Snapshot sn;
        for (int i = 0 ; i < DCS_MAX_CHANNELS ; ++i)
        {
            channel[i].mixMul = dcs_mixing_multiplier(os, channel[i].level, 0xFF);
            sn.pos[i] = channel[i].pos;
            sn.active[i] = channel[i].active;
            sn.mixMul[i] = channel[i].mixMul;
            sn.level[i] = channel[i].level;
            sn.stopPending[i] = channel[i].stopPending;
        }
        snaps.push_back(sn);
    }

    std::vector<int16_t> pcm(static_cast<size_t>(lookahead) * DCS_FRAME_SAMPLES);
    std::vector<int16_t> tailsOut(static_cast<size_t>(lookahead) * 16);
    if (blob.empty())
        blob.assign(16, 0);
    DcsStatus st = dcs_decode_batch(ctx, blob.data(), blob.size(), srcs.empty() ? nullptr : srcs.data(),
                                    static_cast<uint32_t>(srcs.size()), jobs.data(), static_cast<uint32_t>(jobs.size()),
                                    tail, 1, pcm.data(), nullptr, tailsOut.data());
    if (st != DCS_OK)
    {
        Invalidate();
        return;
    }
    for (int t = 0 ; t < lookahead ; ++t)
    {
        ready.emplace_back(pcm.begin() + static_cast<size_t>(t) * DCS_FRAME_SAMPLES,
                           pcm.begin() + static_cast<size_t>(t + 1) * DCS_FRAME_SAMPLES);
        memcpy(snaps[static_cast<size_t>(t)].tail, &tailsOut[static_cast<size_t>(t) * 16], sizeof(tail));
        after.push_back(snaps[static_cast<size_t>(t)]);
    }
    memcpy(tail, &tailsOut[static_cast<size_t>(lookahead - 1) * 16], sizeof(tail));
}

bool DCSDecoderHIP::DecodeStreamsBatch(const std::vector<BatchStream> &streams, unsigned extraFrames,
                                       std::vector<int16_t> &pcm, std::vector<uint32_t> *firstFrameOfStream)
{
    if (ctx == nullptr && !Initialize())
        return false;
    std::vector<DcsStreamRef> refs;
    for (const BatchStream &b : streams)
        refs.push_back(DcsStreamRef{ b.data, b.len, static_cast<int32_t>(AbiOs()), b.volume, b.mixingLevel, 0xFF });
    uint64_t nFrames = 0;
    if (dcs_count_stream_frames(refs.data(), static_cast<uint32_t>(refs.size()), extraFrames, &nFrames) != DCS_OK)
        return false;
    pcm.resize(static_cast<size_t>(nFrames) * DCS_FRAME_SAMPLES);
    std::vector<uint32_t> offs(refs.size() + 1);
    DcsStatus st = dcs_decode_streams(ctx, refs.data(), static_cast<uint32_t>(refs.size()), extraFrames,
                                      pcm.data(), static_cast<size_t>(nFrames), offs.data(), nullptr);
    if (firstFrameOfStream != nullptr)
        *firstFrameOfStream = offs;
    return st == DCS_OK;
}

}   // namespace dcship

// dcs_decoder_hip.cpp -- DCSDecoderHIP: the reference's decoder class surface on top of the C ABI.
// See include/DCSDecoderHIP.h.  The channel bookkeeping below restates what DCSDecoderNative::MainLoop
// and DecodeStream do for streams loaded with LoadAudioStream (DCSDecoderNative.cpp:89-306, :1387-1431,
// :1546-1589); the frame decode itself is dcs_decode_batch -> HIP kernels.
#include "../../include/DCSDecoderHIP.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <mutex>

DCSHIP_NAMESPACE_BEGIN

#ifndef DCSHIP_USE_REFERENCE_BASE
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

int16_t DCSDecoder::RefillAndGetSample()
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

#endif

// the same plug-in seam the reference uses for "native" / "emulator-strict" (DCSDecoderNative.cpp:18)
static DCSDecoder::Registration registration("hip", "MI355X HIP batch decoder",
    [](DCSDecoder::Host *host) -> DCSDecoder * { return new DCSDecoderHIP(host); });

// ---- contexts ------------------------------------------------------------------------------------------------
// The reference's callers make a decoder per file (DCSEncoder.cpp:522-540) or per run and expect that to cost nothing.  A
// DcsCtx is a HIP stream, the decode tables on the device and the live decoder's pinned arenas: 8 ms to make, as long as the
// reference takes to decode 2 000 frames.  So a decoder that goes away leaves its context here, and the next decoder on that
// device takes it over (a context serves one decoder at a time, like the decoder object itself serves one thread).  At most
// kIdleContexts wait here; they are destroyed when the library is unloaded.
namespace {
struct ContextPool
{
    static const size_t kIdleContexts = 4;
    std::mutex m;
    std::vector<std::pair<int, DcsCtx *>> idle;
    ~ContextPool()
    {
        for (auto &c : idle)
            dcs_ctx_destroy(c.second);
    }
    DcsCtx *take(int device)
    {
        std::lock_guard<std::mutex> lock(m);
        for (size_t i = 0 ; i < idle.size() ; ++i)
            if (idle[i].first == device)
            {
                DcsCtx *c = idle[i].second;
                idle.erase(idle.begin() + static_cast<long>(i));
                return c;
            }
        return nullptr;
    }
    void give(int device, DcsCtx *c)
    {
        {
            std::lock_guard<std::mutex> lock(m);
            if (idle.size() < kIdleContexts)
            {
                idle.emplace_back(device, c);
                return;
            }
        }
        dcs_ctx_destroy(c);
    }
};
ContextPool &contextPool()
{
    static ContextPool pool;
    return pool;
}
}   // namespace

// DCS_CLASS_STATS=1: where a decoder object's time went (loading streams = the index walk, planning ahead, decoding), printed when
// it goes away
static const bool g_classStats = getenv("DCS_CLASS_STATS") != nullptr && atoi(getenv("DCS_CLASS_STATS")) != 0;
static double nowUs() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- DCSDecoderHIP -----------------------------------------------------------------------------------------
DCSDecoderHIP::DCSDecoderHIP(Host *host, int deviceId) : DCSDecoder(host), deviceId(deviceId) { }

void DCSDecoderHIP::ReleaseContext()
{
    if (ctx != nullptr)
        contextPool().give(deviceId, ctx);
    ctx = nullptr;
}

DCSDecoderHIP::~DCSDecoderHIP()
{
    if (g_classStats)
        fprintf(stderr, "DCSDecoderHIP %p: %llu ticks handed out, %u refills; load %.1f us, plan %.1f us, decode %.1f us, going back %.1f us (%u times)\n",
                static_cast<void *>(this), static_cast<unsigned long long>(nextTick), stats.refills, stats.loadUs, stats.planUs, stats.decodeUs, stats.syncUs, stats.syncs);
    if (seq != nullptr) dcs_seq_destroy(seq);
    if (roms != nullptr) dcs_romset_destroy(roms);
    ReleaseContext();
}

void DCSDecoderHIP::InitStandalone(OSVersion v)
{
    osVersion = v;
    // each OS version goes with one board (DCSDecoderNative.cpp:32-59); SoftBoot then does not ask the ROMs
    hwVersion = v == OSVersion::OS95 ? HWVersion::DCS95 : v == OSVersion::Invalid ? HWVersion::Invalid
              : v == OSVersion::Unknown ? HWVersion::Unknown : HWVersion::DCS93;
}

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

// ---- ROMs ----------------------------------------------------------------------------------------------------
bool DCSDecoderHIP::EnsureRoms()
{
    if (roms == nullptr)
        roms = dcs_romset_create();
    return roms != nullptr;
}

#ifndef DCSHIP_USE_REFERENCE_BASE
void DCSDecoderHIP::AddROM(int n, const uint8_t *data, size_t size)
{
    if (EnsureRoms())
        dcs_romset_add_rom(roms, n, data, size);
}

bool DCSDecoderHIP::LoadROMFromZipFile(const char *zipFileName, const char *explicitU2, std::string *errorDetails)
{
    if (!EnsureRoms())
        return false;
    const DcsStatus st = dcs_romset_load_zip(roms, zipFileName, explicitU2);
    if (st != DCS_OK && errorDetails != nullptr)
        *errorDetails = dcs_romset_last_error(roms);
    return st == DCS_OK;
}

uint8_t DCSDecoderHIP::CheckROMs()
{
    hwVersion = HWVersion::Invalid;
    osVersion = OSVersion::Invalid;
    nominalVersion = 0;
    DcsRomCheck c;
    if (roms == nullptr || dcs_romset_check(roms, &c) != DCS_OK)
        return 2;
    hwVersion = c.hw == DCS_HW_DCS93 ? HWVersion::DCS93 : c.hw == DCS_HW_DCS95 ? HWVersion::DCS95 : HWVersion::Invalid;
    osVersion = c.os == DCS_OS93A ? OSVersion::OS93a : c.os == DCS_OS93B ? OSVersion::OS93b : c.os == DCS_OS94 ? OSVersion::OS94
              : c.os == DCS_OS95 ? OSVersion::OS95 : OSVersion::Invalid;
    nominalVersion = c.nominalVersion;
    return static_cast<uint8_t>(c.status);
}

void DCSDecoderHIP::SetVersions(HWVersion hw, OSVersion os)
{
    hwVersion = hw;
    osVersion = os;
    if (EnsureRoms())
        dcs_romset_set_version(roms, hw == HWVersion::DCS95 ? DCS_HW_DCS95 : DCS_HW_DCS93, AbiOs());
}

int DCSDecoderHIP::GetVersionNumber() const
{
    return nominalVersion != 0 ? static_cast<int>(nominalVersion)
         : (osVersion == OSVersion::OS93a || osVersion == OSVersion::OS93b) ? 0x0100
         : osVersion == OSVersion::OS94 ? 0x0101 : 0x0000;
}

uint16_t DCSDecoderHIP::GetMaxTrackNumber() const
{
    return static_cast<uint16_t>((roms != nullptr ? dcs_romset_num_tracks(roms) : 0) - 1);      // (sic) DCSDecoder.h:360
}

bool DCSDecoderHIP::GetTrackInfo(uint16_t trackNumber, TrackInfo &ti)
{
    ti = TrackInfo();
    DcsTrackInfo t;
    if (roms == nullptr || dcs_romset_track_info(roms, trackNumber, &t) != DCS_OK)
        return false;
    ti.address = t.address; ti.channel = t.channel; ti.type = t.type; ti.deferCode = t.deferCode;
    ti.time = t.time; ti.looping = t.looping != 0;
    return true;
}

std::vector<DcsTrackOp> DCSDecoderHIP::DecompileTrackProgram(uint16_t trackNumber)
{
    std::vector<DcsTrackOp> v;
    uint32_t n = 0;
    if (roms != nullptr && dcs_romset_decompile(roms, trackNumber, nullptr, 0, &n) == DCS_OK && n != 0)
    {
        v.resize(n);
        dcs_romset_decompile(roms, trackNumber, v.data(), n, &n);
    }
    return v;
}

std::vector<uint32_t> DCSDecoderHIP::ListStreams()
{
    std::vector<uint32_t> v;
    uint32_t n = 0;
    if (roms != nullptr && dcs_romset_list_streams(roms, nullptr, 0, &n) == DCS_OK && n != 0)
    {
        v.resize(n);
        dcs_romset_list_streams(roms, v.data(), n, &n);
    }
    return v;
}

DCSDecoder::ROMPointer DCSDecoderHIP::MakeROMPointer(uint32_t linearAddress) const
{
    const uint8_t *p = nullptr;
    size_t avail = 0;
    int chip = 2;
    if (roms == nullptr || dcs_romset_pointer(roms, linearAddress, &p, &avail, &chip) != DCS_OK)
        return ROMPointer();
    return ROMPointer(chip - 2, p);
}
#endif

// ---- playing ---------------------------------------------------------------------------------------------------
bool DCSDecoderHIP::Initialize()
{
    if (ctx == nullptr)
        ctx = contextPool().take(deviceId);
    if (ctx == nullptr)
    {
        DcsStatus st = dcs_ctx_create(deviceId, &ctx);
        if (st != DCS_OK)
        {
            errorMessage = std::string("HIP decoder unavailable: ") + dcs_last_error(nullptr);
            return false;                                       // InitializationError: GetNextSample returns silence, IsOK() is false
        }
    }
    if (seq != nullptr)
    {
        dcs_seq_destroy(seq);
        seq = nullptr;
    }
    ready = nullptr; readyCount = 0; hostBytes.clear(); hostNext = 0; handedOut = 0; nextTick = 0;
    curLookahead = kFirstLookahead; fatalTick = ~uint64_t(0); lastCommandTick = 0; lastQuiet = 0;
#ifdef DCSHIP_USE_REFERENCE_BASE
    // The base class owns the ROM images (AddROM / LoadROMFromZipFile put them in ROM[], CheckROMs has identified them
    // or the caller has named the versions): hand them to the C ABI's ROM set, with the versions the base holds.
    if (roms != nullptr)
    {
        dcs_romset_destroy(roms);
        roms = nullptr;
    }
    if (ROM[0].data != nullptr && !ROM[0].isDummy && EnsureRoms())
    {
        for (int i = 0 ; i < 8 ; ++i)
            if (ROM[i].data != nullptr && !ROM[i].isDummy)
                dcs_romset_add_rom(roms, i + 2, ROM[i].data, ROM[i].size);
        DcsRomCheck c;
        dcs_romset_check(roms, &c);                             // (catalog, nominal version)
        if (hwVersion == HWVersion::DCS93 || hwVersion == HWVersion::DCS95)
            dcs_romset_set_version(roms, hwVersion == HWVersion::DCS95 ? DCS_HW_DCS95 : DCS_HW_DCS93, AbiOs());
    }
    autobuffer.Set(outputBuffer, 0x1E0, 1);                     // as DCSDecoderNative.cpp:3203
#endif
    const bool haveRoms = roms != nullptr && dcs_romset_num_tracks(roms) != 0;
    if (haveRoms)
    {
        if (hwVersion == HWVersion::Unknown)                    // SoftBoot does this (DCSDecoder.cpp:1522-1523)
            CheckROMs();
        if (hwVersion == HWVersion::Invalid || osVersion == OSVersion::Invalid || osVersion == OSVersion::Unknown)
        {
            errorMessage = "the ROM images could not be identified (CheckROMs); use SetVersions for images without decoder code";
            return false;
        }
        seq = dcs_seq_create(roms);
    }
    else
        seq = dcs_seq_create_standalone(AbiOs());
    if (seq == nullptr)
    {
        errorMessage = "could not create the track sequencer";
        return false;
    }
    dcs_seq_set_rewindable(seq, 1);
    dcs_seq_set_reported_version(seq, reportedVersion);
    dcs_seq_set_master_volume(seq, defaultVolume);              // DCSDecoderNative.cpp:3206
    if (masterVolume >= 0)
        dcs_seq_set_master_volume(seq, masterVolume);
    return true;
}

// Back to the decoder state after the last frame handed out: drops the frames decoded ahead and what the
// sequencer did for them.
void DCSDecoderHIP::Sync()
{
    // a command: whoever sends one may send another soon -- but a caller that let the decoder run for N frames since its last
    // command (a stream loop, DCSExplorer.cpp:1900-1907) will most likely do so again
    // (several commands between two frames -- the bytes of one data-port message -- count as one)
    if (nextTick != lastCommandTick)
    {
        lastQuiet = nextTick - lastCommandTick;
        lastCommandTick = nextTick;
    }
    curLookahead = lastQuiet <= static_cast<uint64_t>(kFirstLookahead) ? kFirstLookahead
                 : lastQuiet >= static_cast<uint64_t>(kMaxLookahead) ? kMaxLookahead : static_cast<int>(lastQuiet);
    if (seq == nullptr || handedOut == readyCount)
        return;
    const double t0 = g_classStats ? nowUs() : 0.0;
    dcs_seq_rewind(seq, handedOut);
    if (g_classStats) { stats.syncUs += nowUs() - t0; ++stats.syncs; }
    readyCount = handedOut;
    while (hostBytes.size() > hostNext && hostBytes.back().tick >= nextTick)
        hostBytes.pop_back();
}

void DCSDecoderHIP::SetMasterVolume(int vol)
{
    masterVolume = vol;
    if (seq != nullptr)
    {
        Sync();
        dcs_seq_set_master_volume(seq, vol);
    }
}

void DCSDecoderHIP::SetReportedVersionNumber(uint16_t vsn)
{
    reportedVersion = vsn;
    if (seq != nullptr)
    {
        Sync();
        dcs_seq_set_reported_version(seq, vsn);
    }
}

#ifdef DCSHIP_USE_REFERENCE_BASE
void DCSDecoderHIP::IRQ2Handler()
{
    // one byte of what the host wrote (the base class queued it in WriteDataPort, DCSDecoder.cpp:1543-1577)
    const uint8_t data = ReadDataPort();
    if (seq != nullptr)
    {
        Sync();
        dcs_seq_write_data_port(seq, data);
    }
}
#else
void DCSDecoderHIP::IRQ2Handler() { }

void DCSDecoderHIP::WriteDataPort(uint8_t data)
{
    if (state == State::HardBoot)
    {
        SoftBoot();                                             // the first byte only wakes the board (DCSDecoder.cpp:1531-1538)
        return;
    }
    if (seq != nullptr)
    {
        Sync();
        dcs_seq_write_data_port(seq, data);
    }
}
#endif

void DCSDecoderHIP::AddTrackCommand(uint16_t trackNum)
{
    if (seq != nullptr)
    {
        Sync();
        dcs_seq_add_track_command(seq, trackNum);
    }
}

void DCSDecoderHIP::ClearTracks()
{
    if (seq == nullptr)
        return;
    // Behind the last frame handed out no channel has a program or a stream (the end of ExtractToWAV, DCSExplorer.cpp:1716-1718,
    // clears a decoder that has run out twice per stream): nothing to clear, and what was decoded ahead stands.
    if (!dcs_seq_tracks_active_at(seq, handedOut))
        return;
    Sync();
    dcs_seq_clear_tracks(seq);
}

// what lies behind a stream pointer: the rest of the ROM image it points into (the base's, or the ROM set's copy), else 64 MB
size_t DCSDecoderHIP::BytesBehind(const ROMPointer &p) const
{
    if (p.IsNull())
        return 0;
#ifdef DCSHIP_USE_REFERENCE_BASE
    for (int i = 0 ; i < 8 ; ++i)
        if (ROM[i].data != nullptr && p.p >= ROM[i].data && p.p < ROM[i].data + ROM[i].size)
            return static_cast<size_t>(ROM[i].data + ROM[i].size - p.p);
#endif
    const size_t n = roms != nullptr ? dcs_romset_bytes_behind(roms, p.p) : 0;
    return n != 0 ? n : size_t(1) << 26;
}

DCSDecoderHIP::StreamInfo DCSDecoderHIP::GetStreamInfo(const ROMPointer &p) { return GetStreamInfoBounded(p, BytesBehind(p)); }

DCSDecoderHIP::StreamInfo DCSDecoderHIP::GetStreamInfoBounded(const ROMPointer &p, size_t maxLen)
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

void DCSDecoderHIP::LoadAudioStream(int ch, const ROMPointer &p, int mixingLevel) { LoadAudioStreamBounded(ch, p, mixingLevel, BytesBehind(p)); }

void DCSDecoderHIP::LoadAudioStreamBounded(int ch, const ROMPointer &p, int mixingLevel, size_t maxLen)
{
    if (seq == nullptr || ch < 0 || ch >= DCS_MAX_CHANNELS || p.IsNull())       // :1390
        return;
    Sync();
    const double t0 = g_classStats ? nowUs() : 0.0;
    dcs_seq_load_audio_stream_mem(seq, ch, p.p, maxLen, mixingLevel);
    if (g_classStats) stats.loadUs += nowUs() - t0;
}

bool DCSDecoderHIP::IsStreamPlaying(int ch)
{
    // (a question, not a command: callers ask it between every two buffers, EncoderTester.cpp:106 -- answered for the last frame
    // handed out, the look-ahead stays)
    return seq != nullptr && dcs_seq_stream_playing_at(seq, handedOut, ch) != 0;
}

// run the sequencer ahead and decode what it planned in one launch
bool DCSDecoderHIP::Refill()
{
    handedOut = 0;
    readyCount = 0;
    ready = nullptr;
    DcsStatus st;
    const double t0 = g_classStats ? nowUs() : 0.0;
    if (lookahead >= 1)
        st = dcs_seq_plan(seq, static_cast<uint32_t>(lookahead));
    else
    {
        st = dcs_seq_plan_ahead(seq, static_cast<uint32_t>(curLookahead), 2, nullptr);
        // (Sync() takes it back to the start.)  What was planned beyond a command is thrown away, half of the last refill on
        // average: a caller that has been sending commands gets refills that double, one that never has, eightfold ones.
        const int grown = curLookahead * (lastQuiet != 0 ? 2 : 8);
        curLookahead = grown > kMaxLookahead ? kMaxLookahead : grown;
    }
    const double t1 = g_classStats ? nowUs() : 0.0;
    if (st == DCS_OK)
        st = dcs_seq_decode_view(ctx, seq, &ready, &readyCount, nullptr);
    if (g_classStats) { stats.planUs += t1 - t0; stats.decodeUs += nowUs() - t1; ++stats.refills; }
    if (st != DCS_OK || readyCount == 0)
    {
        readyCount = 0;
        return false;
    }
    const uint32_t nb = dcs_seq_host_bytes(seq, nullptr, 0);
    if (nb != 0)
    {
        const size_t at = hostBytes.size();
        hostBytes.resize(at + nb);
        dcs_seq_host_bytes(seq, &hostBytes[at], nb);
    }
    fatalTick = dcs_seq_fatal_tick(seq);
    return true;
}

void DCSDecoderHIP::MainLoop()
{
    if (seq == nullptr)
    {
        state = State::DecoderFatalError;
        errorMessage = "decoder not initialised";
        return;
    }
    if (handedOut == readyCount && !Refill())
    {
        memset(outputBuffer, 0, sizeof(outputBuffer));
        state = State::DecoderFatalError;
        errorMessage = std::string("HIP decode failed: ") + dcs_last_error(ctx);
        return;
    }
    // this tick's bytes go to the host now, its frame to the output buffer
    while (hostNext < hostBytes.size() && hostBytes[hostNext].tick <= nextTick)
        host->ReceiveDataPort(static_cast<uint8_t>(hostBytes[hostNext++].byte));
    if (hostNext == hostBytes.size() && hostNext != 0)
    {
        hostBytes.clear();
        hostNext = 0;
    }
    if (fatalTick <= nextTick)
    {
        // the reference gives up after four failed passes in a row (DCSDecoder.cpp:1637-1645) and answers THIS call with
        // silence (:1661); the base class, which sees this MainLoop return normally, hands out the buffer's first sample
        memset(outputBuffer, 0, sizeof(outputBuffer));
        state = State::DecoderFatalError;
        // (the reference's text, DCSDecoder.cpp:1657-1659: callers print it)
        errorMessage = "The decoder performed a self-reset after encountering "
                       "multiple fatal errors decoding track data.  This usually indicates "
                       "that the ROM image is invalid or corrupted.";
        return;
    }
    memcpy(outputBuffer, ready + static_cast<size_t>(handedOut) * DCS_FRAME_SAMPLES, sizeof(int16_t) * DCS_FRAME_SAMPLES);
    ++handedOut;
    ++nextTick;
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

DCSHIP_NAMESPACE_END

// TEST INFRASTRUCTURE ONLY -- never linked into the shipped library.
//
// Thin C-ABI driver around the *unmodified* reference decoder
// (/root/reference/DCSDecoder/DCSDecoder.cpp + DCSDecoderNative.cpp), compiled
// from where the sources lie by oracle/Makefile into oracle/_ref/libdcsref.so.
// This file is ours; it only #includes the reference headers.  It exists to
// (1) pin oracle/dcs_oracle.c against the real reference and (2) generate the
// golden vectors under tests/golden/ (tests/golden/make_golden.py).
//
// The drive recipe is the reference's own ROM-less one (DCSEncoder.cpp:522-571,
// EncoderTester.cpp:85-137): MinHost + DCSDecoderNative + InitStandalone(os) +
// SetDefaultVolume(v) + SoftBoot() + LoadAudioStream(ch, ROMPointer(0,bytes),
// level) + 240 x GetNextSample() per frame.
//
// White-box probes (bit cursor, bandTypeBuf, mixing multiplier) read protected
// members; the standard headers are included first so that the access hack
// below only touches the reference's own class declarations.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <list>
#include <map>
#include <string>
#include <vector>
#include <unordered_map>
#include <unordered_set>
#include <functional>
#include <type_traits>
#include <regex>
#include <set>
#include <algorithm>
#include <thread>

#define protected public
#define private public
#include "DCSDecoderNative.h"
#undef protected
#undef private

extern "C" {

// per-frame probe record, captured immediately BEFORE the MainLoop() that
// produces the frame
struct RefProbe
{
    int32_t active;          // stream still playing on channel 0?
    int32_t bitOff;          // bits consumed from the start of the stream payload (after header)
    uint16_t mixMul;         // channel[0].mixingMultiplier before MainLoop rescales it
    uint16_t volMult;        // master volumeMultiplier
    uint16_t bandType[16];   // carried band type codes before the frame
};

static DCSDecoder::OSVersion OsFromInt(int os)
{
    switch (os)
    {
    case 0: return DCSDecoder::OSVersion::OS93a;
    case 1: return DCSDecoder::OSVersion::OS93b;
    case 2: return DCSDecoder::OSVersion::OS94;
    default: return DCSDecoder::OSVersion::OS95;
    }
}

// Decode nch streams loaded on channels 0..nch-1 at tick 0, pulling
// nFramesOut frames (240 samples each).  streams[i]/lens[i] are copied into
// padded buffers (the reference bit reader looks ahead up to 4 bytes).
// probes (optional) receives nFramesOut records for channel 0.
int ref_decode(int os, int volume, int nch,
    const uint8_t *const *streams, const size_t *lens, const int *levels,
    int nFramesOut, int16_t *pcm, RefProbe *probes)
{
    if (nch < 1 || nch > 8)
        return -1;

    std::vector<std::vector<uint8_t>> bufs(nch);
    for (int i = 0 ; i < nch ; ++i)
    {
        bufs[i].assign(streams[i], streams[i] + lens[i]);
        bufs[i].resize(lens[i] + 64, 0);
    }

    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(OsFromInt(os));
    dec.SetDefaultVolume(volume);
    dec.SoftBoot();
    for (int i = 0 ; i < nch ; ++i)
        dec.LoadAudioStream(i, DCSDecoder::ROMPointer(0, bufs[i].data()), levels[i]);

    for (int f = 0 ; f < nFramesOut ; ++f)
    {
        if (probes != nullptr)
        {
            auto &ch = dec.channel[0];
            auto &s = ch.audioStream;
            RefProbe &pr = probes[f];
            pr.active = s.playbackBitPtr.IsNull() ? 0 : 1;
            pr.bitOff = pr.active
                ? static_cast<int32_t>((s.playbackBitPtr.p.p - s.startPtr.p) * 8 - s.playbackBitPtr.nBits)
                : -1;
            pr.mixMul = ch.mixingMultiplier;
            pr.volMult = dec.volumeMultiplier;
            // note: at a stream (re)start the reference zeroes bandTypeBuf inside
            // DecodeStream, after this probe; report what the frame will see
            bool atStart = pr.active && (s.playbackBitPtr == s.startPtr);
            for (int i = 0 ; i < 16 ; ++i)
                pr.bandType[i] = atStart ? 0 : s.bandTypeBuf[i];
        }
        for (int i = 0 ; i < 240 ; ++i)
            *pcm++ = dec.GetNextSample();
    }
    return dec.IsOK() ? 0 : -2;
}

// DCSDecoderNative::GetStreamInfo (DCSDecoderNative.cpp:1486-1537)
int ref_stream_info(int os, const uint8_t *stream, size_t len,
    int *nFrames, int *nBytes, int *formatType, int *formatSubType, uint8_t *header16)
{
    std::vector<uint8_t> buf(stream, stream + len);
    buf.resize(len + 64, 0);

    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(OsFromInt(os));
    dec.SoftBoot();
    auto info = dec.GetStreamInfo(DCSDecoder::ROMPointer(0, buf.data()));
    *nFrames = info.nFrames;
    *nBytes = info.nBytes;
    *formatType = info.formatType;
    *formatSubType = info.formatSubType;
    memcpy(header16, info.header, 16);
    return 0;
}

// White-box frame-level entry: run DecompressFrame for one frame of one stream
// into a zeroed 512-word frame buffer and return the frequency-domain words,
// then TransformFrame(volShift) with a caller-given overlap tail.  Used to pin
// the transform stages separately from the unpackers.
int ref_transform(int os, uint16_t *frameBuf512, int volShift,
    uint16_t *overlap16, int16_t *pcm240)
{
    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(OsFromInt(os));
    dec.SoftBoot();
    memcpy(dec.frameBuffer, frameBuf512, sizeof(dec.frameBuffer));
    memcpy(dec.overlapBuffer, overlap16, sizeof(dec.overlapBuffer));
    dec.decoderImpl->TransformFrame(volShift);
    memcpy(pcm240, dec.outputBuffer, 240 * sizeof(int16_t));
    memcpy(overlap16, dec.overlapBuffer, sizeof(dec.overlapBuffer));
    memcpy(frameBuf512, dec.frameBuffer, sizeof(dec.frameBuffer));
    return 0;
}

// White-box: decompress `nFrames` consecutive frames of one stream with a given
// mixing multiplier, each into a fresh zeroed frame buffer (no transform).
// out = nFrames x 512 words.  Also returns the per-frame bit offsets.
int ref_decompress(int os, const uint8_t *stream, size_t len, uint16_t mixMul,
    int nFrames, uint16_t *out, int32_t *bitOffs, uint16_t *bandTypes /* nFrames x 16, after frame */,
    int32_t *stopFlags)
{
    std::vector<uint8_t> buf(stream, stream + len);
    buf.resize(len + 64, 0);

    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(OsFromInt(os));
    dec.SoftBoot();

    auto &ch = dec.channel[0];
    dec.InitChannelStream(ch, DCSDecoder::ROMPointer(0, buf.data()));
    dec.InitStreamPlayback(ch);
    for (int f = 0 ; f < nFrames ; ++f)
    {
        auto &s = ch.audioStream;
        bitOffs[f] = static_cast<int32_t>((s.playbackBitPtr.p.p - s.startPtr.p) * 8 - s.playbackBitPtr.nBits);
        ch.mixingMultiplier = mixMul;
        ch.stop = false;
        uint16_t fb[0x200];
        memset(fb, 0, sizeof(fb));
        dec.decoderImpl->DecompressFrame(ch, fb);
        memcpy(out + f * 0x200, fb, sizeof(fb));
        memcpy(bandTypes + f * 16, s.bandTypeBuf, 16 * sizeof(uint16_t));
        stopFlags[f] = ch.stop ? 1 : 0;
    }
    return 0;
}

// Volume / mixing parameter probes (DCSDecoderNative.cpp:3250-3282, :3042-3121)
uint16_t ref_volume_multiplier(int volume)
{
    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(DCSDecoder::OSVersion::OS94);
    dec.SetDefaultVolume(volume);
    dec.SoftBoot();
    return dec.volumeMultiplier;
}

uint16_t ref_mixing_multiplier(int os, int levelSum, int channelVolume)
{
    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(OsFromInt(os));
    dec.SoftBoot();
    dec.channel[0].channelVolume = static_cast<uint16_t>(channelVolume);
    dec.channel[0].mixer[0].curLevel = levelSum;
    dec.UpdateMixingLevels();
    return dec.channel[0].mixingMultiplier;
}

// CPU-baseline helper: decode n single-channel streams `repeat` times on `nThreads` host threads
// (streams range-partitioned over the threads).  Each thread owns ONE decoder object per OS version and
// plays its streams through it one after the other -- LoadAudioStream(0, ptr, level), then
// nFrames x 240 GetNextSample() -- which is how the reference's own batch decode drives the path
// (DCSExplorer --extract-streams, DCSExplorer.cpp:1628-1907).  Timing only: the PCM is discarded.
// Returns the number of frames decoded.
long long ref_decode_many(const int *os, const int *volume, const int *level,
    const uint8_t *const *streams, const size_t *lens, int n, int repeat, int nThreads)
{
    // padded copies: the reference bit reader looks a few bytes ahead
    std::vector<std::vector<uint8_t>> bufs(n);
    for (int i = 0 ; i < n ; ++i)
    {
        bufs[i].assign(streams[i], streams[i] + lens[i]);
        bufs[i].resize(lens[i] + 64, 0);
    }
    std::vector<long long> frames(nThreads, 0);
    std::vector<std::thread> pool;
    for (int t = 0 ; t < nThreads ; ++t)
        pool.emplace_back([&, t]() {
            DCSDecoder::MinHost host;
            std::unique_ptr<DCSDecoderNative> dec[4];
            volatile int16_t sink = 0;
            for (int rep = 0 ; rep < repeat ; ++rep)
                for (int i = t ; i < n ; i += nThreads)
                {
                    auto &d = dec[os[i] & 3];
                    if (!d)
                    {
                        d.reset(new DCSDecoderNative(&host));
                        d->InitStandalone(OsFromInt(os[i]));
                        d->SetDefaultVolume(volume[i]);
                        d->SoftBoot();
                    }
                    d->SetMasterVolume(volume[i]);
                    const int nf = (bufs[i][0] << 8) | bufs[i][1];
                    d->LoadAudioStream(0, DCSDecoder::ROMPointer(0, bufs[i].data()), level[i]);
                    int16_t acc = 0;
                    for (int k = 0 ; k < nf * 240 ; ++k)
                        acc ^= d->GetNextSample();
                    sink = acc;
                    frames[t] += nf;
                }
            (void)sink;
        });
    for (auto &th : pool) th.join();
    long long total = 0;
    for (long long f : frames) total += f;
    return total;
}

// The stream loop of DCSExplorer --extract-streams (DCSExplorer.cpp:1628-1907) on ONE decoder object: for every
// stream LoadAudioStream(0, ptr, level), then nFrames + extraFrames frames of 240 GetNextSample() with
// ClearTracks() after each of the last two (ExtractToWAV, :1670-1721).  Decoder state (mixing multiplier,
// overlap tail) carries from one stream to the next exactly as it does there.  pcm receives the frames of
// all streams back to back.
int ref_decode_sequence(int os, int volume, int n, const uint8_t *const *streams, const size_t *lens,
    const int *levels, int extraFrames, int16_t *pcm)
{
    std::vector<std::vector<uint8_t>> bufs(n);
    for (int i = 0 ; i < n ; ++i)
    {
        bufs[i].assign(streams[i], streams[i] + lens[i]);
        bufs[i].resize(lens[i] + 64, 0);
    }
    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    dec.InitStandalone(OsFromInt(os));
    dec.SetDefaultVolume(volume);
    dec.SoftBoot();
    dec.SetMasterVolume(volume);
    for (int i = 0 ; i < n ; ++i)
    {
        dec.LoadAudioStream(0, DCSDecoder::ROMPointer(0, bufs[i].data()), levels[i]);
        const int nFrames = ((bufs[i][0] << 8) | bufs[i][1]) + extraFrames;
        for (int frame = 0 ; frame < nFrames ; ++frame)
        {
            for (int k = 0 ; k < 240 ; ++k)
                *pcm++ = dec.GetNextSample();
            if (frame + 2 >= nFrames)
                dec.ClearTracks();
        }
    }
    return 0;
}

// ROM ingestion: everything the reference derives from a set of ROM images, as text, one item
// per line -- CheckROMs, the catalog, every track's TrackInfo and decompiled program, ListStreams with
// MakeROMPointer, and the stream list of the --extract-streams loop (the loop itself lives in the Windows
// program DCSExplorer.cpp:1742-1810; its 20 lines of level tracking are restated here on top of the
// reference's DecompileTrackProgram).  forceHw/forceOs >= 0 override what CheckROMs detected (synthetic
// images carry no ADSP code to detect the OS version from).  Returns the text length.
size_t ref_rom_dump(const uint8_t *const *roms, const size_t *sizes, int forceHw, int forceOs, char *out, size_t cap)
{
    DCSDecoder::MinHost host;
    DCSDecoderNative dec(&host);
    for (int i = 0 ; i < 8 ; ++i)
        if (roms[i] != nullptr && sizes[i] != 0)
            dec.AddROM(i + 2, roms[i], sizes[i]);
    std::string t;
    char line[512];
    const int status = dec.CheckROMs();
    auto hwNum = [](DCSDecoder::HWVersion h) { return h == DCSDecoder::HWVersion::DCS93 ? 2 : h == DCSDecoder::HWVersion::DCS95 ? 3 : h == DCSDecoder::HWVersion::Invalid ? 1 : 0; };
    auto osNum = [](DCSDecoder::OSVersion o) { return o == DCSDecoder::OSVersion::OS93a ? 0 : o == DCSDecoder::OSVersion::OS93b ? 1 : o == DCSDecoder::OSVersion::OS94 ? 2 : o == DCSDecoder::OSVersion::OS95 ? 3 : -1; };
    snprintf(line, sizeof(line), "check status=%d hw=%d os=%d nominal=%04x catalog=%x ntracks=%u sig=%s\n", status,
             hwNum(dec.hwVersion), osNum(dec.osVersion), dec.nominalVersion, dec.GetCatalogOffset(),
             static_cast<unsigned>(dec.catalog.nTracks), dec.GetSignature().c_str());
    t += line;
    if (forceHw >= 0)
        dec.hwVersion = forceHw == 3 ? DCSDecoder::HWVersion::DCS95 : DCSDecoder::HWVersion::DCS93;
    if (forceOs >= 0)
        dec.osVersion = OsFromInt(forceOs);
    std::unordered_set<uint32_t> seen;
    std::string plan;
    for (unsigned trackNum = 0 ; trackNum < dec.catalog.nTracks ; ++trackNum)
    {
        DCSDecoder::TrackInfo ti;
        if (!dec.GetTrackInfo(static_cast<uint16_t>(trackNum), ti))
            continue;
        snprintf(line, sizeof(line), "track %u addr=%06x ch=%d type=%d defer=%04x time=%u loop=%d\n", trackNum, ti.address,
                 ti.channel, ti.type, ti.deferCode & 0xFFFF, ti.time, ti.looping ? 1 : 0);
        t += line;
        if (ti.type != 1)
            continue;
        int mixerLevel[8] = { 0x64, 0x64, 0x64, 0x64, 0x64, 0x64, 0x64, 0x64 };
        int streamNum = 0;
        for (auto &op : dec.DecompileTrackProgram(static_cast<uint16_t>(trackNum)))
        {
            snprintf(line, sizeof(line), " op off=%d nest=%d parent=%d delay=%04x opc=%02x n=%d bytes=", op.offset, op.nestingLevel,
                     op.loopParent, op.delayCount, op.opcode, op.nOperandBytes);
            t += line;
            for (int i = 0 ; i < op.nOperandBytes && i < 8 ; ++i)
            {
                snprintf(line, sizeof(line), "%02x", op.operandBytes[i]);
                t += line;
            }
            t += "\n";
            const int ch = op.operandBytes[0] & 7;
            if (op.opcode == 0x07 || op.opcode == 0x0A) mixerLevel[ch] = op.operandBytes[1];
            else if (op.opcode == 0x08 || op.opcode == 0x0B) mixerLevel[ch] += op.operandBytes[1];
            else if (op.opcode == 0x09 || op.opcode == 0x0A) mixerLevel[ch] -= op.operandBytes[1];
            else if (op.opcode == 0x01)
            {
                const uint32_t addr = (op.operandBytes[1] << 16) | (op.operandBytes[2] << 8) | op.operandBytes[3];
                if (seen.insert(addr).second)
                {
                    snprintf(line, sizeof(line), "extract track=%u num=%d addr=%06x level=%d\n", trackNum, ++streamNum, addr, mixerLevel[ch]);
                    plan += line;
                }
            }
        }
    }
    for (uint32_t addr : dec.ListStreams())
    {
        auto rp = dec.MakeROMPointer(addr);
        const auto &rom = dec.ROM[rp.chipSelect];
        snprintf(line, sizeof(line), "stream %06x chip=%d off=%zx\n", addr, rp.NominalChipNumber(), static_cast<size_t>(rp.p - rom.data));
        t += line;
    }
    t += plan;
    if (out != nullptr && cap != 0)
    {
        const size_t n = t.size() < cap - 1 ? t.size() : cap - 1;
        memcpy(out, t.data(), n);
        out[n] = 0;
    }
    return t.size();
}

// Track-program sequencer: the real decoder with ROMs, driven tick by tick.  events = triples
// (tick, kind, value): kind 0 WriteDataPort(value), 1 AddTrackCommand(value), 2 SetMasterVolume(value),
// 3 ClearTracks(); applied before the tick's first sample.  pcm = nTicks x 240; host bytes the decoder sent
// are returned as (tick, byte) pairs.  forceNominal >= 0 overrides the detected nominal version.
struct RefCapHost : public DCSDecoder::Host
{
    std::vector<uint32_t> *log; uint32_t *tick;
    void ReceiveDataPort(uint8_t data) override { log->push_back(*tick); log->push_back(data); }
    void ClearDataPort() override { }
    void BootTimerControl(bool) override { }
};

int ref_seq_run(const uint8_t *const *roms, const size_t *sizes, int forceHw, int forceOs, int forceNominal, int volume,
    const uint32_t *events, int nEvents, int nTicks, int16_t *pcm, uint32_t *hostBytes, int hostCap, int *nHost, int *fatal)
{
    std::vector<uint32_t> log;
    uint32_t tick = 0;
    RefCapHost host;
    host.log = &log; host.tick = &tick;
    DCSDecoderNative dec(&host);
    for (int i = 0 ; i < 8 ; ++i)
        if (roms[i] != nullptr && sizes[i] != 0)
            dec.AddROM(i + 2, roms[i], sizes[i]);
    dec.CheckROMs();
    if (forceHw >= 0) dec.hwVersion = forceHw == 3 ? DCSDecoder::HWVersion::DCS95 : DCSDecoder::HWVersion::DCS93;
    if (forceOs >= 0) dec.osVersion = OsFromInt(forceOs);
    if (forceNominal >= 0) dec.nominalVersion = static_cast<uint16_t>(forceNominal);
    dec.SetDefaultVolume(volume);
    dec.SoftBoot();
    dec.SetMasterVolume(volume);
    int e = 0;
    for (tick = 0 ; tick < static_cast<uint32_t>(nTicks) ; ++tick)
    {
        for ( ; e < nEvents && events[3 * e] <= tick ; ++e)
        {
            const uint32_t kind = events[3 * e + 1], value = events[3 * e + 2];
            if (kind == 0) dec.WriteDataPort(static_cast<uint8_t>(value));
            else if (kind == 1) dec.AddTrackCommand(static_cast<uint16_t>(value));
            else if (kind == 2) dec.SetMasterVolume(static_cast<int>(value));
            else if (kind == 3) dec.ClearTracks();
        }
        for (int k = 0 ; k < 240 ; ++k)
            *pcm++ = dec.GetNextSample();
    }
    const int n = static_cast<int>(log.size() / 2);
    if (nHost != nullptr) *nHost = n;
    for (int i = 0 ; i < n && i < hostCap ; ++i)
    {
        hostBytes[2 * i] = log[2 * i];
        hostBytes[2 * i + 1] = log[2 * i + 1];
    }
    if (fatal != nullptr) *fatal = dec.IsOK() ? 0 : 1;
    return 0;
}

}   // extern "C"

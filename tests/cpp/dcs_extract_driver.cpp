// dcs_extract_driver -- the caller's side of `DCSExplorer --extract-streams[=raw] / --extract-tracks / --streams`
// (DCSExplorer.cpp:1628-1939, :696-770), written once as a template over the decoder class and built twice by
// oracle/Makefile (target extract; build container only, the binaries travel to the GPU box under oracle/_ref/):
//
//   dcs_extract_native   Decoder = the reference's unmodified DCSDecoderNative (linked with its DCSDecoder.cpp and
//                        DCSDecoderNative.cpp): what the reference writes -- the expected files;
//   dcs_extract_hip      Decoder = DCSDecoderHIP compiled with -DDCSHIP_USE_REFERENCE_BASE against the reference's REAL
//                        base class; the object comes out of the real registration map as `--decoder=hip` would get it.
//
// Both make the member calls the reference's loops make, in their order: GetVersionInfo, SoftBoot, SetMasterVolume(255),
// GetMaxTrackNumber, GetTrackInfo, DecompileTrackProgram, MakeROMPointer, GetStreamInfo, LoadAudioStream(0, ptr, level),
// ROMPointer::GetU16, GetNextSample x 240 per frame, ClearTracks behind each of the last two frames, AddTrackCommand,
// ROMPointerOffset, NominalChipNumber.  tests/test_extract.py holds every file the two write against each other, byte
// for byte.  This is test infrastructure (our code, the shape of the caller); nothing of it is part of libdcs_hip.so.
//
//   dcs_extract_<which> <mode> <outPrefix> <chip>=<romfile> ...
//     mode: wav | raw | tracks | list;  files go to <outPrefix>_..., what the reference prints to <outPrefix>.log
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>
#include "DCSDecoder.h"
#ifdef DCS_EXTRACT_HIP
#include "DCSDecoderHIP.h"
typedef DCSDecoderHIP DecoderUnderTest;
#else
#include "DCSDecoderNative.h"
typedef DCSDecoderNative DecoderUnderTest;
#endif

static FILE *g_log;
static void say(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(g_log, fmt, ap);
    va_end(ap);
}

static uint32_t be24(const uint8_t *p) { return (uint32_t(p[0]) << 16) | (uint32_t(p[1]) << 8) | p[2]; }
static void le16(uint8_t *p, uint32_t v) { p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); }
static void le32(uint8_t *p, uint32_t v) { le16(p, v); le16(p + 2, v >> 16); }

struct Tally { int ok = 0, failed = 0; };

// nFrames + 2 frames of whatever the decoder is playing into a WAV file, playback cancelled behind each of the last two
// frames (ExtractToWAV, DCSExplorer.cpp:1665-1735; the frame count is a uint16_t there)
template <class Decoder>
static void framesToWav(Decoder *decoder, const std::string &path, const char *what, uint16_t nFrames, Tally &tally)
{
    nFrames = uint16_t(nFrames + 2);
    FILE *fp = fopen(path.c_str(), "wb");
    if (fp == nullptr)
    {
        say("Unable to open extraction output file \"%s\"\n", path.c_str());
        ++tally.failed;
        return;
    }
    uint8_t h[44] = { 0 };
    memcpy(h, "RIFF", 4); memcpy(h + 8, "WAVEfmt ", 8); memcpy(h + 36, "data", 4);
    le32(h + 4, uint32_t(nFrames) * 480 + 36); le32(h + 16, 16); le16(h + 20, 1); le16(h + 22, 1);
    le32(h + 24, 31250); le32(h + 28, 62500); le16(h + 32, 2); le16(h + 34, 16); le32(h + 40, uint32_t(nFrames) * 480);
    bool good = fwrite(h, sizeof(h), 1, fp) == 1;
    for (uint16_t f = 0 ; f < nFrames ; ++f)
    {
        int16_t pcm[240];
        for (int16_t &s : pcm)
            s = decoder->GetNextSample();
        good = fwrite(pcm, sizeof(pcm), 1, fp) == 1 && good;
        if (f + 2 >= nFrames)
            decoder->ClearTracks();
    }
    good = fclose(fp) == 0 && good;
    if (good) { say("OK %s\n", what); ++tally.ok; }
    else { say("Error extracting %s\n", what); ++tally.failed; }
}

// the stream's bytes in the "DCSa" container (DCSExplorer.cpp:1825-1889)
template <class Decoder>
static void streamToRaw(Decoder *decoder, DCSDecoder::OSVersion os, const DCSDecoder::ROMPointer &ptr, const std::string &path, const char *what, Tally &tally)
{
    FILE *fp = fopen(path.c_str(), "wb");
    if (fp == nullptr)
    {
        say("Unable to open extraction output file \"%s\"\n", path.c_str());
        ++tally.failed;
        return;
    }
    const bool is93 = os == DCSDecoder::OSVersion::OS93a || os == DCSDecoder::OSVersion::OS93b;
    uint8_t h[36] = { 0 };
    memcpy(h, "DCSa", 4);
    h[4] = is93 ? 0x93 : 0x94;
    h[5] = os == DCSDecoder::OSVersion::OS93a ? 1 : os == DCSDecoder::OSVersion::OS93b ? 2 : 0;
    h[7] = 1; h[8] = 0x7A; h[9] = 0x12;
    const auto info = decoder->GetStreamInfo(ptr);
    for (int i = 0 ; i < 4 ; ++i)
        h[32 + i] = uint8_t(uint32_t(info.nBytes) >> (24 - 8 * i));
    if (fwrite(h, 1, sizeof(h), fp) != sizeof(h) || fwrite(ptr.p, 1, size_t(info.nBytes), fp) != size_t(info.nBytes))
    {
        say("Error writing extraction output file \"%s\"\n", path.c_str());
        ++tally.failed;
    }
    fclose(fp);
    say("OK %s\n", what);
    ++tally.ok;
}

// ExtractTracksOrStreams (DCSExplorer.cpp:1628-1939)
template <class Decoder>
static void extract(Decoder *decoder, bool streamsMode, bool raw, const std::string &prefix)
{
    DCSDecoder::OSVersion os;
    DCSDecoder::HWVersion hw;
    decoder->GetVersionInfo(&hw, &os);
    const char *unit = streamsMode ? "stream" : "track";
    say("\n*** Extracting %ss ***\n", unit);
    decoder->SoftBoot();
    decoder->SetMasterVolume(255);

    std::unordered_set<uint32_t> seen;
    Tally tally;
    char name[512], what[128];
    for (uint16_t track = 0 ; track <= decoder->GetMaxTrackNumber() ; ++track)
    {
        DCSDecoder::TrackInfo ti;
        if (!decoder->GetTrackInfo(track, ti) || ti.type != 1)
            continue;
        int level[8] = { 0x64, 0x64, 0x64, 0x64, 0x64, 0x64, 0x64, 0x64 };      // per channel, as the program sets them
        int nStreams = 0;
        for (auto &op : decoder->DecompileTrackProgram(track))
        {
            switch (op.opcode)
            {
            case 0x07: case 0x0A: level[op.operandBytes[0] & 7] = op.operandBytes[1]; break;
            case 0x08: case 0x0B: level[op.operandBytes[0] & 7] += op.operandBytes[1]; break;
            case 0x09:            level[op.operandBytes[0] & 7] -= op.operandBytes[1]; break;   // (0x0C is not followed there, :1778)
            case 0x01:
            {
                const int channel = op.operandBytes[0];
                const uint32_t addr = be24(&op.operandBytes[1]);
                if (!streamsMode)
                {
                    ++nStreams;
                    break;
                }
                if (!seen.insert(addr).second)
                    break;
                ++nStreams;
                auto ptr = decoder->MakeROMPointer(addr);
                snprintf(what, sizeof(what), "track %04x, stream #%d, address $%06x", track, nStreams, addr);
                snprintf(name, sizeof(name), "%s_%04X_%02X_%06X.%s", prefix.c_str(), track, nStreams, addr, raw ? "dcs" : "wav");
                if (raw)
                    streamToRaw(decoder, os, ptr, name, what, tally);
                else
                {
                    decoder->LoadAudioStream(0, ptr, level[channel & 7]);
                    framesToWav(decoder, name, what, ptr.GetU16(), tally);
                }
                break;
            }
            default: break;
            }
        }
        if (!streamsMode && nStreams != 0)
        {
            decoder->ClearTracks();
            decoder->AddTrackCommand(track);
            snprintf(name, sizeof(name), "%s_%04x.wav", prefix.c_str(), track);
            snprintf(what, sizeof(what), "track %04x", track);
            framesToWav(decoder, name, what, uint16_t(ti.time), tally);
        }
    }
    say("\n*** Extraction summary ***\n%-24s%d\nSuccessfully extracted: %d\nErrors:                 %d\n",
        streamsMode ? "Streams found:" : "Tracks found", tally.ok + tally.failed, tally.ok, tally.failed);
}

// the `--streams` table (DCSExplorer.cpp:696-770): every stream some track program plays, by address
template <class Decoder>
static void listStreams(Decoder *decoder)
{
    DCSDecoder::OSVersion os;
    DCSDecoder::HWVersion hw;
    decoder->GetVersionInfo(&hw, &os);
    decoder->SoftBoot();
    std::set<uint32_t> addrs;
    const uint16_t maxTrack = decoder->GetMaxTrackNumber();
    for (uint16_t t = 0 ; t <= maxTrack ; ++t)
        for (auto &op : decoder->DecompileTrackProgram(t))
            if (op.opcode == 0x01)
                addrs.insert(be24(&op.operandBytes[1]));
    say("\nAddress             Fmt Stream Header                                    Time (sec)   Bytes Compressed  Uncompressed  Ratio\n");
    for (uint32_t addr : addrs)
    {
        auto ptr = decoder->MakeROMPointer(addr);
        const auto info = decoder->GetStreamInfo(ptr);
        char fmt[16];
        if (os == DCSDecoder::OSVersion::OS93a || os == DCSDecoder::OSVersion::OS93b)
            snprintf(fmt, sizeof(fmt), "%d", info.formatType);
        else
            snprintf(fmt, sizeof(fmt), "%d.%d", info.formatType, info.formatSubType);
        const float seconds = float(info.nFrames) * 0.00768f;
        const int pcmBytes = info.nFrames * 480;
        const float ratio = float(pcmBytes) / float(info.nBytes);
        say("%07lx [U%d %05x]  %-3s", static_cast<unsigned long>(addr), ptr.NominalChipNumber(), static_cast<unsigned>(decoder->ROMPointerOffset(ptr)), fmt);
        for (int i = 0 ; i < 16 ; ++i)
            say(" %02x", info.header[i]);
        say(" %9.2f            %6u      %8u     %.1f:1 (%.1f%%)\n", seconds, static_cast<unsigned>(info.nBytes), static_cast<unsigned>(pcmBytes),
            ratio, (1.0f - 1.0f / ratio) * 100.0f);
    }
}

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (f == nullptr) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    uint8_t buf[65536];
    for (size_t n ; (n = fread(buf, 1, sizeof(buf), f)) > 0 ; )
        v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s wav|raw|tracks|list <outPrefix> <chip>=<romfile> ...\n", argv[0]); return 2; }
    const std::string mode = argv[1], prefix = argv[2];
    g_log = fopen((prefix + ".log").c_str(), "w");
    if (g_log == nullptr) { fprintf(stderr, "cannot write %s.log\n", prefix.c_str()); return 2; }

    DCSDecoder::MinHost host;
#ifdef DCS_EXTRACT_HIP
    // what a maintainer's patch at DCSExplorer.cpp:1636 and :712 amounts to: the decoder `--decoder=hip` made is asked
    // for DCSDecoderHIP instead of DCSDecoderNative, and the (now templated) loops run on it
    auto &registry = DCSDecoder::GetRegistrationMap();
    auto it = registry.find("hip");
    if (it == registry.end()) { fprintf(stderr, "decoder 'hip' is not registered\n"); return 3; }
    DCSDecoder *base = it->second.factory(&host);
    DecoderUnderTest *decoder = dynamic_cast<DecoderUnderTest *>(base);
#else
    DCSDecoder *base = new DCSDecoderNative(&host);
    DecoderUnderTest *decoder = dynamic_cast<DecoderUnderTest *>(base);
#endif
    if (decoder == nullptr) { fprintf(stderr, "not the decoder class this driver was built for\n"); return 3; }

    std::vector<std::vector<uint8_t>> roms;
    for (int i = 3 ; i < argc ; ++i)
    {
        const char *eq = strchr(argv[i], '=');
        if (eq == nullptr) { fprintf(stderr, "bad ROM argument %s\n", argv[i]); return 2; }
        roms.push_back(slurp(eq + 1));
        base->AddROM(atoi(argv[i]), roms.back().data(), roms.back().size());
    }
    const int post = base->CheckROMs();
    if (post != 1) { fprintf(stderr, "CheckROMs: %d\n", post); return 4; }

    if (mode == "list") listStreams(decoder);
    else if (mode == "wav") extract(decoder, true, false, prefix);
    else if (mode == "raw") extract(decoder, true, true, prefix);
    else if (mode == "tracks") extract(decoder, false, false, prefix);
    else { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2; }
    // (a track program may end the decoder in DecoderFatalError: part of what the two builds must agree on)
    say("decoder %s\n", base->IsOK() ? "ok" : "in error state");
    fclose(g_log);
    delete base;
    return 0;
}

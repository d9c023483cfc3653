// DCSDecoderHIP.h -- host-side C++ class surface of the MI355X decoder.
//
// Mirrors the part of the reference's class surface that sits on the frame-decode hot path
// (DCSDecoder/DCSDecoder.h and DCSDecoder/DCSDecoderNative.h in mjrgh/DCSExplorer), with the same
// names, argument meaning and error behaviour, so that the callers of the reference --
// DCSExplorer's stream extraction (DCSExplorer.cpp:1628-1907), DCSEncoderTester
// (EncoderTester.cpp:85-137), DCSEncoder::EncodeDCSFile (DCSEncoder.cpp:522-571) -- can switch decoder
// with a one-line change.  Everything below the class is the C ABI of dcs_hip.h; the
// frame decode itself runs in HIP kernels, there is no CPU decode path in this class.
//
// Two builds of this header:
//   * stand-alone (default): the client-facing part of the base class is mirrored in namespace dcship, so that the
//     header is self-contained and the library builds without the reference tree;
//   * -DDCSHIP_USE_REFERENCE_BASE, inside the reference tree: DCSDecoderHIP derives from the REAL ::DCSDecoder
//     (DCSDecoder/DCSDecoder.h).  Everything the base already does -- AddROM, LoadROMFromZipFile, CheckROMs,
//     GetTrackInfo, DecompileTrackProgram, ListStreams, MakeROMPointer, WriteDataPort, the boot states, the sample
//     pump -- is inherited, not re-declared, so a caller that holds a plain DCSDecoder* (DCSExplorer.cpp:457-488)
//     reaches this decoder through the base's own members: the ROM images the base collected in ROM[] are handed to
//     the C ABI's ROM set in Initialize(), and the data-port bytes the base queues reach the sequencer through
//     IRQ2Handler() -> ReadDataPort().  oracle/Makefile (target refbase) compiles exactly that against the
//     unmodified base class, and tests/test_refbase.py drives it through a DCSDecoder* only.  In that build the class
//     is ::DCSDecoderHIP, in the stand-alone build dcship::DCSDecoderHIP.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "dcs_hip.h"

#ifdef DCSHIP_USE_REFERENCE_BASE
#include "DCSDecoder.h"                     // the reference's own base class
// (the class then lives in the global namespace next to ::DCSDecoder and ::DCSDecoderNative)
#define DCSHIP_NAMESPACE_BEGIN
#define DCSHIP_NAMESPACE_END
#else
#define DCSHIP_NAMESPACE_BEGIN namespace dcship {
#define DCSHIP_NAMESPACE_END }
namespace dcship {

// ---------------------------------------------------------------------------------------------------------
// Mirror of the client-facing part of class DCSDecoder (DCSDecoder.h:107-1372)
// ---------------------------------------------------------------------------------------------------------
class DCSDecoder
{
public:
    static const int SAMPLE_RATE = 31250;                       // DCSDecoder.h:123

    class Host                                                  // DCSDecoder.h:127-193
    {
    public:
        virtual void ReceiveDataPort(uint8_t data) = 0;
        virtual void ClearDataPort() = 0;
        virtual void BootTimerControl(bool set) = 0;
        virtual ~Host() { }
    };
    class MinHost : public Host                                 // DCSDecoder.h:196-201
    {
        void ReceiveDataPort(uint8_t) override { }
        void ClearDataPort() override { }
        void BootTimerControl(bool) override { }
    };

    class ROMPointer                                            // DCSDecoder.h:730-785
    {
    public:
        ROMPointer() { }
        ROMPointer(int chipSelect, const uint8_t *p) : chipSelect(chipSelect), p(p) { }
        int chipSelect = 0;
        const uint8_t *p = nullptr;
        bool IsNull() const { return p == nullptr; }
        void Clear() { chipSelect = 0; p = nullptr; }
    };

    enum class OSVersion { Unknown, Invalid, OS93a, OS93b, OS94, OS95 };        // DCSDecoder.h:846-900

    explicit DCSDecoder(Host *host) : host(host) { }
    virtual ~DCSDecoder() { }

    virtual const char *Name() const = 0;                       // DCSDecoder.h:210
    virtual void SetMasterVolume(int vol) = 0;                  // DCSDecoder.h:546
    void SetDefaultVolume(int vol) { defaultVolume = vol; }     // DCSDecoder.h:559

    // DCSDecoder.cpp:1518-1540: initialise the subclass and enter the Running state
    void SoftBoot();
    // DCSDecoder.cpp:1579-1690: one PCM sample; refills 240 samples through MainLoop() when empty.  Inline: callers pull samples
    // one at a time in a bare loop, 240 calls a frame (DCSEncoder.cpp:565-567, DCSExplorer.cpp:1709-1711).
    int16_t GetNextSample()
    {
        if (sampleCounter < DCS_FRAME_SAMPLES && state == State::Running)
            return outputBuffer[sampleCounter++];
        return RefillAndGetSample();
    }

    bool IsOK() const { return state != State::DecoderFatalError && state != State::InitializationError; }
    bool IsRunning() const { return state == State::Running; }
    const std::string &GetErrorMessage() const { return errorMessage; }

    // subclass registry (DCSDecoder.h:1115-1132, DCSDecoder.cpp:1955-1974): `--decoder=<name>`
    struct Registration
    {
        using FactoryFunc = std::function<DCSDecoder *(Host *)>;
        Registration(const char *name, const char *desc, FactoryFunc factory);
        std::string name, desc;
        FactoryFunc factory;
    };
    static const std::map<std::string, const Registration &> &GetRegistrationMap();

protected:
    int16_t RefillAndGetSample();                               // the rest of GetNextSample: state check, MainLoop()
    virtual bool Initialize() = 0;                              // DCSDecoder.h:1137
    virtual void IRQ2Handler() = 0;                             // DCSDecoder.h:1140
    virtual void MainLoop() = 0;                                // DCSDecoder.h:1143

    enum class State { HardBoot, Running, DecoderFatalError, InitializationError };
    State state = State::HardBoot;
    std::string errorMessage;
    Host *host;
    OSVersion osVersion = OSVersion::Unknown;
    int defaultVolume = 0x67;                                   // DCSDecoder.h:1146
    int16_t outputBuffer[DCS_FRAME_SAMPLES] = { 0 };            // the "autobuffer" half MainLoop fills
    int sampleCounter = 30000;
};

}   // namespace dcship
#endif  // DCSHIP_USE_REFERENCE_BASE

DCSHIP_NAMESPACE_BEGIN

// ---------------------------------------------------------------------------------------------------------
// Mirror of class DCSDecoderNative's public surface (DCSDecoderNative.h:11-129) plus the ROM-facing part of
// DCSDecoder's (DCSDecoder.h:230-500); the sequencer runs on the host, the frame decode on the GPU
// ---------------------------------------------------------------------------------------------------------
class DCSDecoderHIP : public DCSDecoder
{
public:
    explicit DCSDecoderHIP(Host *host, int deviceId = 0);
    ~DCSDecoderHIP() override;

    const char *Name() const override { return "MI355X HIP batch decoder"; }

#ifndef DCSHIP_USE_REFERENCE_BASE
    // ---- ROMs (DCSDecoder.h:230-360) ------------------------------------------------------------------------
    enum class HWVersion { Unknown, Invalid, DCS93, DCS95 };                    // DCSDecoder.h:816-822
    void AddROM(int n, const uint8_t *data, size_t size);                       // DCSDecoder.cpp:26-66 (the image is copied)
    bool LoadROMFromZipFile(const char *zipFileName, const char *explicitU2 = nullptr, std::string *errorDetails = nullptr);
    uint8_t CheckROMs();                                                        // DCSDecoder.cpp:236-495: 1 = good, else Ux
    void SetVersions(HWVersion hw, OSVersion os);                               // explicit override (images without ADSP code)
    HWVersion GetHWVersion() const { return hwVersion; }
    OSVersion GetOSVersion() const { return osVersion; }
    int GetVersionNumber() const;                                               // DCSDecoder.cpp:497-503
    uint16_t GetMaxTrackNumber() const;                                         // DCSDecoder.h:360
    struct TrackInfo { uint32_t address = 0; int channel = 0; int type = 0; int deferCode = 0; uint32_t time = 0; bool looping = false; };
    bool GetTrackInfo(uint16_t trackNumber, TrackInfo &ti);                     // DCSDecoder.cpp:672-905
    std::vector<DcsTrackOp> DecompileTrackProgram(uint16_t trackNumber);        // DCSDecoder.cpp:907-1160 (no text)
    std::vector<uint32_t> ListStreams();                                        // DCSDecoder.cpp:1248-1293
    ROMPointer MakeROMPointer(uint32_t linearAddress) const;                    // DCSDecoder.cpp:68-76
    void WriteDataPort(uint8_t data);                           // DCSDecoder.cpp:1529-1543: the WPC board's commands
#endif

    // ---- playing (DCSDecoder.h:540-620, DCSDecoderNative.h:34-129) -------------------------------------------
    // DCSDecoderNative.h:34 -- no ROMs: streams come from the caller, the OS version is given
    void InitStandalone(OSVersion osVersion);
    void SetMasterVolume(int vol) override;                     // DCSDecoderNative.h:47
    void SetReportedVersionNumber(uint16_t vsn);
    void AddTrackCommand(uint16_t trackNum);                    // DCSDecoderNative.cpp:1475
    void ClearTracks();                                         // DCSDecoderNative.h:126

    // DCSDecoderNative.h:98, the reference's signature.  The reference takes a bare pointer and trusts the stream to end: so
    // does this member, except that a pointer into one of the decoder's ROM images is bounded by that image's end, and any
    // other pointer by 64 MB (the stream's own frame count and codes end it long before; bytes past a bound read as zero).
    void LoadAudioStream(int channel, const ROMPointer &streamPtr, int mixingLevel);
    // the same for a caller that knows the size of the buffer behind streamPtr (not in the reference)
    void LoadAudioStreamBounded(int channel, const ROMPointer &streamPtr, int mixingLevel, size_t maxLen);
    bool IsStreamPlaying(int channel);                          // DCSDecoderNative.h:101

    struct StreamInfo                                           // DCSDecoderNative.h:106-122
    {
        int nFrames;
        int nBytes;
        int formatType;
        int formatSubType;
        uint8_t header[16];
    };
    StreamInfo GetStreamInfo(const ROMPointer &streamPtr);                      // DCSDecoderNative.h:123, the reference's signature
    StreamInfo GetStreamInfoBounded(const ROMPointer &streamPtr, size_t maxLen);    // (buffer size known; not in the reference)

    // ---- the batch-submit path (new): decode whole streams, each played alone from a fresh decoder at
    // (volume, mixingLevel), extraFrames taper frames appended per stream; one kernel launch for all.
    struct BatchStream
    {
        const uint8_t *data;
        size_t len;
        int volume;
        int mixingLevel;
    };
    bool DecodeStreamsBatch(const std::vector<BatchStream> &streams, unsigned extraFrames,
                            std::vector<int16_t> &pcm, std::vector<uint32_t> *firstFrameOfStream = nullptr);

    // Live playback decodes several MainLoop ticks per kernel launch.  The sequencer simply runs that far ahead; anything that
    // can change what it does -- WriteDataPort, AddTrackCommand, LoadAudioStream, SetMasterVolume, ClearTracks -- first takes it
    // back to the last frame handed out, so the result does not depend on the look-ahead.  By DEFAULT the look-ahead is the
    // decoder's own business ("N frames of look-ahead when no commands are pending"): behind a command as many ticks as
    // the caller pulled between its last two commands (at least kFirstLookahead), twice as many with every refill that no command
    // preceded (eight times for a caller that has not sent a command since its first frame), up to kMaxLookahead, and never further than two ticks into silence (nothing playing, no track program,
    // nothing queued).  A caller that pulls samples in a bare loop and has
    // never heard of look-ahead gets this.  SetLookahead(n), n >= 1, fixes it at n ticks per launch (1 = tick by tick, as the
    // reference works; for measurements and tests); SetLookahead(0) gives it back to the decoder.
    void SetLookahead(int frames) { lookahead = frames < 0 ? 0 : frames > kMaxLookahead ? kMaxLookahead : frames; }
    static const int kFirstLookahead = 64, kMaxLookahead = 4096;

protected:
    bool Initialize() override;
    // the base class queues what the host writes to the data port and calls this per queued byte from GetNextSample
    // (DCSDecoder.cpp:1625-1626); the mirror base has no queue, its WriteDataPort goes to the sequencer directly
    void IRQ2Handler() override;
    void MainLoop() override;

private:
    void Sync();                                // back to the state after the last frame handed out
    bool Refill();                              // plan ahead and decode: one launch
    void ReleaseContext();
    size_t BytesBehind(const ROMPointer &p) const;     // to the end of the ROM image p points into, else 64 MB
    DcsOsVersion AbiOs() const;
    bool EnsureRoms();

    DcsCtx *ctx = nullptr;
    DcsRomSet *roms = nullptr;
    DcsSequencer *seq = nullptr;
    int deviceId;
#ifdef DCSHIP_USE_REFERENCE_BASE
    uint16_t outputBuffer[0x1E0] = { 0 };       // what the base's autobuffer reads (DCSDecoderNative.cpp:3203)
#else
    HWVersion hwVersion = HWVersion::Unknown;
    uint32_t nominalVersion = 0;
#endif
    uint16_t reportedVersion = 0x0106;
    int masterVolume = -1;                      // last SetMasterVolume before the sequencer existed
    int lookahead = 0;                          // 0 = the decoder's own (see SetLookahead)
    int curLookahead = kFirstLookahead;         // ... which is this many ticks for the next refill
    const int16_t *ready = nullptr;             // frames decoded ahead: readyCount x 240 samples in the context's pinned memory
    uint32_t readyCount = 0;
    uint32_t handedOut = 0;                     // frames of the current batch already handed out
    uint64_t fatalTick = ~uint64_t(0);          // first tick the sequencer answers with silence (DecoderFatalError), as of the last refill
    std::vector<DcsHostByte> hostBytes;         // bytes for the host, by tick; [hostNext, end) not delivered yet
    size_t hostNext = 0;
    uint64_t nextTick = 0;                      // tick of the next frame to hand out
    uint64_t lastCommandTick = 0;               // ... and what it was when the last command arrived
    uint64_t lastQuiet = 0;                     // frames handed out between the last two commands
    std::string zipError;
    struct { double loadUs = 0, planUs = 0, decodeUs = 0, syncUs = 0; unsigned refills = 0, syncs = 0; } stats;    // DCS_CLASS_STATS=1
};

DCSHIP_NAMESPACE_END

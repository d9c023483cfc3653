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
// In this repository the base-class mirror lives in namespace dcship so that the header is
// self-contained; inside the reference tree DCSDecoderHIP derives from the real ::DCSDecoder instead
// (INTEGRATION.md shows the ten-line adapter).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "dcs_hip.h"

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
    // DCSDecoder.cpp:1579-1690: one PCM sample; refills 240 samples through MainLoop() when empty
    int16_t GetNextSample();

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

// ---------------------------------------------------------------------------------------------------------
// Mirror of class DCSDecoderNative's public surface (DCSDecoderNative.h:11-129), frame decode on the GPU
// ---------------------------------------------------------------------------------------------------------
class DCSDecoderHIP : public DCSDecoder
{
public:
    explicit DCSDecoderHIP(Host *host, int deviceId = 0);
    ~DCSDecoderHIP() override;

    const char *Name() const override { return "MI355X HIP batch decoder"; }

    // DCSDecoderNative.h:34 -- no ROMs: streams come from the caller, the OS version is given
    void InitStandalone(OSVersion osVersion);
    void SetMasterVolume(int vol) override;                     // DCSDecoderNative.h:47
    void SetReportedVersionNumber(uint16_t vsn) { reportedVersion = vsn; }

    // DCSDecoderNative.h:98.  The reference takes a bare pointer and trusts the stream to end; pass
    // maxLen when the size of the buffer behind streamPtr is known (bytes past it read as zero).
    void LoadAudioStream(int channel, const ROMPointer &streamPtr, int mixingLevel, size_t maxLen = size_t(1) << 26);
    bool IsStreamPlaying(int channel);                          // DCSDecoderNative.h:101

    struct StreamInfo                                           // DCSDecoderNative.h:106-122
    {
        int nFrames;
        int nBytes;
        int formatType;
        int formatSubType;
        uint8_t header[16];
    };
    StreamInfo GetStreamInfo(const ROMPointer &streamPtr, size_t maxLen = size_t(1) << 26);

    void ClearTracks();                                         // DCSDecoderNative.h:126
    // Track programs live in the ROM catalog, which is outside this path;
    // the command is recorded and ignored.
    void AddTrackCommand(uint16_t trackNum) { ignoredCommands.push_back(trackNum); }

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

    // Live playback decodes `frames` MainLoop ticks per kernel launch (default 1 = tick by tick).  Any
    // LoadAudioStream / SetMasterVolume / ClearTracks discards ticks decoded ahead and replans.
    void SetLookahead(int frames) { lookahead = frames < 1 ? 1 : frames > 4096 ? 4096 : frames; }

protected:
    bool Initialize() override;
    void IRQ2Handler() override { }
    void MainLoop() override;

private:
    struct Channel
    {
        bool active = false;
        std::vector<uint8_t> bytes;             // private copy: the stream must outlive the lookahead
        std::vector<DcsFrameIndex> index;
        DcsStreamInfo info{};
        uint32_t pos = 0;                       // next frame
        int level = 0;                          // mixer[ch].curLevel (level byte << 6)
        uint16_t mixMul = 0x7FFF;               // Channel::mixingMultiplier (DCSDecoderNative.h:514)
        bool stopPending = false;               // AudioStream::stop: the next tick resets the mixer level (:95-116)
    };
    void PlanAndDecode();
    void Invalidate();
    DcsOsVersion AbiOs() const;

    DcsCtx *ctx = nullptr;
    int deviceId;
    uint16_t reportedVersion = 0x0106;
    uint16_t volumeMultiplier = 0x0391;         // DCSDecoderNative.h:161
    Channel channel[DCS_MAX_CHANNELS];
    int lookahead = 1;
    std::deque<std::vector<int16_t>> ready;     // frames decoded ahead
    struct Snapshot
    {
        uint32_t pos[DCS_MAX_CHANNELS]; bool active[DCS_MAX_CHANNELS]; uint16_t mixMul[DCS_MAX_CHANNELS];
        int level[DCS_MAX_CHANNELS]; bool stopPending[DCS_MAX_CHANNELS]; int16_t tail[16];
    };
    std::deque<Snapshot> after;                 // decoder state after each ready frame
    Snapshot rewind{};                          // decoder state after the last frame handed out
    int16_t tail[16] = { 0 };                   // overlapBuffer (DCSDecoderNative.h:149)
    std::vector<uint16_t> ignoredCommands;
};

}   // namespace dcship

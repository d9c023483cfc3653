// dcs_validate_driver -- the caller's side of `DCSExplorer --autoplay --silent --terse --validate=<log>` (DCSExplorer.cpp:1029-1566;
// it is the reference's own test, DCSDecoder/Tests/test-all.bat:59): two decoders on the same ROMs, booted the same way
// (HardBoot, StartSelfTests, SetDefaultVolume), fed the same data-port bytes -- autoplay starts every type-1 track in turn,
// two bytes per track, when the previous one's running time has passed --, 240 samples pulled from each per frame and compared
// sample by sample, their bytes to the host compared per frame, differing frames written to the log in the reference's layout,
// and the report at the end ("Validation Succeeded" iff no frame and no data-port difference).
//
// The reference validates DCSDecoderNative against DCSDecoderEmulated, which executes the ROM's own ADSP-2105 code; the synthetic
// ROM sets of this repository have no such code, so the reference decoder here is the reference's unmodified DCSDecoderNative
// and the decoder under test is DCSDecoderHIP behind the reference's real base class, taken from the real registration map and
// used through a plain DCSDecoder* -- everything this loop calls is a member of the base.  Built by oracle/Makefile (target
// validate; build container only, the binary travels under oracle/_ref/); our code, the shape of the caller, test
// infrastructure only.
//
//   dcs_validate_hip <volume> <log file> <report file> [--native] [--flip <frame>] <zip base name> <chip>=<romfile> ...
//     --native      the decoder under test is a second DCSDecoderNative (what a faultless run must print)
//     --flip <n>    one sample of the decoder under test is changed in frame n (exercises the log of a differing frame)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <list>
#include <memory>
#include <string>
#include <vector>
#include "DCSDecoder.h"
#include "DCSDecoderNative.h"

struct LoggingHost : DCSDecoder::Host
{
    DCSDecoder *decoder = nullptr;
    std::list<uint8_t> history;
    void ReceiveDataPort(uint8_t data) override { history.push_back(data); }
    void ClearDataPort() override { }
    void BootTimerControl(bool) override { }
    void logHistory(FILE *fp) const
    {
        fprintf(fp, "Data port bytes sent to host from %s:", decoder->Name());
        for (uint8_t b : history)
            fprintf(fp, " %02x", b);
        fprintf(fp, "\n");
    }
};

// the last sixteen bytes sent to the decoders, printed in front of a differing frame (:1133-1172)
struct CommandRing
{
    struct Entry { uint64_t frame; uint8_t byte; } e[16];
    int write = 0;
    size_t count = 0;
    void add(uint64_t frame, uint8_t b)
    {
        e[write] = Entry{ frame, b };
        write = (write + 1) % 16;
        count += count >= 16 ? 0 : 1;
    }
    void print(FILE *fp)
    {
        if (count == 0)
            return;
        fprintf(fp, "Recent commands: ");
        uint64_t at = UINT64_MAX;
        int i = (write - static_cast<int>(count) + 16) % 16;
        for (size_t k = 0 ; k < count ; ++k, i = (i + 1) % 16)
        {
            if (e[i].frame != at)
            {
                fprintf(fp, "%sFrame %llu:", at == UINT64_MAX ? "" : "; ", static_cast<unsigned long long>(e[i].frame));
                at = e[i].frame;
            }
            fprintf(fp, " %02x", e[i].byte);
        }
        fprintf(fp, "\n");
        count = 0;
    }
};

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
    if (argc < 6) { fprintf(stderr, "usage: see the comment at the top of dcs_validate_driver.cpp\n"); return 2; }
    const int volume = atoi(argv[1]);
    FILE *logFp = fopen(argv[2], "w"), *reportFp = fopen(argv[3], "w");
    if (logFp == nullptr || reportFp == nullptr) { fprintf(stderr, "cannot write the log or the report\n"); return 2; }
    int argi = 4;
    bool nativeUnderTest = false;
    long long flipFrame = -1;
    for ( ; argi < argc && argv[argi][0] == '-' ; ++argi)
    {
        if (strcmp(argv[argi], "--native") == 0) nativeUnderTest = true;
        else if (strcmp(argv[argi], "--flip") == 0 && argi + 1 < argc) flipFrame = atoll(argv[++argi]);
        else { fprintf(stderr, "unknown option %s\n", argv[argi]); return 2; }
    }
    if (argi >= argc) { fprintf(stderr, "missing zip base name\n"); return 2; }
    const std::string romName = argv[argi++];

    LoggingHost mainHost, refHost;
    std::unique_ptr<DCSDecoder> decoder, refDecoder;
    if (nativeUnderTest)
        decoder.reset(new DCSDecoderNative(&mainHost));
    else
    {
        auto &registry = DCSDecoder::GetRegistrationMap();          // as `--decoder=hip` gets it (:457-488)
        auto it = registry.find("hip");
        if (it == registry.end()) { fprintf(stderr, "decoder 'hip' is not registered\n"); return 3; }
        decoder.reset(it->second.factory(&mainHost));
    }
    refDecoder.reset(new DCSDecoderNative(&refHost));
    mainHost.decoder = decoder.get();
    refHost.decoder = refDecoder.get();

    std::vector<std::vector<uint8_t>> roms;
    for ( ; argi < argc ; ++argi)
    {
        const char *eq = strchr(argv[argi], '=');
        if (eq == nullptr) { fprintf(stderr, "bad ROM argument %s\n", argv[argi]); return 2; }
        roms.push_back(slurp(eq + 1));
        decoder->AddROM(atoi(argv[argi]), roms.back().data(), roms.back().size());
        refDecoder->AddROM(atoi(argv[argi]), roms.back().data(), roms.back().size());
    }
    if (decoder->CheckROMs() != 1 || refDecoder->CheckROMs() != 1) { fprintf(stderr, "CheckROMs failed\n"); return 4; }
    const uint16_t maxTrack = decoder->GetMaxTrackNumber();

    fprintf(logFp, "DCSExplorer - validation mode log, ROM %s\nListing frames containing differences in PCM output\n"
                   "%s output shown on left | Reference emulator output shown on right\n\n", romName.c_str(), decoder->Name());

    // boot both, the decoder under test first (:1109-1119)
    decoder->HardBoot();
    decoder->StartSelfTests();
    decoder->SetDefaultVolume(volume);
    refDecoder->SetDefaultVolume(volume);
    refDecoder->HardBoot();
    refDecoder->StartSelfTests();

    CommandRing ring;
    uint64_t sampleErrors = 0, frameDiffs = 0, portDiffs = 0, totalPlayTime = 0, nextCommandFrame = 0;
    int track = -1, tracksPlayed = 0;
    bool quit = false;
    std::string stopped;
    uint64_t frame = 0;
    for ( ; !quit ; ++frame)
    {
        if (!decoder->IsOK()) { stopped = std::string("Decoder error: ") + decoder->GetErrorMessage(); break; }
        if (!refDecoder->IsOK()) { stopped = std::string("Error in emulator: ") + refDecoder->GetErrorMessage(); break; }

        // autoplay: the next type-1 track when the last one's time is up (:1272-1336)
        if (decoder->IsRunning() && frame >= nextCommandFrame)
            for (;;)
            {
                const uint16_t n = static_cast<uint16_t>(++track);
                if (n > maxTrack) { quit = true; break; }
                DCSDecoder::TrackInfo ti;
                if (!decoder->GetTrackInfo(n, ti) || ti.type != 1)
                    continue;
                const uint8_t b0 = static_cast<uint8_t>(n >> 8), b1 = static_cast<uint8_t>(n);
                decoder->WriteDataPort(b0); decoder->WriteDataPort(b1);
                refDecoder->WriteDataPort(b0); refDecoder->WriteDataPort(b1);
                ring.add(frame, b0); ring.add(frame, b1);
                track = n;
                ++tracksPlayed;
                totalPlayTime += ti.time;
                nextCommandFrame = frame + ti.time + 1;
                break;
            }

        int16_t mine[240], theirs[240];
        for (int16_t &s : mine) s = decoder->GetNextSample();
        for (int16_t &s : theirs) s = refDecoder->GetNextSample();
        if (static_cast<long long>(frame) == flipFrame)
            mine[17] = static_cast<int16_t>(mine[17] ^ 0x0100);

        int diffs = 0;
        for (int i = 0 ; i < 240 ; ++i)
            diffs += mine[i] != theirs[i] ? 1 : 0;
        const bool portDiffers = refHost.history != mainHost.history;
        portDiffs += portDiffers ? 1 : 0;
        if (diffs != 0 || portDiffers)
            ring.print(logFp);
        if (diffs != 0)
        {
            sampleErrors += static_cast<uint64_t>(diffs);
            ++frameDiffs;
            fprintf(logFp, "--- Frame %llu - %d sample differences ---\n", static_cast<unsigned long long>(frame), diffs);
            for (int i = 0 ; i < 240 ; i += 16)
            {
                for (int k = 0 ; k < 16 ; ++k) fprintf(logFp, "%6d ", mine[i + k]);
                fprintf(logFp, "|");
                for (int k = 0 ; k < 16 ; ++k) fprintf(logFp, " %6d", theirs[i + k]);
                fprintf(logFp, "\n");
            }
            fprintf(logFp, "\n");
        }
        if (portDiffers)
        {
            fprintf(logFp, "--- Frame %llu - data port traffic was different ---\n", static_cast<unsigned long long>(frame));
            mainHost.logHistory(logFp);
            refHost.logHistory(logFp);
            fprintf(logFp, "\n");
        }
        mainHost.history.clear();
        refHost.history.clear();
    }

    // the report (:1517-1541), to the report file and to the end of the log
    for (FILE *fp : { reportFp, logFp })
    {
        if (!stopped.empty())
            fprintf(fp, "%s\n\n", stopped.c_str());
        fprintf(fp, "***** Validation Test Report *****\n\nROM file:          %s\nDecoder tested:    %s\nReference decoder: %s\nResult:            %s\n\n",
                romName.c_str(), decoder->Name(), refDecoder->Name(), frameDiffs == 0 && portDiffs == 0 ? "Validation Succeeded" : "Validation Failed");
        if (frameDiffs == 0)
            fprintf(fp, "No PCM sample differences detected - playback from both sources matched exactly\n");
        else
            fprintf(fp, "PCM sample differences were detected:\n  Total number of non-matching PCM samples: %llu\n  Number of frames containing differences:  %llu\n",
                    static_cast<unsigned long long>(sampleErrors), static_cast<unsigned long long>(frameDiffs));
        if (portDiffs == 0)
            fprintf(fp, "No data port traffic differences detected\n");
        else
            fprintf(fp, "Data port traffic differences were detected\n  Number of frames with differing data port bytes: %llu\n",
                    static_cast<unsigned long long>(portDiffs));
        const double seconds = static_cast<double>(totalPlayTime) * 7.68 / 1000.0;
        const int mm = static_cast<int>(seconds / 60.0), ss = static_cast<int>(seconds - mm * 60.0);
        fprintf(fp, "%d tracks tested (%d:%02d play time, %llu frames)\n", tracksPlayed, mm, ss, static_cast<unsigned long long>(frame));
    }
    fclose(logFp);
    fclose(reportFp);
    return 0;
}

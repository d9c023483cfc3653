// dcs_refbase_test -- DCSDecoderHIP compiled against the REAL base class of the reference (DCSDecoder/DCSDecoder.h,
// -DDCSHIP_USE_REFERENCE_BASE) and driven the way DCSExplorer drives whatever `--decoder=<name>` names
// (DCSExplorer.cpp:457-488): the object comes out of the real DCSDecoder::GetRegistrationMap() and is used through
// a plain DCSDecoder* only -- AddROM, CheckROMs, SetDefaultVolume, SoftBoot, SetMasterVolume, WriteDataPort,
// GetNextSample.  Test infrastructure: built by oracle/Makefile (target refbase) into oracle/_ref/, where the
// reference's base class is compiled in; never part of libdcs_hip.so.
//
//   dcs_refbase_test standalone <os 0..3> <volume> <level> <lookahead> <nFrames> <stream.bin> <out.pcm>
//     the ROM-less recipe of DCSEncoder.cpp:522-571 and EncoderTester.cpp:85-137 with the class under its own name, as
//     those callers hold it: InitStandalone(os), SetDefaultVolume, SoftBoot, LoadAudioStream(0, ROMPointer(0, p), level),
//     nFrames x 240 GetNextSample -- the sample pump, the boot state machine and the autobuffer are the REAL base's
//   dcs_refbase_test <volume> <nTicks> <events.txt> <outPrefix> <chip>=<romfile> ...
//     events.txt: one "<tick> <kind> <value>" per line; kind 0 = WriteDataPort(value), 2 = SetMasterVolume(value)
//     writes <outPrefix>.pcm (int16, nTicks x 240), <outPrefix>.host ("<tick> <byte>" per byte sent to the host) and
//     <outPrefix>.info (name, POST code, IsOK)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "DCSDecoder.h"
#include "DCSDecoderHIP.h"

struct CaptureHost : DCSDecoder::Host
{
    std::vector<std::pair<unsigned, unsigned>> bytes;
    unsigned tick = 0;
    void ReceiveDataPort(uint8_t data) override { bytes.emplace_back(tick, data); }
    void ClearDataPort() override { }
    void BootTimerControl(bool) override { }
};

static std::vector<uint8_t> readFile(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0)
        v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

static int standalone(int argc, char **argv)
{
    if (argc != 9) { fprintf(stderr, "usage: see the comment at the top of dcs_refbase_test.cpp\n"); return 2; }
    static const DCSDecoder::OSVersion kOs[4] = { DCSDecoder::OSVersion::OS93a, DCSDecoder::OSVersion::OS93b, DCSDecoder::OSVersion::OS94, DCSDecoder::OSVersion::OS95 };
    const int os = atoi(argv[2]), volume = atoi(argv[3]), level = atoi(argv[4]), lookahead = atoi(argv[5]), nFrames = atoi(argv[6]);
    std::vector<uint8_t> stream = readFile(argv[7]);
    stream.resize(stream.size() + 64, 0);           // (the reference's callers hand over buffers that end behind the stream too)
    DCSDecoder::MinHost host;
    DCSDecoderHIP dec(&host);
    dec.SetLookahead(lookahead);
    dec.InitStandalone(kOs[os & 3]);
    dec.SetDefaultVolume(volume);
    dec.SoftBoot();
    if (!dec.IsOK()) { fprintf(stderr, "decoder not OK: %s\n", dec.GetErrorMessage().c_str()); return 4; }
    dec.LoadAudioStream(0, DCSDecoder::ROMPointer(0, stream.data()), level);
    std::vector<int16_t> pcm;
    for (int f = 0 ; f < nFrames ; ++f)
        for (int i = 0 ; i < 240 ; ++i)
            pcm.push_back(dec.GetNextSample());
    if (!dec.IsOK()) { fprintf(stderr, "decoder failed: %s\n", dec.GetErrorMessage().c_str()); return 5; }
    FILE *f = fopen(argv[8], "wb");
    if (f == nullptr) return 2;
    fwrite(pcm.data(), sizeof(int16_t), pcm.size(), f);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && strcmp(argv[1], "standalone") == 0)
        return standalone(argc, argv);
    if (argc < 6) { fprintf(stderr, "usage: see the comment at the top of dcs_refbase_test.cpp\n"); return 2; }
    const int volume = atoi(argv[1]), nTicks = atoi(argv[2]);
    const std::string prefix = argv[4];
    struct Event { unsigned tick, kind, value; };
    std::vector<Event> events;
    if (FILE *f = fopen(argv[3], "r"))
    {
        Event e;
        while (fscanf(f, "%u %u %u", &e.tick, &e.kind, &e.value) == 3)
            events.push_back(e);
        fclose(f);
    }

    CaptureHost host;
    // the decoder comes from the registration map of the REAL base class, as `--decoder=hip` would get it
    auto &reg = DCSDecoder::GetRegistrationMap();
    auto it = reg.find("hip");
    if (it == reg.end()) { fprintf(stderr, "decoder 'hip' is not in DCSDecoder::GetRegistrationMap()\n"); return 3; }
    DCSDecoder *dec = it->second.factory(&host);            // from here on: a DCSDecoder*, nothing else

    std::vector<std::vector<uint8_t>> roms;
    for (int i = 5 ; i < argc ; ++i)
    {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "bad ROM argument %s\n", argv[i]); return 2; }
        roms.push_back(readFile(eq + 1));
        dec->AddROM(atoi(argv[i]), roms.back().data(), roms.back().size());
    }
    const int post = dec->CheckROMs();
    dec->SetDefaultVolume(volume);
    dec->SoftBoot();
    dec->SetMasterVolume(volume);

    std::vector<int16_t> pcm;
    size_t e = 0;
    for (host.tick = 0 ; host.tick < static_cast<unsigned>(nTicks) ; ++host.tick)
    {
        for ( ; e < events.size() && events[e].tick <= host.tick ; ++e)
        {
            if (events[e].kind == 0) dec->WriteDataPort(static_cast<uint8_t>(events[e].value));
            else if (events[e].kind == 2) dec->SetMasterVolume(static_cast<int>(events[e].value));
            else { fprintf(stderr, "event kind %u is not reachable through a DCSDecoder*\n", events[e].kind); return 2; }
        }
        for (int k = 0 ; k < 240 ; ++k)
            pcm.push_back(dec->GetNextSample());
    }

    FILE *f = fopen((prefix + ".pcm").c_str(), "wb");
    fwrite(pcm.data(), sizeof(int16_t), pcm.size(), f);
    fclose(f);
    f = fopen((prefix + ".host").c_str(), "w");
    for (auto &b : host.bytes) fprintf(f, "%u %u\n", b.first, b.second);
    fclose(f);
    f = fopen((prefix + ".info").c_str(), "w");
    fprintf(f, "name=%s\npost=%d\nok=%d\nrunning=%d\nmaxtrack=%u\nerror=%s\n", dec->Name(), post, dec->IsOK() ? 1 : 0, dec->IsRunning() ? 1 : 0,
            static_cast<unsigned>(dec->GetMaxTrackNumber()), dec->GetErrorMessage().c_str());
    fclose(f);
    delete dec;
    return 0;
}

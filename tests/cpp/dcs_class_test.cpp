// dcs_class_test -- a C++ client written the way the reference's callers drive DCSDecoderNative
// (DCSEncoder.cpp:522-571, EncoderTester.cpp:85-137): MinHost + decoder + InitStandalone + SetDefaultVolume
// + SoftBoot + LoadAudioStream + 240 x GetNextSample per frame.  tests/test_gpu_class.py compares the PCM it
// writes with the oracle.
//
//   dcs_class_test live  <os> <volume> <lookahead> <out.pcm> <nFramesOut> <level0> <stream0.bin> [<level1> <stream1.bin> ...]
//   dcs_class_test batch <os> <volume> <extraFrames> <out.pcm> <level> <stream0.bin> [<stream1.bin> ...]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/DCSDecoderHIP.h"

using namespace dcship;

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

static DCSDecoder::OSVersion osOf(int os)
{
    switch (os)
    {
    case 0: return DCSDecoder::OSVersion::OS93a;
    case 1: return DCSDecoder::OSVersion::OS93b;
    case 2: return DCSDecoder::OSVersion::OS94;
    default: return DCSDecoder::OSVersion::OS95;
    }
}

int main(int argc, char **argv)
{
    if (argc < 7)
    {
        fprintf(stderr, "usage: see the comment at the top of dcs_class_test.cpp\n");
        return 2;
    }
    const std::string mode = argv[1];
    const int os = atoi(argv[2]), volume = atoi(argv[3]);
    DCSDecoder::MinHost host;

    // the decoder is created through the same registration map DCSExplorer's --decoder=<name> uses
    auto &reg = DCSDecoder::GetRegistrationMap();
    auto it = reg.find("hip");
    if (it == reg.end()) { fprintf(stderr, "decoder 'hip' is not registered\n"); return 3; }
    DCSDecoderHIP *dec = static_cast<DCSDecoderHIP *>(it->second.factory(&host));
    dec->InitStandalone(osOf(os));
    dec->SetDefaultVolume(volume);
    dec->SoftBoot();
    if (!dec->IsOK()) { fprintf(stderr, "decoder not OK: %s\n", dec->GetErrorMessage().c_str()); return 4; }

    std::vector<int16_t> pcm;
    if (mode == "live")
    {
        dec->SetLookahead(atoi(argv[4]));
        const int nFramesOut = atoi(argv[6]);
        std::vector<std::vector<uint8_t>> keep;
        int ch = 0;
        for (int i = 7 ; i + 1 < argc ; i += 2, ++ch)
        {
            keep.push_back(readFile(argv[i + 1]));
            dec->LoadAudioStreamBounded(ch, DCSDecoder::ROMPointer(0, keep.back().data()), atoi(argv[i]), keep.back().size());
        }
        auto info = dec->GetStreamInfoBounded(DCSDecoder::ROMPointer(0, keep[0].data()), keep[0].size());
        fprintf(stderr, "stream 0: %d frames, %d bytes, type %d/%d\n", info.nFrames, info.nBytes, info.formatType, info.formatSubType);
        // (IsStreamPlaying between every two frames, as EncoderTester.cpp:94-106 asks it: <OUT>.playing, one digit per frame and channel)
        std::string playing;
        for (int f = 0 ; f < nFramesOut ; ++f)
        {
            for (int i = 0 ; i < 240 ; ++i)
                pcm.push_back(dec->GetNextSample());
            for (int c = 0 ; c < ch ; ++c)
                playing += dec->IsStreamPlaying(c) ? '1' : '0';
            playing += '\n';
        }
        if (!dec->IsOK()) { fprintf(stderr, "decoder failed: %s\n", dec->GetErrorMessage().c_str()); return 5; }
        FILE *pf = fopen((std::string(argv[5]) + ".playing").c_str(), "w");
        if (pf != nullptr) { fputs(playing.c_str(), pf); fclose(pf); }
    }
    else if (mode == "script")
    {
        // ROM mode, the way DCSExplorer drives a decoder: AddROM per chip, CheckROMs, SoftBoot, then data-port
        // bytes / track commands while pulling samples.  argv: script os volume lookahead OUT nTicks <script file>
        // <u2> <u3> <u4>; script lines "tick kind value" (kind 0 WriteDataPort, 1 AddTrackCommand, 2 SetMasterVolume,
        // 3 ClearTracks); host bytes go to <OUT>.host as "tick byte" lines
        struct Cap : public DCSDecoder::Host
        {
            std::vector<std::pair<unsigned, unsigned>> log; unsigned tick = 0;
            void ReceiveDataPort(uint8_t d) override { log.emplace_back(tick, d); }
            void ClearDataPort() override { }
            void BootTimerControl(bool) override { }
        } cap;
        delete dec;
        dec = new DCSDecoderHIP(&cap);
        std::vector<std::vector<uint8_t>> images;
        for (int i = 8, chip = 2 ; i < argc ; ++i, ++chip)
        {
            images.push_back(readFile(argv[i]));
            dec->AddROM(chip, images.back().data(), images.back().size());
        }
        const uint8_t status = dec->CheckROMs();
        fprintf(stderr, "CheckROMs %d, version %04x, %d tracks, %zu streams\n", status, dec->GetVersionNumber(),
                dec->GetMaxTrackNumber() + 1, dec->ListStreams().size());
        dec->SetDefaultVolume(volume);
        dec->SoftBoot();
        dec->SetMasterVolume(volume);
        dec->SetLookahead(atoi(argv[4]));
        if (!dec->IsOK()) { fprintf(stderr, "decoder not OK: %s\n", dec->GetErrorMessage().c_str()); return 4; }
        std::vector<unsigned> ev;
        {
            FILE *f = fopen(argv[7], "r");
            unsigned a, b, c;
            while (f != nullptr && fscanf(f, "%u %u %u", &a, &b, &c) == 3) { ev.push_back(a); ev.push_back(b); ev.push_back(c); }
            if (f != nullptr) fclose(f);
        }
        const unsigned nTicks = static_cast<unsigned>(atoi(argv[6]));
        size_t e = 0;
        for (cap.tick = 0 ; cap.tick < nTicks ; ++cap.tick)
        {
            for ( ; e < ev.size() && ev[e] <= cap.tick ; e += 3)
            {
                if (ev[e + 1] == 0) dec->WriteDataPort(static_cast<uint8_t>(ev[e + 2]));
                else if (ev[e + 1] == 1) dec->AddTrackCommand(static_cast<uint16_t>(ev[e + 2]));
                else if (ev[e + 1] == 2) dec->SetMasterVolume(static_cast<int>(ev[e + 2]));
                else if (ev[e + 1] == 3) dec->ClearTracks();
            }
            for (int si = 0 ; si < 240 ; ++si)
                pcm.push_back(dec->GetNextSample());
        }
        FILE *hf = fopen((std::string(argv[5]) + ".host").c_str(), "w");
        for (auto &hb : cap.log) fprintf(hf, "%u %u\n", hb.first, hb.second);
        fprintf(hf, "fatal %d\n", dec->IsOK() ? 0 : 1);
        fclose(hf);
        FILE *out = fopen(argv[5], "wb");
        fwrite(pcm.data(), sizeof(int16_t), pcm.size(), out);
        fclose(out);
        delete dec;
        return 0;
    }
    else if (mode == "extract")
    {
        // the stream loop of DCSExplorer --extract-streams, written the way it is there (DCSExplorer.cpp:1670-1721,
        // :1900-1907): one decoder object, stream after stream; each stream also goes to <OUT>.<k>.wav
        dec->SetLookahead(atoi(argv[4]));
        dec->SetMasterVolume(volume);
        std::vector<std::vector<uint8_t>> keep;
        int k = 0;
        for (int i = 7 ; i + 1 < argc ; i += 2, ++k)
        {
            keep.push_back(readFile(argv[i + 1]));
            const std::vector<uint8_t> &data = keep.back();
            dec->LoadAudioStreamBounded(0, DCSDecoder::ROMPointer(0, data.data()), atoi(argv[i]), data.size());
            const int nFrames = ((data[0] << 8) | data[1]) + 2;
            const size_t first = pcm.size();
            for (int frame = 0 ; frame < nFrames ; ++frame)
            {
                for (int si = 0 ; si < 240 ; ++si)
                    pcm.push_back(dec->GetNextSample());
                if (frame + 2 >= nFrames)
                    dec->ClearTracks();
            }
            const std::string wav = std::string(argv[5]) + "." + std::to_string(k) + ".wav";
            if (dcs_write_wav(wav.c_str(), pcm.data() + first, static_cast<uint32_t>(nFrames)) != DCS_OK) return 7;
        }
        if (!dec->IsOK()) { fprintf(stderr, "decoder failed: %s\n", dec->GetErrorMessage().c_str()); return 5; }
    }
    else
    {
        const unsigned extra = static_cast<unsigned>(atoi(argv[4]));
        const int level = atoi(argv[6]);
        std::vector<std::vector<uint8_t>> keep;
        std::vector<DCSDecoderHIP::BatchStream> streams;
        for (int i = 7 ; i < argc ; ++i)
        {
            keep.push_back(readFile(argv[i]));
            streams.push_back({ keep.back().data(), keep.back().size(), volume, level });
        }
        if (!dec->DecodeStreamsBatch(streams, extra, pcm)) { fprintf(stderr, "batch decode failed\n"); return 6; }
    }
    FILE *out = fopen(argv[5], "wb");
    fwrite(pcm.data(), sizeof(int16_t), pcm.size(), out);
    fclose(out);
    delete dec;
    return 0;
}

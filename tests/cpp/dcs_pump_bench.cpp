// dcs_pump_bench -- samples per second through the decoder CLASS, the way the reference's callers use it: a bare
// GetNextSample() loop (DCSDecoder.cpp:1579), nothing else.  ONE source, three builds:
//
//   dcsexplorer_amd/dcs_pump_bench         DCSDecoderHIP on its own mirror of the base class (csrc/Makefile)
//   oracle/_ref/dcs_pump_bench_refbase     DCSDecoderHIP behind the reference's REAL ::DCSDecoder (-DDCSHIP_USE_REFERENCE_BASE)
//   oracle/_ref/dcs_pump_bench_native      the reference's own DCSDecoderNative (-DPUMP_NATIVE): the CPU pump beside it
//
// so that the three numbers come from the same caller code on the same box.  The caller never mentions look-ahead
// unless asked to (lookahead < 0 = whatever the decoder does by default; the native build ignores it).
//
//   dcs_pump_bench recipe  <os 0..3> <volume> <level> <lookahead> <reps> <out.pcm|-> <stream.bin>
//       the ROM-less recipe of DCSEncoder.cpp:522-571: per repetition a NEW decoder, InitStandalone, SetDefaultVolume,
//       SoftBoot, LoadAudioStream(0, ROMPointer(0, p), level), (nFrames + 1) x 240 GetNextSample
//   dcs_pump_bench extract <os 0..3> <volume> <level> <lookahead> <reps> <out.pcm|-> <stream0.bin> [<stream1.bin> ...]
//       the stream loop of DCSExplorer --extract-streams (DCSExplorer.cpp:1670-1721, :1900-1907) on ONE decoder: per stream
//       LoadAudioStream(0, ptr, level), nFrames + 2 frames, ClearTracks() behind each of the last two
//   dcs_pump_bench script  <volume> <lookahead> <reps> <out.pcm|-> <nTicks> <events.txt> <chip>=<romfile> ...
//       ROM mode (DCSExplorer.cpp:457-488): AddROM per chip, CheckROMs, SoftBoot, SetMasterVolume, then per tick the events
//       of that tick ("<tick> <kind> <value>": 0 WriteDataPort, 2 SetMasterVolume) and 240 GetNextSample
//
//   dcs_pump_bench oneshot <os 0..3> <volume> <level> <iters> <stream.bin> <nFrames> [<nFrames> ...]        (HIP builds only)
//       the one-shot C ABI underneath the class: dcs_decode_batch over the stream's first nFrames frames, `iters` calls, host
//       microseconds per call (median), next to what a synchronous call on the box cannot get under (dcs_ctx_call_floor)
//
// Prints one JSON object per run on stdout: the decoder's name, frames, per-repetition milliseconds (boot = construction
// to SoftBoot, play = first command to last sample), the FNV-1a-64 of the last repetition's PCM.  Test infrastructure.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#if defined(PUMP_NATIVE)
#include "DCSDecoder.h"
#include "DCSDecoderNative.h"
typedef DCSDecoderNative Decoder;
static const char *kBuild = "native";
#elif defined(DCSHIP_USE_REFERENCE_BASE)
#include "DCSDecoder.h"
#include "DCSDecoderHIP.h"
typedef DCSDecoderHIP Decoder;
static const char *kBuild = "hip-refbase";
#else
#include "../../include/DCSDecoderHIP.h"
using namespace dcship;
typedef DCSDecoderHIP Decoder;
static const char *kBuild = "hip-mirror";
#endif

static double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
    v.resize(v.size() + 64, 0);                 // (the reference's callers hand over buffers that end behind the stream too)
    return v;
}

static uint64_t fnv(const std::vector<int16_t> &pcm)
{
    uint64_t h = 0xcbf29ce484222325ull;
    const uint8_t *p = reinterpret_cast<const uint8_t *>(pcm.data());
    for (size_t i = 0 ; i < pcm.size() * 2 ; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

static void setLookahead(Decoder &dec, int lookahead)
{
#ifndef PUMP_NATIVE
    if (lookahead >= 1)
        dec.SetLookahead(lookahead);
#else
    (void)dec; (void)lookahead;
#endif
}

static void printRuns(const char *scenario, const Decoder *dec, int lookahead, size_t frames, const std::vector<double> &boot,
                      const std::vector<double> &play, const std::vector<int16_t> &pcm, const char *outPath)
{
    printf("{\"scenario\": \"%s\", \"build\": \"%s\", \"decoder\": \"%s\", \"lookahead\": %d, \"frames\": %zu, \"samples\": %zu, \"boot_ms\": [",
           scenario, kBuild, dec != nullptr ? dec->Name() : "", lookahead, frames, frames * 240);
    for (size_t i = 0 ; i < boot.size() ; ++i) printf("%s%.4f", i ? ", " : "", boot[i]);
    printf("], \"play_ms\": [");
    for (size_t i = 0 ; i < play.size() ; ++i) printf("%s%.4f", i ? ", " : "", play[i]);
    printf("], \"fnv1a64\": \"%016llx\"}\n", static_cast<unsigned long long>(fnv(pcm)));
    fflush(stdout);
    if (outPath != nullptr && strcmp(outPath, "-") != 0)
    {
        FILE *f = fopen(outPath, "wb");
        if (f == nullptr) { fprintf(stderr, "cannot write %s\n", outPath); exit(2); }
        fwrite(pcm.data(), sizeof(int16_t), pcm.size(), f);
        fclose(f);
    }
}

static const DCSDecoder::OSVersion kOs[4] = { DCSDecoder::OSVersion::OS93a, DCSDecoder::OSVersion::OS93b, DCSDecoder::OSVersion::OS94, DCSDecoder::OSVersion::OS95 };

static int recipe(int argc, char **argv)
{
    if (argc != 9) return 2;
    const int os = atoi(argv[2]), volume = atoi(argv[3]), level = atoi(argv[4]), lookahead = atoi(argv[5]), reps = atoi(argv[6]);
    const std::vector<uint8_t> stream = readFile(argv[8]);
    const size_t nFrames = ((static_cast<size_t>(stream[0]) << 8) | stream[1]) + 1;     // (one more: the fade to silence, DCSEncoder.cpp:562)
    std::vector<double> boot, play;
    std::vector<int16_t> pcm(nFrames * 240);
    std::unique_ptr<Decoder> last;
    for (int r = 0 ; r < reps ; ++r)
    {
        DCSDecoder::MinHost host;
        const double t0 = nowMs();
        std::unique_ptr<Decoder> dec(new Decoder(&host));
        setLookahead(*dec, lookahead);
        dec->InitStandalone(kOs[os & 3]);
        dec->SetDefaultVolume(volume);
        dec->SoftBoot();
        if (!dec->IsOK()) { fprintf(stderr, "decoder not OK: %s\n", dec->GetErrorMessage().c_str()); return 4; }
        const double t1 = nowMs();
        dec->LoadAudioStream(0, DCSDecoder::ROMPointer(0, stream.data()), level);
        int16_t *out = pcm.data();
        for (size_t f = 0 ; f < nFrames ; ++f)
        {
            int16_t buf[240];
            for (int s = 0 ; s < 240 ; ++s)
                buf[s] = dec->GetNextSample();
            memcpy(out, buf, sizeof(buf));      // (where the caller's WriteStream / fwrite stands)
            out += 240;
        }
        const double t2 = nowMs();
        if (!dec->IsOK()) { fprintf(stderr, "decoder failed: %s\n", dec->GetErrorMessage().c_str()); return 5; }
        boot.push_back(t1 - t0);
        play.push_back(t2 - t1);
        last = std::move(dec);
    }
    printRuns("recipe", last.get(), lookahead, nFrames, boot, play, pcm, argv[7]);
    return 0;
}

static int extract(int argc, char **argv)
{
    if (argc < 9) return 2;
    const int os = atoi(argv[2]), volume = atoi(argv[3]), level = atoi(argv[4]), lookahead = atoi(argv[5]), reps = atoi(argv[6]);
    std::vector<std::vector<uint8_t>> streams;
    size_t total = 0;
    for (int i = 8 ; i < argc ; ++i)
    {
        streams.push_back(readFile(argv[i]));
        total += ((static_cast<size_t>(streams.back()[0]) << 8) | streams.back()[1]) + 2;
    }
    std::vector<double> boot, play;
    std::vector<int16_t> pcm(total * 240);
    std::unique_ptr<Decoder> last;
    for (int r = 0 ; r < reps ; ++r)
    {
        DCSDecoder::MinHost host;
        const double t0 = nowMs();
        std::unique_ptr<Decoder> dec(new Decoder(&host));
        setLookahead(*dec, lookahead);
        dec->InitStandalone(kOs[os & 3]);
        dec->SetDefaultVolume(volume);
        dec->SoftBoot();
        dec->SetMasterVolume(volume);           // (DCSExplorer.cpp:1647)
        if (!dec->IsOK()) { fprintf(stderr, "decoder not OK: %s\n", dec->GetErrorMessage().c_str()); return 4; }
        const double t1 = nowMs();
        int16_t *out = pcm.data();
        for (const std::vector<uint8_t> &s : streams)
        {
            const unsigned nFrames = ((static_cast<unsigned>(s[0]) << 8) | s[1]) + 2;
            dec->LoadAudioStream(0, DCSDecoder::ROMPointer(0, s.data()), level);
            for (unsigned frame = 0 ; frame < nFrames ; ++frame)
            {
                int16_t buf[240];
                for (int si = 0 ; si < 240 ; ++si)
                    buf[si] = dec->GetNextSample();
                memcpy(out, buf, sizeof(buf));
                out += 240;
                if (frame + 2 >= nFrames)
                    dec->ClearTracks();
            }
        }
        const double t2 = nowMs();
        if (!dec->IsOK()) { fprintf(stderr, "decoder failed: %s\n", dec->GetErrorMessage().c_str()); return 5; }
        boot.push_back(t1 - t0);
        play.push_back(t2 - t1);
        last = std::move(dec);
    }
    printRuns("extract", last.get(), lookahead, total, boot, play, pcm, argv[7]);
    return 0;
}

static int script(int argc, char **argv)
{
    if (argc < 9) return 2;
    const int volume = atoi(argv[2]), lookahead = atoi(argv[3]), reps = atoi(argv[4]);
    const size_t nTicks = static_cast<size_t>(atol(argv[6]));
    struct Event { unsigned tick, kind, value; };
    std::vector<Event> events;
    {
        FILE *f = fopen(argv[7], "r");
        if (f == nullptr) { fprintf(stderr, "cannot open %s\n", argv[7]); return 2; }
        Event e;
        while (fscanf(f, "%u %u %u", &e.tick, &e.kind, &e.value) == 3)
            events.push_back(e);
        fclose(f);
    }
    std::vector<std::pair<int, std::vector<uint8_t>>> roms;
    for (int i = 8 ; i < argc ; ++i)
    {
        const char *eq = strchr(argv[i], '=');
        if (eq == nullptr) return 2;
        roms.emplace_back(atoi(argv[i]), readFile(eq + 1));
        roms.back().second.resize(roms.back().second.size() - 64);      // (ROM images have their exact size)
    }
    std::vector<double> boot, play;
    std::vector<int16_t> pcm(nTicks * 240);
    std::unique_ptr<Decoder> last;
    for (int r = 0 ; r < reps ; ++r)
    {
        DCSDecoder::MinHost host;
        const double t0 = nowMs();
        std::unique_ptr<Decoder> dec(new Decoder(&host));
        setLookahead(*dec, lookahead);
        for (auto &rom : roms)
            dec->AddROM(rom.first, rom.second.data(), rom.second.size());
        if (dec->CheckROMs() != 1) { fprintf(stderr, "CheckROMs failed\n"); return 4; }
        dec->SetDefaultVolume(volume);
        dec->SoftBoot();
        dec->SetMasterVolume(volume);
        if (!dec->IsOK()) { fprintf(stderr, "decoder not OK: %s\n", dec->GetErrorMessage().c_str()); return 4; }
        const double t1 = nowMs();
        size_t e = 0;
        int16_t *out = pcm.data();
        for (size_t tick = 0 ; tick < nTicks ; ++tick)
        {
            for ( ; e < events.size() && events[e].tick <= tick ; ++e)
            {
                if (events[e].kind == 0) dec->WriteDataPort(static_cast<uint8_t>(events[e].value));
                else if (events[e].kind == 2) dec->SetMasterVolume(static_cast<int>(events[e].value));
            }
            int16_t buf[240];
            for (int s = 0 ; s < 240 ; ++s)
                buf[s] = dec->GetNextSample();
            memcpy(out, buf, sizeof(buf));
            out += 240;
        }
        const double t2 = nowMs();
        if (!dec->IsOK()) { fprintf(stderr, "decoder failed: %s\n", dec->GetErrorMessage().c_str()); return 5; }
        boot.push_back(t1 - t0);
        play.push_back(t2 - t1);
        last = std::move(dec);
    }
    printRuns("script", last.get(), lookahead, nTicks, boot, play, pcm, argv[5]);
    return 0;
}

#ifndef PUMP_NATIVE
static int oneshot(int argc, char **argv)
{
    if (argc < 8) return 2;
    const int os = atoi(argv[2]), volume = atoi(argv[3]), level = atoi(argv[4]), iters = atoi(argv[5]);
    const std::vector<uint8_t> stream = readFile(argv[6]);
    const uint32_t total = (static_cast<uint32_t>(stream[0]) << 8) | stream[1];
    std::vector<DcsFrameIndex> index(total);
    DcsStreamInfo info;
    if (dcs_index_stream(static_cast<DcsOsVersion>(os), stream.data(), stream.size(), index.data(), total, &info) != DCS_OK) return 3;
    std::vector<uint16_t> mm(total);
    std::vector<uint8_t> vs(total);
    if (dcs_stream_params(static_cast<DcsOsVersion>(os), volume, level, 255, total, mm.data(), vs.data()) != DCS_OK) return 3;
    DcsCtx *ctx = nullptr;
    if (dcs_ctx_create(0, &ctx) != DCS_OK) { fprintf(stderr, "no context: %s\n", dcs_last_error(nullptr)); return 4; }
    printf("{\"scenario\": \"oneshot\", \"build\": \"%s\", \"iters\": %d, \"calls\": [", kBuild, iters);
    for (int a = 7 ; a < argc ; ++a)
    {
        const uint32_t n = static_cast<uint32_t>(atoi(argv[a]));
        if (n == 0 || n > static_cast<uint32_t>(info.nValidFrames)) return 2;
        std::vector<DcsSrcDesc> srcs(n);
        std::vector<DcsFrameJob> jobs(n);
        for (uint32_t f = 0 ; f < n ; ++f)
        {
            memset(&srcs[f], 0, sizeof(DcsSrcDesc));
            srcs[f].streamOff = 0; srcs[f].mixMul = mm[f]; srcs[f].format = static_cast<uint8_t>(info.format);
            srcs[f].hdrLen = static_cast<uint8_t>(info.hdrLen); srcs[f].idx = index[f];
            memset(&jobs[f], 0, sizeof(DcsFrameJob));
            jobs[f].firstSrc = f; jobs[f].nSrc = 1; jobs[f].volShift = vs[f];
            jobs[f].xform = os <= 1 ? DCS_XFORM_93 : DCS_XFORM_94;
            jobs[f].prev = f == 0 ? DCS_PREV_NONE : f - 1;
        }
        std::vector<int16_t> pcm(static_cast<size_t>(n) * 240);
        std::vector<uint32_t> err(n);
        std::vector<double> us;
        for (int i = 0 ; i < iters + 5 ; ++i)
        {
            const double t0 = nowMs();
            if (dcs_decode_batch(ctx, stream.data(), stream.size(), srcs.data(), n, jobs.data(), n, nullptr, 0, pcm.data(), err.data(), nullptr) != DCS_OK)
            { fprintf(stderr, "dcs_decode_batch: %s\n", dcs_last_error(ctx)); return 5; }
            if (i >= 5) us.push_back((nowMs() - t0) * 1e3);
        }
        std::sort(us.begin(), us.end());
        float fl = 0, fc = 0;
        dcs_ctx_call_floor(ctx, n, iters, &fl, &fc);
        printf("%s{\"frames\": %u, \"us_per_call\": %.2f, \"us_min\": %.2f, \"floor_launch_wait_us\": %.2f, \"floor_launch_copy_wait_us\": %.2f, \"fnv1a64\": \"%016llx\"}",
               a > 7 ? ", " : "", n, us[us.size() / 2], us[0], fl, fc, static_cast<unsigned long long>(fnv(pcm)));
    }
    printf("]}\n");
    dcs_ctx_destroy(ctx);
    return 0;
}
#endif

int main(int argc, char **argv)
{
    int rc = 2;
    if (argc > 1 && strcmp(argv[1], "recipe") == 0) rc = recipe(argc, argv);
    else if (argc > 1 && strcmp(argv[1], "extract") == 0) rc = extract(argc, argv);
    else if (argc > 1 && strcmp(argv[1], "script") == 0) rc = script(argc, argv);
#ifndef PUMP_NATIVE
    else if (argc > 1 && strcmp(argv[1], "oneshot") == 0) rc = oneshot(argc, argv);
#endif
    if (rc == 2)
        fprintf(stderr, "usage: see the comment at the top of dcs_pump_bench.cpp\n");
    return rc;
}

// dcs_class_fuzz -- DCSDecoderHIP (behind the reference's real base class) and the reference's unmodified DCSDecoderNative in ONE
// process, driven in lock step by a seeded random caller: LoadAudioStream on random channels at random levels, ClearTracks,
// SetMasterVolume, IsStreamPlaying, and pulls of random numbers of frames in between -- every sample and every answer compared.
// What it is after: the machinery between the caller and the kernels that the reference does not have -- the look-ahead that grows
// and is taken back by every command, the sequencer's snapshots and replays, streams walked on a second thread while their first
// frames are handed out, tails carried from refill to refill, channels that stop on damaged frames.  The caller never mentions
// look-ahead (argv can fix it for comparison).  Streams come in as files (seeded synthetic ones, a few of them damaged, made by
// tests/test_refbase.py).  Built by oracle/Makefile (target fuzz; build container only, the binary travels under oracle/_ref/); our
// code, test infrastructure only.
//
//   dcs_class_fuzz <os 0..3> <seed> <nOps> <lookahead|-1> <stream0.bin> [<stream1.bin> ...]
//   prints "ok: <frames> frames, <ops> operations ..." or the first difference, exit code 1
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <string>
#include <vector>
#include "DCSDecoder.h"
#include "DCSDecoderNative.h"
#include "DCSDecoderHIP.h"

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
    v.resize(v.size() + 4096, 0);               // (a damaged stream may read on behind its end: both decoders find zeros there)
    return v;
}

struct Rng
{
    uint64_t x;
    uint64_t next()
    {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return static_cast<uint32_t>(next() % n); }
};

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: see the comment at the top of dcs_class_fuzz.cpp\n"); return 2; }
    static const DCSDecoder::OSVersion kOs[4] = { DCSDecoder::OSVersion::OS93a, DCSDecoder::OSVersion::OS93b, DCSDecoder::OSVersion::OS94, DCSDecoder::OSVersion::OS95 };
    const int os = atoi(argv[1]) & 3, nOps = atoi(argv[3]), lookahead = atoi(argv[4]);
    Rng rng{ strtoull(argv[2], nullptr, 0) };
    std::vector<std::vector<uint8_t>> streams;
    for (int i = 5 ; i < argc ; ++i)
        streams.push_back(readFile(argv[i]));

    // (DCS_FUZZ_REF_ONLY=1: the reference alone runs the same calls -- to tell, where the process dies, whose fault it is: the
    // reference has undefined behaviour on some damaged streams)
    const bool refOnly = getenv("DCS_FUZZ_REF_ONLY") != nullptr;
    DCSDecoder::MinHost hostA, hostB;
    DCSDecoderNative ref(&hostA);
    DCSDecoderNative second(&hostB);
    DCSDecoderHIP hipReal(&hostB);
    struct Both
    {
        bool refOnly; DCSDecoderNative &n; DCSDecoderHIP &h;
        void SetLookahead(int v) { if (!refOnly) h.SetLookahead(v); }
        void InitStandalone(DCSDecoder::OSVersion v) { if (refOnly) n.InitStandalone(v); else h.InitStandalone(v); }
        void SetDefaultVolume(int v) { if (refOnly) n.SetDefaultVolume(v); else h.SetDefaultVolume(v); }
        void SoftBoot() { if (refOnly) n.SoftBoot(); else h.SoftBoot(); }
        bool IsOK() { return refOnly ? n.IsOK() : h.IsOK(); }
        std::string GetErrorMessage() { return refOnly ? n.GetErrorMessage() : h.GetErrorMessage(); }
        int16_t GetNextSample() { return refOnly ? n.GetNextSample() : h.GetNextSample(); }
        void LoadAudioStream(int c, const DCSDecoder::ROMPointer &p, int l) { if (refOnly) n.LoadAudioStream(c, p, l); else h.LoadAudioStream(c, p, l); }
        void ClearTracks() { if (refOnly) n.ClearTracks(); else h.ClearTracks(); }
        void SetMasterVolume(int v) { if (refOnly) n.SetMasterVolume(v); else h.SetMasterVolume(v); }
        bool IsStreamPlaying(int c) { return refOnly ? n.IsStreamPlaying(c) : h.IsStreamPlaying(c); }
    } hip{ refOnly, second, hipReal };
    if (lookahead >= 0)
        hip.SetLookahead(lookahead);
    ref.InitStandalone(kOs[os]);
    hip.InitStandalone(kOs[os]);
    const int volume0 = 0x40 + static_cast<int>(rng.below(0xC0));
    ref.SetDefaultVolume(volume0);
    hip.SetDefaultVolume(volume0);
    ref.SoftBoot();
    hip.SoftBoot();
    if (!hip.IsOK()) { fprintf(stderr, "decoder not OK: %s\n", hip.GetErrorMessage().c_str()); return 4; }

    uint64_t frames = 0, loads = 0, clears = 0, volumes = 0, asks = 0;
    auto pull = [&](uint32_t n) -> bool {
        for (uint32_t f = 0 ; f < n ; ++f, ++frames)
            for (int i = 0 ; i < 240 ; ++i)
            {
                const int16_t a = ref.GetNextSample(), b = hip.GetNextSample();
                if (a != b)
                {
                    printf("DIFFERENT at frame %llu sample %d: reference %d, hip %d (after %llu loads, %llu clears, %llu volume changes)\n",
                           static_cast<unsigned long long>(frames), i, a, b, static_cast<unsigned long long>(loads),
                           static_cast<unsigned long long>(clears), static_cast<unsigned long long>(volumes));
                    return false;
                }
            }
        return true;
    };
    for (int op = 0 ; op < nOps ; ++op)
    {
        const uint32_t kind = rng.below(100);
        if (kind < 30)
        {
            const int ch = static_cast<int>(rng.below(rng.below(4) == 0 ? 8 : 2));      // (mostly the channels callers use)
            const size_t k = rng.below(static_cast<uint32_t>(streams.size()));
            const int level = static_cast<int>(0x30 + rng.below(0x50));
            ref.LoadAudioStream(ch, DCSDecoder::ROMPointer(0, streams[k].data()), level);
            hip.LoadAudioStream(ch, DCSDecoder::ROMPointer(0, streams[k].data()), level);
            ++loads;
        }
        else if (kind < 36)
        {
            ref.ClearTracks();
            hip.ClearTracks();
            ++clears;
        }
        else if (kind < 42)
        {
            const int v = static_cast<int>(rng.below(256));
            ref.SetMasterVolume(v);
            hip.SetMasterVolume(v);
            ++volumes;
        }
        else if (kind < 60)
        {
            for (int ch = 0 ; ch < 8 ; ++ch, ++asks)
                if (ref.IsStreamPlaying(ch) != hip.IsStreamPlaying(ch))
                {
                    printf("DIFFERENT IsStreamPlaying(%d) behind frame %llu: reference %d, hip %d\n", ch, static_cast<unsigned long long>(frames),
                           ref.IsStreamPlaying(ch) ? 1 : 0, hip.IsStreamPlaying(ch) ? 1 : 0);
                    return 1;
                }
        }
        // frames in between: mostly a few, now and then a long quiet stretch (the look-ahead grows there)
        const uint32_t r = rng.below(100);
        const uint32_t n = r < 50 ? rng.below(4) : r < 85 ? rng.below(40) : r < 97 ? rng.below(400) : rng.below(2500);
        if (!pull(n))
            return 1;
        if (!ref.IsOK() || !hip.IsOK())
        {
            if (ref.IsOK() != hip.IsOK())
            {
                printf("DIFFERENT IsOK behind frame %llu: reference %d, hip %d (%s)\n", static_cast<unsigned long long>(frames),
                       ref.IsOK() ? 1 : 0, hip.IsOK() ? 1 : 0, hip.GetErrorMessage().c_str());
                return 1;
            }
            break;
        }
    }
    if (!pull(3))
        return 1;
    printf("ok: %llu frames, %d operations (%llu loads, %llu clears, %llu volume changes, %llu questions), every sample equal\n",
           static_cast<unsigned long long>(frames), nOps, static_cast<unsigned long long>(loads), static_cast<unsigned long long>(clears),
           static_cast<unsigned long long>(volumes), static_cast<unsigned long long>(asks));
    return 0;
}

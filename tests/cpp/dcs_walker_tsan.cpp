// dcs_walker_tsan -- the sequencer with its background walker under ThreadSanitizer (host only; tools/tsan_walker.sh): long streams planned
// while they are walked, a rewind, a second load, ClearTracks and destruction with a walk in progress.  The GPU entry points are stubs.
#include "../../include/dcs_hip.h"
#include <stdio.h>
#include <vector>
#include <string.h>
extern "C" DcsStatus dcs_decode_batch_live(DcsCtx *, const uint8_t *, size_t, uint64_t, const DcsSrcDesc *, uint32_t, const DcsFrameJob *, uint32_t,
                                           const int16_t *, uint32_t, const int16_t **, const uint32_t **, const int16_t **) { return DCS_ERR_NO_DEVICE; }
extern "C" DcsStatus dcs_decode_batch(DcsCtx *, const uint8_t *, size_t, const DcsSrcDesc *, uint32_t, const DcsFrameJob *, uint32_t,
                                      const int16_t *, uint32_t, int16_t *, uint32_t *, int16_t *) { return DCS_ERR_NO_DEVICE; }
extern "C" const char *dcs_last_error(const DcsCtx *) { return "stub"; }
int main()
{
    for (int fmt = 0 ; fmt < 6 ; ++fmt)
    {
        const int os = fmt == 2 ? DCS_OS93A : fmt < 2 ? DCS_OS93B : DCS_OS95;
        DcsSynthParams p = { uint64_t(900 + fmt), fmt, 1500, fmt == 2 ? 18 : 12, 16, fmt >= 3 ? 5 : 0, 0 };
        size_t len = 0;
        dcs_synth_stream(&p, nullptr, 0, &len);
        std::vector<uint8_t> s(len + 64), s2;
        dcs_synth_stream(&p, s.data(), len, &len);
        p.seed += 100; p.nFrames = 700;
        size_t len2 = 0;
        dcs_synth_stream(&p, nullptr, 0, &len2);
        s2.resize(len2 + 64);
        dcs_synth_stream(&p, s2.data(), len2, &len2);
        for (int rep = 0 ; rep < 3 ; ++rep)
        {
            DcsSequencer *q = dcs_seq_create_standalone(static_cast<DcsOsVersion>(os));
            dcs_seq_set_rewindable(q, 1);
            dcs_seq_load_audio_stream_mem(q, 0, s.data(), s.size(), 0x64);
            uint32_t total = 0, n = 0;
            for (int k = 0 ; k < 6 ; ++k) { dcs_seq_plan_ahead(q, k == 0 ? 64 : 150, 2, &n); total += n; }
            dcs_seq_rewind(q, total / 2);
            total /= 2;
            if (rep == 1) dcs_seq_clear_tracks(q);
            dcs_seq_load_audio_stream_mem(q, 1, s2.data(), s2.size(), 0x60);      // waits for the first walk
            for (int k = 0 ; k < 40 ; ++k) { dcs_seq_plan_ahead(q, 4096, 2, &n); total += n; if (!dcs_seq_stream_playing_at(q, total, 0) && !dcs_seq_stream_playing_at(q, total, 1)) break; }
            printf("fmt %d rep %d: %u ticks planned\n", fmt, rep, total);
            if (rep == 2) { dcs_seq_load_audio_stream_mem(q, 2, s.data(), s.size(), 0x50); }     // destroyed with a walk in progress
            dcs_seq_destroy(q);
        }
    }
    printf("done\n");
    return 0;
}

// Fixture generator (build container only): encodes one deterministic signal with the reference's own encoder
// (DCSEncoder::OpenStream / WriteStream / CloseStream, DCSEncoder.h:238-249) in the stream layouts it can produce.
//   enc_driver <formatVersion hex: 9400 | 9301 | 9302> <type 0|1> <subtype 0|3> <nFrames> <out.bin> [variant 0..3]
// (the variant moves the pitches and reseeds the noise: four different recordings of the same shape)
// The signal: 31 250 Hz mono -- a rising tone over a stack of harmonics, a chord with tremolo, a noise burst, near
// silence, a decaying low note -- so that frames visit loud, quiet, tonal and noisy band statistics at about the
// density of real material (110-130 bytes per frame).  Integer LCG for the noise.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "DCSEncoder.h"

int main(int argc, char **argv)
{
    if (argc != 6 && argc != 7) { fprintf(stderr, "usage: enc_driver <formatVersion> <type> <subtype> <nFrames> <out.bin> [variant]\n"); return 2; }
    const int variant = argc == 7 ? atoi(argv[6]) : 0;
    const double pitch = 1.0 + 0.19 * variant;
    const int nFrames = atoi(argv[4]);
    const size_t n = static_cast<size_t>(nFrames) * 240;
    std::vector<float> pcm(n);      // (the encoder's float entry point; its int16 one skips every other buffer slot, DCSEncoder.cpp:638-639)
    uint32_t lcg = 0x2545F491u + 0x9E3779B9u * static_cast<uint32_t>(variant);
    const double rate = 31250.0, pi = 3.14159265358979323846;
    for (size_t i = 0 ; i < n ; ++i)
    {
        const double t = static_cast<double>(i) / rate, u = static_cast<double>(i) / static_cast<double>(n);
        lcg = lcg * 1664525u + 1013904223u;
        const double noise = (static_cast<int32_t>(lcg >> 8) % 65536 - 32768) / 32768.0;
        // a sawtooth-like stack of harmonics (energy in every band, like instruments) under everything tonal
        double rich = 0;
        for (int h = 1 ; h <= 24 ; ++h)
            rich += sin(2 * pi * 173.0 * pitch * h * t + 0.7 * h) / h;
        double v;
        if (u < 0.25)       v = 0.40 * sin(2 * pi * pitch * (200.0 + 3800.0 * u * 4) * t) + 0.18 * rich + 0.16 * noise;      // rising tone over a chord
        else if (u < 0.50)  v = (0.30 * sin(2 * pi * 440.0 * pitch * t) + 0.25 * sin(2 * pi * 1320.0 * pitch * t) + 0.15 * rich) * (0.6 + 0.4 * sin(2 * pi * 6.0 * t)) + 0.14 * noise;
        else if (u < 0.65)  v = 0.40 * noise;                                                                          // noise burst
        else if (u < 0.75)  v = 0.0008 * noise;                                                                        // near silence
        else                v = 0.70 * exp(-(u - 0.75) * 12.0) * (sin(2 * pi * 110.0 * pitch * t) + 0.4 * rich) + 0.10 * noise; // decaying low note
        pcm[i] = static_cast<float>(lrint(v * 30000.0) / 32768.0);     // 16-bit sample values, as a WAV reader would deliver them
    }

    DCSEncoder enc;
    enc.compressionParams.formatVersion = static_cast<uint16_t>(strtoul(argv[1], nullptr, 16));
    enc.compressionParams.streamFormatType = atoi(argv[2]);
    enc.compressionParams.streamFormatSubType = atoi(argv[3]);
    std::string err;
    DCSEncoder::Stream *s = enc.OpenStream(31250, err);
    if (s == nullptr) { fprintf(stderr, "OpenStream: %s\n", err.c_str()); return 3; }
    enc.WriteStream(s, pcm.data(), pcm.size());
    DCSEncoder::DCSAudio obj;
    if (!enc.CloseStream(s, obj, err)) { fprintf(stderr, "CloseStream: %s\n", err.c_str()); return 4; }
    FILE *f = fopen(argv[5], "wb");
    fwrite(obj.data.get(), 1, obj.nBytes, f);
    fclose(f);
    fprintf(stderr, "%s type %s sub %s: %d frames, %zu bytes (%.1f B/frame)\n", argv[1], argv[2], argv[3], obj.nFrames, obj.nBytes,
            static_cast<double>(obj.nBytes) / obj.nFrames);
    return 0;
}

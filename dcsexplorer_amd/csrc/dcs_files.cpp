// dcs_files.cpp -- the data formats on the output side of the decode path: what `DCSExplorer
// --extract-streams` / `--extract-tracks` write and `--validate` logs.  Byte-for-byte the reference's
// layouts; plain buffers in and out, file I/O only in the two convenience writers.
#include "dcs_common.h"
#include <stdio.h>
#include <string.h>

static void putLE32(uint8_t *p, uint32_t v) { p[0] = v & 0xFF; p[1] = (v >> 8) & 0xFF; p[2] = (v >> 16) & 0xFF; p[3] = (v >> 24) & 0xFF; }
static void putLE16(uint8_t *p, uint32_t v) { p[0] = v & 0xFF; p[1] = (v >> 8) & 0xFF; }
static void putBE32(uint8_t *p, uint32_t v) { p[0] = (v >> 24) & 0xFF; p[1] = (v >> 16) & 0xFF; p[2] = (v >> 8) & 0xFF; p[3] = v & 0xFF; }

// The 44-byte header ExtractToWAV writes (DCSExplorer.cpp:1686-1699): mono, 16 bits, 31 250 samples/s,
// nFrames x 240 samples of data.
extern "C" void dcs_wav_header(uint32_t nFrames, uint8_t out[44])
{
    const uint32_t dataBytes = nFrames * DCS_FRAME_SAMPLES * 2;
    memset(out, 0, 44);
    memcpy(out, "RIFF", 4);
    putLE32(out + 4, dataBytes + 44 - 8);
    memcpy(out + 8, "WAVEfmt ", 8);
    putLE32(out + 16, 16);                  // fmt chunk length
    putLE16(out + 20, 1);                   // PCM
    putLE16(out + 22, 1);                   // channels
    putLE32(out + 24, 31250);               // samples per second
    putLE32(out + 28, 31250 * 16 / 8);      // bytes per second
    putLE16(out + 32, 2);                   // block align
    putLE16(out + 34, 16);                  // bits per sample
    memcpy(out + 36, "data", 4);
    putLE32(out + 40, dataBytes);
}

// The 36-byte "DCSa" raw-stream container header (DCSExplorer.cpp:1831-1866, DCSExplorer/README.md:274-289):
// signature, format version 0x9301 / 0x9302 / 0x9400, channels 1, rate 0x7A12, 22 reserved bytes, data size;
// all big-endian.  The data section is the stream's bytes, GetStreamInfo().nBytes of them.
extern "C" DcsStatus dcs_dcsa_header(DcsOsVersion os, uint32_t nBytes, uint8_t out[36])
{
    if (os < DCS_OS93A || os > DCS_OS95)
        return DCS_ERR_INVALID_ARG;
    memset(out, 0, 36);
    memcpy(out, "DCSa", 4);
    out[4] = (os == DCS_OS93A || os == DCS_OS93B) ? 0x93 : 0x94;
    out[5] = os == DCS_OS93A ? 0x01 : os == DCS_OS93B ? 0x02 : 0x00;
    out[6] = 0x00; out[7] = 0x01;
    out[8] = 0x7A; out[9] = 0x12;
    putBE32(out + 32, nBytes);
    return DCS_OK;
}

// The reader side, as DCSEncoder::IsDCSFile / EncodeDCSFile accept it (DCSEncoder.cpp:358-400, :432-470).
// osOut: DCS_OS93A / DCS_OS93B / DCS_OS94 (the container does not distinguish OS94 from OS95: same format).
extern "C" DcsStatus dcs_dcsa_parse(const uint8_t *file, size_t len, DcsOsVersion *osOut,
                                    const uint8_t **streamOut, uint32_t *nBytesOut)
{
    if (file == nullptr || len < 36)
        return DCS_ERR_BAD_STREAM;
    if (memcmp(file, "DCSa", 4) != 0 || (file[4] != 0x93 && file[4] != 0x94)
        || file[6] != 0 || file[7] != 1 || file[8] != 0x7A || file[9] != 0x12)
        return DCS_ERR_BAD_STREAM;
    const uint32_t nBytes = (static_cast<uint32_t>(file[32]) << 24) | (static_cast<uint32_t>(file[33]) << 16)
                          | (static_cast<uint32_t>(file[34]) << 8) | file[35];
    if (nBytes > len - 36)
        return DCS_ERR_BAD_STREAM;
    if (osOut != nullptr)
        *osOut = file[4] == 0x94 ? DCS_OS94 : file[5] == 0x01 ? DCS_OS93A : DCS_OS93B;
    if (streamOut != nullptr)
        *streamOut = file + 36;
    if (nBytesOut != nullptr)
        *nBytesOut = nBytes;
    return DCS_OK;
}

extern "C" DcsStatus dcs_write_wav(const char *path, const int16_t *pcm, uint32_t nFrames)
{
    if (path == nullptr || (pcm == nullptr && nFrames != 0))
        return DCS_ERR_INVALID_ARG;
    FILE *fp = fopen(path, "wb");
    if (fp == nullptr)
        return DCS_ERR_INVALID_ARG;
    uint8_t hdr[44];
    dcs_wav_header(nFrames, hdr);
    bool ok = fwrite(hdr, 44, 1, fp) == 1;
    // int16 little-endian on disk = the in-memory layout on every host this library runs on
    if (nFrames != 0)
        ok = ok && fwrite(pcm, sizeof(int16_t) * DCS_FRAME_SAMPLES, nFrames, fp) == nFrames;
    ok = (fclose(fp) == 0) && ok;
    return ok ? DCS_OK : DCS_ERR_INVALID_ARG;
}

extern "C" DcsStatus dcs_write_dcsa(const char *path, DcsOsVersion os, const uint8_t *stream, uint32_t nBytes)
{
    uint8_t hdr[36];
    if (path == nullptr || stream == nullptr || dcs_dcsa_header(os, nBytes, hdr) != DCS_OK)
        return DCS_ERR_INVALID_ARG;
    FILE *fp = fopen(path, "wb");
    if (fp == nullptr)
        return DCS_ERR_INVALID_ARG;
    bool ok = fwrite(hdr, 1, 36, fp) == 36 && fwrite(stream, 1, nBytes, fp) == nBytes;
    ok = (fclose(fp) == 0) && ok;
    return ok ? DCS_OK : DCS_ERR_INVALID_ARG;
}

// One frame of the --validate log (DCSExplorer.cpp:1358-1447): counts the differing samples and, if there
// are any and `text` is given, formats the block the reference writes to its log file -- a title line, 15
// lines of 16 samples of `mine` | 16 samples of `theirs`, a blank line.  Returns the number of differing
// samples; *textLen receives the length of the text (0 when the frames agree).
extern "C" int dcs_frame_diff(uint64_t frameNo, const int16_t *mine, const int16_t *theirs,
                              char *text, size_t textCap, size_t *textLen)
{
    int nDiffs = 0;
    for (int i = 0 ; i < DCS_FRAME_SAMPLES ; ++i)
        nDiffs += mine[i] != theirs[i];
    size_t used = 0;
    if (nDiffs != 0 && text != nullptr && textCap != 0)
    {
        auto putStr = [&](const char *str) {
            for ( ; *str != 0 && used + 1 < textCap ; ++str)
                text[used++] = *str;
            text[used] = 0;
        };
        char num[96];
        snprintf(num, sizeof(num), "--- Frame %llu - %d sample differences ---\n", static_cast<unsigned long long>(frameNo), nDiffs);
        putStr(num);
        for (int i = 0 ; i < DCS_FRAME_SAMPLES ; i += 16)
        {
            for (int k = 0 ; k < 16 ; ++k)
            {
                snprintf(num, sizeof(num), k == 0 ? "%6d" : " %6d", mine[i + k]);
                putStr(num);
            }
            putStr(" |");
            for (int k = 0 ; k < 16 ; ++k)
            {
                snprintf(num, sizeof(num), " %6d", theirs[i + k]);
                putStr(num);
            }
            putStr("\n");
        }
        putStr("\n");
    }
    if (textLen != nullptr)
        *textLen = used;
    return nDiffs;
}

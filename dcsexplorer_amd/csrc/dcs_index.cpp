// dcs_index.cpp -- host index pass: walks a DCS stream and records, for every frame, the carried
// state that makes it independently decodable (bit offset, band-type codes).
//
// Replaces DCSDecoderNative::GetStreamInfo (DCSDecoderNative.cpp:1486-1537), which finds the end of
// a stream the same way: by running the frame decompressor over every frame.  This walker follows
// only the LENGTHS of the coded fields (it never reconstructs a sample, so it cannot serve as a CPU
// decode path); the layouts it parses are those of DecoderImpl94x/93/93a::DecompressFrame
// (:1679-2261, :2293-2684, :2831-3032) and the container of InitChannelStream (:1433-1463).
#include "dcs_common.h"
#include <string.h>

namespace {

// MSB-first reader with the reference's look-ahead policy (ROMBitPointer, DCSDecoderNative.h:229-289):
// Peek(n) pulls whole bytes while nBits <= n.  The policy matters only for StreamInfo.nBytes, which
// the reference computes from the reader's BYTE pointer (:1509).
struct Bits
{
    const uint8_t *data = nullptr;
    size_t len = 0;
    size_t payOff = 0;
    size_t p = 0;
    uint32_t buf = 0;
    int nBits = 0;

    uint32_t byteAt(size_t i) const { return i < len ? data[i] : 0u; }
    uint32_t peek(int n)
    {
        while (nBits <= n)
        {
            buf |= byteAt(p++) << (24 - nBits);
            nBits += 8;
        }
        return buf >> (32 - n);
    }
    uint32_t get(int n)
    {
        uint32_t r = peek(n);
        nBits -= n;
        buf <<= n;
        return r;
    }
    uint32_t bitPos() const { return static_cast<uint32_t>((p - payOff) * 8 - static_cast<size_t>(nBits)); }
};

// one prefix code through the trie of dcs_common.h, bit-serial like the reference's tree walks
// (:1819-1828, :2653-2661) so the reader's look-ahead -- and therefore nBytes -- is the same
inline int readVlc(Bits &b, const uint16_t *trie)
{
    uint32_t e = trie[b.get(1)];
    while (!(e & 0x8000))
        e = trie[e + b.get(1)];
    return static_cast<int>(e & 0xFF);
}

struct Scan
{
    Bits b;
    uint8_t header[16];
    uint16_t bandType[16];
    uint32_t err = 0;
};

inline void fatal(Scan &s) { s.err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; }

// --- 1994+ frame (:1679-2261) -----------------------------------------------------------------
void scan94(Scan &s)
{
    const DcsLdsTables &T = dcsTables().lds;
    const uint8_t *hdr = s.header;
    const bool type1 = (hdr[0] & 0x80) != 0;

    // Type 1 indexes its pre-adjust map with the previous frame's codes of bands 0..2 (:1771-1773)
    if (type1)
        for (int i = 0 ; i < 3 ; ++i)
            if (s.bandType[i] > 15) { fatal(s); return; }

    // frame header: one delta code per populated band (:1780-1834)
    for (int i = 0 ; i < 16 && (hdr[i] & 0x7F) != 0x7F ; ++i)
        s.bandType[i] = static_cast<uint16_t>(s.bandType[i] + readVlc(s.b, T.trie94) - 16);

    for (int band = 0 ; band < 16 ; ++band)
    {
        const int hb = hdr[band] & 0x7F;
        if (hb == 0x7F)
            break;
        int count = band == 0 ? 7 : band == 1 ? 8 : band == 15 ? 32 : 16;      // :1848-1850
        if (hb & 0x40)
            count /= 2;
        int code = s.bandType[band];
        if (code == 0)
            continue;
        if (type1)
        {
            if (code > 15) { fatal(s); return; }
            code = T.xlat94[(band < 3 ? 0 : band < 6 ? 16 : 32) + code] & 0xFF;
        }
        if (code == 0)
        {
            s.err |= DCS_FRAME_STOP;                    // :1985-1991, consumes nothing
        }
        else if (code <= 6)
        {
            const int maxBits = T.cbInfo[code] & 0xF;
            const uint16_t *book = T.cb94 + (T.cbInfo[code] >> 4);
            for (int i = count ; i != 0 ; --i)
            {
                const uint32_t e = book[s.b.peek(maxBits)];
                s.b.get(static_cast<int>(e >> 8));
                if (e & 0x80)
                {
                    if (i >= 2) --i;
                    else { s.err |= DCS_FRAME_STOP; i = 1; }        // :2213-2218
                }
            }
        }
        else
        {
            if (code > 16) { fatal(s); return; }
            for (int i = 0 ; i < count ; ++i)
                s.b.get(code);
        }
    }
}

// --- 1993 frame, Type 0 and OS93b Type 1 (:2293-2615) --------------------------------------------
void scan93(Scan &s)
{
    const DcsLdsTables &T = dcsTables().lds;
    const bool type1 = (s.header[0] & 0x80) != 0;
    bool first = true, reuse = false;
    int code = 0;

    for (int band = 0 ; band < 16 ; ++band)
    {
        const int hb = s.header[band] & 0x7F;
        if (hb == 0x7F)
            break;
        const bool strided = (hb >> 6) != 0;
        const int nSamples = !type1 ? 16 : strided ? 8 : first ? 15 : 16;       // :2351-2383

        if (reuse)
            reuse = s.b.get(1) != 0;
        if (!reuse)
        {
            if (!type1)
            {
                if (s.b.get(1))
                    s.b.get(1);                                 // sub-type step direction (:2402-2414)
                code = static_cast<int>(s.b.get(4));
            }
            else
            {
                int v = readVlc(s.b, T.trie93);
                v = v < 0x1E ? v - 0x0F : v - 0x2E;             // :2668-2681
                s.bandType[band] = static_cast<uint16_t>(s.bandType[band] + v);
                code = s.bandType[band];
            }
        }

        if (code == 0)
            reuse = true;                                       // :2455
        else
        {
            const int width = code + (type1 ? 0 : 1);
            if (width > 16) { fatal(s); return; }
            for (int i = 0 ; i < nSamples ; ++i)
                s.b.get(width);
        }
        first = false;
    }
}

// --- OS93a Type 1 frame (:2831-3032) ---------------------------------------------------------------
void scan93a(Scan &s)
{
    const DcsLdsTables &T = dcsTables().lds;
    const int hb = s.header[0];
    const uint16_t *bbBook = &T.bandBits93a[(hb & 0x60) >> 1];
    const int numBands = hb & 0x1F;

    for (int band = 0 ; band < numBands ; ++band)
    {
        if (band >= 18) { fatal(s); return; }
        const uint32_t e = bbBook[s.b.peek(4)];
        s.b.get(static_cast<int>(e >> 8));
        const int bandBits = static_cast<int>(e & 0xFF);
        if (bandBits == 0xFF)
            break;
        if (bandBits == 0)
            continue;
        uint32_t sc = T.scaleCb93a[s.b.peek(4)];
        s.b.get(static_cast<int>((sc >> 8) & 0xF));
        if ((sc & 0xFF) == 0xFF)
        {
            sc = T.scaleCb93a[((sc >> 12) << 4) + s.b.peek(4)];
            s.b.get(static_cast<int>((sc >> 8) & 0xF) - 4);
        }
        for (int i = 0 ; i < T.inputs93a[band] ; ++i)
            s.b.get(bandBits);
    }
}

}   // namespace

extern "C" DcsStatus dcs_index_stream(DcsOsVersion os, const uint8_t *stream, size_t len,
                                      DcsFrameIndex *out, uint32_t cap, DcsStreamInfo *info)
{
    if (stream == nullptr || len < 3 || os < DCS_OS93A || os > DCS_OS95 || (out == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;

    Scan s;
    s.b.data = stream;
    s.b.len = len;

    // container (InitChannelStream :1433-1463, InitStreamPlayback :1595-1641)
    const int nFrames = (stream[0] << 8) | stream[1];
    const bool typeBit = (stream[2] & 0x80) != 0;
    const int hdrLen = (os == DCS_OS93A && typeBit) ? 1 : 16;
    memset(s.header, 0, sizeof(s.header));
    for (int i = 0 ; i < hdrLen ; ++i)
        s.header[i] = static_cast<uint8_t>(s.b.byteAt(2 + static_cast<size_t>(i)));
    memset(s.bandType, 0, sizeof(s.bandType));
    s.b.payOff = s.b.p = 2 + static_cast<size_t>(hdrLen);

    int format;
    if (os == DCS_OS93A)
        format = typeBit ? DCS_FMT_93A_T1 : DCS_FMT_93_T0;
    else if (os == DCS_OS93B)
        format = typeBit ? DCS_FMT_93B_T1 : DCS_FMT_93_T0;
    else if (!typeBit)
        format = DCS_FMT_94_T0;
    else
        format = (((s.header[1] | s.header[2]) & 0x80) == 0) ? DCS_FMT_94_T1_S0 : DCS_FMT_94_T1_S3;

    DcsStreamInfo si;
    memset(&si, 0, sizeof(si));
    si.nFrames = nFrames;
    si.formatType = typeBit ? 1 : 0;
    if (os == DCS_OS94 || os == DCS_OS95)       // sic: GetStreamInfo tests header[1] twice (:1517)
        si.formatSubType = ((s.header[1] & 0x80) >> 6) | ((s.header[1] & 0x80) >> 7);
    memcpy(si.header, s.header, 16);
    si.format = format;
    si.hdrLen = hdrLen;

    DcsStatus status = DCS_OK;
    int valid = 0;
    for (int f = 0 ; f < nFrames ; ++f)
    {
        DcsFrameIndex fi;
        fi.bitOff = s.b.bitPos();
        memcpy(fi.bandType, s.bandType, sizeof(fi.bandType));
        s.err = 0;
        switch (format)
        {
        case DCS_FMT_93_T0:
        case DCS_FMT_93B_T1: scan93(s); break;
        case DCS_FMT_93A_T1: scan93a(s); break;
        default:             scan94(s); break;
        }
        fi.nBits = s.b.bitPos() - fi.bitOff;
        fi.err = s.err;
        if (static_cast<uint32_t>(valid) < cap)
            out[valid] = fi;
        else if (out != nullptr || cap != 0)
            status = DCS_ERR_CAPACITY;
        ++valid;
        si.payloadBits = s.b.bitPos();
        if (s.err != 0)
            break;                  // the reference stops the channel on the next tick (:95-116)
    }
    si.nValidFrames = valid;
    si.nBytes = static_cast<int32_t>(s.b.p);
    if (out == nullptr && cap == 0)
        status = DCS_OK;
    if (info != nullptr)
        *info = si;
    if (nFrames == 0)
        return DCS_ERR_BAD_STREAM;
    return status;
}

// dcs_scan.h -- the index pass: walks a DCS stream and records, for every frame, the carried state that
// makes it independently decodable (DcsFrameIndex: bit offset, band-type codes, split records).
//
// Compiled for the host (dcs_index.cpp, dcs_index_stream: the product's host index pass and the checker of the
// device's).  The device walks a stream with a whole wavefront (dcs_index_wave.hip.h: the same parse, its Huffman runs
// and frame headers taken 64 candidate positions at a time) and must produce identical records (tests/, tools/fuzz_parity.py).
// It replaces the serial walk of DCSDecoderNative::GetStreamInfo (DCSDecoderNative.cpp:1486-1537), which
// finds the end of a stream the same way: by running the frame decompressor over every frame.  This
// walker follows the LENGTHS of the coded fields (plus, for the 1993 formats, the two carried sample
// words); it never reconstructs a spectrum or a PCM sample, so it cannot serve as a CPU decode path.
// The layouts it parses are those of DecoderImpl94x/93/93a::DecompressFrame (:1679-2261, :2293-2684,
// :2831-3032) and the container of InitChannelStream (:1433-1463).
#pragma once
#include "dcs_common.h"
#include <string.h>

#ifdef __HIPCC__
#define DCS_HD __host__ __device__
#else
#define DCS_HD
#endif

// (host only) several 1994+ sample codes per look: entry [code - 1][next DCS_MULTI_BITS bits] = total length of the
// codes that lie entirely inside those bits, look-ahead of the last one included | sample positions they stand for << 8
#define DCS_MULTI_BITS 14
struct DcsScanTables
{
    const DcsLdsTables *lds;
    const uint16_t *trie94;
    const uint16_t *multi94 = nullptr;      // [6][1 << DCS_MULTI_BITS], or null (the device walk: no room in LDS)
    const uint16_t *fast94 = nullptr;       // first-level table of the 1994+ band-type delta code (dcs_common.h), or null
};

// MSB-first reader with the reference's look-ahead policy (ROMBitPointer, DCSDecoderNative.h:229-289):
// Peek(n) pulls whole bytes while nBits <= n.  The policy matters only for StreamInfo.nBytes, which
// the reference computes from the reader's BYTE pointer (:1509).  Fetch(i) returns byte i of the stream
// (0 past its end).
template <class Fetch>
struct DcsBits
{
    static constexpr bool kAnalytic = false;    // the byte pointer is kept, look by look
    Fetch fetch;
    size_t payOff = 0;
    size_t p = 0;
    uint32_t buf = 0;
    int nBits = 0;

    DCS_HD uint32_t byteAt(size_t i) { return fetch(i); }
    DCS_HD void setPayload(size_t off) { payOff = p = off; }
    DCS_HD size_t bytesFetched() const { return p; }
    DCS_HD uint32_t peek(int n)
    {
        while (nBits <= n)
        {
            buf |= fetch(p++) << (24 - nBits);
            nBits += 8;
        }
        return buf >> (32 - n);
    }
    DCS_HD uint32_t get(int n)
    {
        uint32_t r = peek(n);
        nBits -= n;
        buf <<= n;
        return r;
    }
    // n bits of what a peek(m >= n) at this position has just looked at (that peek has pulled every byte a get(n) would)
    DCS_HD void consume(int n)
    {
        nBits -= n;
        buf <<= n;
    }
    // (a look that must not move the byte pointer is not something this literal reader can do: it is not used with the
    // multi-code table)
    DCS_HD uint32_t look(int n) { return peek(n); }
    // `count` fields of `width` bits whose values nobody needs: read one by one, the byte pointer must end up where the
    // reference's does (the device reader, which computes that pointer instead of keeping it, steps over them at once)
    DCS_HD void skipRun(int count, int width)
    {
        for (int i = 0 ; i < count ; ++i)
            get(width);
    }
    DCS_HD uint32_t bitPos() const { return static_cast<uint32_t>((p - payOff) * 8 - static_cast<size_t>(nBits)); }
};

// working storage of one walk (the host keeps it on the stack, the index kernel in LDS: the arrays
// are indexed with run-time band numbers, which registers cannot do)
struct DcsScanMem
{
    uint8_t header[16];
    uint16_t bandType[16];
    DcsFrameIndex fi;
};

template <class R>
struct DcsScan
{
    R b;
    uint8_t *header;
    uint16_t *bandType;
    int nBands = 0;
    uint32_t err = 0;
};

// one prefix code through the trie of dcs_common.h, bit-serial like the reference's tree walks
// (:1819-1828, :2653-2661) so the reader's look-ahead -- and therefore nBytes -- is the same
template <class R>
DCS_HD inline int dcsReadVlc(R &b, const uint16_t *trie)
{
    uint32_t e = trie[b.get(1)];
    while (!(e & 0x8000))
        e = trie[e + b.get(1)];
    return static_cast<int>(e & 0xFF);
}

// The same through the first-level table: eight bits at a look, the rest of a longer code bit by bit.  For readers that
// COMPUTE the reference reader's byte pointer (R::kAnalytic: WinBits, DevBits): the tree walk's looks, one bit each, reach
// as far as the code's last bit, which is what took() records.  The literal reader keeps the walk above.
template <class R>
DCS_HD inline int dcsReadVlcFast(R &b, const uint16_t *fast, const uint16_t *trie)
{
    if constexpr (R::kAnalytic)
    {
        if (fast != nullptr)
        {
            uint32_t e = fast[b.look(8)];
            if (e & 0x8000)
            {
                b.took(static_cast<int>((e >> 8) & 0xF));
                return static_cast<int>(e & 0xFF);
            }
            b.took(8);
            do
                e = trie[e + b.get(1)];
            while (!(e & 0x8000));
            return static_cast<int>(e & 0xFF);
        }
    }
    return dcsReadVlc(b, trie);
}

template <class R>
DCS_HD inline void dcsFatal(DcsScan<R> &s) { s.err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; }

// split[band - 1] = the decoder state at the start of `band` (1..15)
template <class R>
DCS_HD inline void dcsPutSplit(DcsFrameIndex &fi, int band, uint32_t frameStart, const DcsScan<R> &s, int outIdx,
                               uint32_t prv = 0, uint32_t prvDelta = 0, int subType = 0, bool reuse = false)
{
    if (band < 1 || band > 15)
        return;
    DcsSplit &sp = fi.split[band - 1];
    sp.bitDelta = static_cast<uint16_t>(s.b.bitPos() - frameStart);
    sp.prv = static_cast<uint16_t>(prv);
    sp.prvDelta = static_cast<uint16_t>(prvDelta);
    sp.state = static_cast<uint16_t>((outIdx & 0x1FF) | (subType << 9) | (reuse ? 0x800 : 0));
}

// 1994+: the decoder state in the MIDDLE of band 15 (32 samples, twice any other band), kept in the two fields of that
// band's split record that only the 1993 formats use: prv = bits from the frame's first bit to the first code that starts
// with at least half of the band's samples done, prvDelta = output index there | DCS_MID15_STRADDLE when a two-zeros
// code carried one sample across the middle.  prv == 0: no such point (the band is empty or the frame ended before it).
template <class R>
DCS_HD inline void dcsPutMid15(DcsFrameIndex &fi, uint32_t frameStart, const DcsScan<R> &s, int outIdx, bool straddle)
{
    DcsSplit &sp = fi.split[14];
    sp.prv = static_cast<uint16_t>(s.b.bitPos() - frameStart);
    sp.prvDelta = static_cast<uint16_t>((outIdx & 0x1FF) | (straddle ? DCS_MID15_STRADDLE : 0u));
}

// --- 1994+ frame (:1679-2261) -----------------------------------------------------------------
template <class R>
DCS_HD void dcsScan94(DcsScan<R> &s, const DcsScanTables &tabs, DcsFrameIndex &fi)
{
    const DcsLdsTables &T = *tabs.lds;
    const uint16_t *trie94 = tabs.trie94;
    const uint8_t *hdr = s.header;
    const bool type1 = (hdr[0] & 0x80) != 0;
    const bool sub0 = ((hdr[1] | hdr[2]) & 0x80) == 0;
    const uint32_t frameStart = s.b.bitPos();

    // Type 1 indexes its pre-adjust map with the previous frame's codes of bands 0..2 (:1744-1773)
    if (type1)
    {
        for (int i = 0 ; i < 3 ; ++i)
            if (s.bandType[i] > 15) { dcsFatal(s); return; }
        const uint8_t *map = T.preAdj94 + (sub0 ? 0 : 16);
        fi.preAdj = static_cast<uint16_t>(map[s.bandType[0]] | (map[s.bandType[1]] << 4) | (map[s.bandType[2]] << 8));
    }

    // frame header: one delta code per populated band (:1780-1834)
    for (int i = 0 ; i < s.nBands ; ++i)
        s.bandType[i] = static_cast<uint16_t>(s.bandType[i] + dcsReadVlcFast(s.b, tabs.fast94, trie94) - 16);
    fi.hdrBits = static_cast<uint16_t>(s.b.bitPos() - frameStart);
    for (int i = 0 ; i < 16 ; ++i)
        fi.bandType[i] = static_cast<uint8_t>(s.bandType[i] > 255 ? 255 : s.bandType[i]);

    int outIdx = 1;
    for (int band = 0 ; band < s.nBands ; ++band)
    {
        dcsPutSplit(fi, band, frameStart, s, outIdx);
        const int hb = hdr[band] & 0x7F;
        int count = band == 0 ? 7 : band == 1 ? 8 : band == 15 ? 32 : 16;      // :1848-1850
        int inc = 1;
        if (hb & 0x40) { count /= 2; inc = 2; }
        int code = s.bandType[band];
        if (code == 0)
        {
            outIdx += count;                            // the halved count (:1886)
            continue;
        }
        if (type1)
        {
            if (code > 15) { dcsFatal(s); return; }
            code = T.xlat94[(band < 3 ? 0 : band < 6 ? 16 : 32) + code] & 0xFF;
        }
        if (code > 16) { dcsFatal(s); return; }
        outIdx += count * inc;
        if (code == 0)
        {
            s.err |= DCS_FRAME_STOP;                    // :1985-1991, consumes nothing
        }
        else if (code <= 6)
        {
            const int maxBits = T.cbInfo[code] & 0xF;
            const uint16_t *book = T.cb94 + (T.cbInfo[code] >> 4);
            int i = count;
            // Band 15 is twice as long as the others: its samples are walked in two halves, and where the second one
            // starts -- the first code boundary with at least half of the samples done -- is recorded, so that two lanes
            // can share the band (dcsPutMid15).  Every other band is one piece.
            for (int piece = band == 15 ? 0 : 1 ; piece < 2 ; ++piece)
            {
                const int lim = piece == 0 ? count / 2 : 0;         // samples left when the piece is done
                // several codes per look while more samples remain than the look covers (so that the "two zeros with one
                // slot left" case, and every band's last look -- the one that can decide nBytes --, stay with the loop below)
                if (tabs.multi94 != nullptr)
                {
                    const uint16_t *multi = tabs.multi94 + (static_cast<size_t>(code - 1) << DCS_MULTI_BITS);
                    for (;;)
                    {
                        const uint32_t e = multi[s.b.look(DCS_MULTI_BITS)];
                        const int steps = static_cast<int>(e >> 8);
                        if (steps == 0 || steps >= i - lim)
                            break;
                        s.b.consume(static_cast<int>(e & 0xFF));
                        i -= steps;
                    }
                }
                for ( ; i > lim ; --i)
                {
                    const uint32_t e = book[s.b.peek(maxBits)];
                    s.b.consume(static_cast<int>((e >> 8) & 0x1F));     // (a code is never longer than its book's look-ahead)
                    if ((e >> 13) == 2)
                    {
                        if (i >= 2) --i;
                        else { s.err |= DCS_FRAME_STOP; i = 1; }        // :2213-2218
                    }
                }
                if (piece == 0)
                    dcsPutMid15(fi, frameStart, s, outIdx - i * inc, i < lim);
            }
        }
        else if (band == 15)
        {
            s.b.skipRun(count - count / 2, code);
            dcsPutMid15(fi, frameStart, s, outIdx - (count / 2) * inc, false);
            s.b.skipRun(count / 2, code);
        }
        else
        {
            s.b.skipRun(count, code);
        }
    }
}

// --- 1993 frame, Type 0 and OS93b Type 1 (:2293-2615) --------------------------------------------
template <class R>
DCS_HD void dcsScan93(DcsScan<R> &s, const DcsScanTables &tabs, DcsFrameIndex &fi)
{
    const DcsLdsTables &T = *tabs.lds;
    const bool type1 = (s.header[0] & 0x80) != 0;
    const uint32_t frameStart = s.b.bitPos();
    bool first = true, reuse = false;
    int code = 0;
    int subType = type1 ? 0 : 2;
    uint32_t prv = 0, prvDelta = 0;
    int outIdx = 1;

    for (int i = 0 ; i < 16 ; ++i)
        fi.bandType[i] = static_cast<uint8_t>(s.bandType[i] > 255 ? 255 : s.bandType[i]);

    for (int band = 0 ; band < s.nBands ; ++band)
    {
        dcsPutSplit(fi, band, frameStart, s, outIdx, prv, prvDelta, subType, reuse);
        const int hb = s.header[band] & 0x7F;
        const bool strided = (hb >> 6) != 0;
        int nSamples, inc = 1, fixup = 0, stride;
        if (!type1)
        {
            nSamples = 16;
            if (!strided) stride = 16;
            else { ++outIdx; inc = 2; fixup = -1; stride = 31; }
        }
        else
        {
            if (!strided) nSamples = stride = first ? 15 : 16;
            else { inc = 2; nSamples = stride = 8; }
        }

        if (reuse)
            reuse = s.b.get(1) != 0;
        if (!reuse)
        {
            if (!type1)
            {
                if (s.b.get(1))
                    subType = s.b.get(1) ? (subType + 1) % 3 : (subType + 2) % 3;   // :2402-2414
                code = static_cast<int>(s.b.get(4));
            }
            else
            {
                int v = dcsReadVlcFast(s.b, T.fast93, T.trie93);
                if (v < 0x1E)
                    v -= 0x0F;                                      // :2668-2681
                else
                {
                    v -= 0x2E;
                    subType = subType != 0 ? 0 : 1;
                }
                s.bandType[band] = static_cast<uint16_t>(s.bandType[band] + v);
                code = s.bandType[band];
            }
        }

        if (code == 0)
        {
            reuse = true;                                           // :2455
            if (subType == 0) { outIdx += stride; prv = 0; prvDelta = 0; }
            else if (subType == 1) { prvDelta = 0; outIdx += nSamples * inc + fixup; }
            else
            {
                for (int i = 0 ; i < nSamples ; ++i)
                    prv = (prv + prvDelta) & 0xFFFF;
                outIdx += nSamples * inc + fixup;
            }
        }
        else
        {
            const int width = code + (type1 ? 0 : 1);
            if (width > 16) { dcsFatal(s); return; }
            // the sample values are needed only for the carried (prv, prvDelta) pair (:2565-2599)
            uint32_t last = 0, last2 = 0;
            // (directly coded samples: only the last two are carried on)
            const int skipped = (subType == 0 && nSamples > 2) ? nSamples - 2 : 0;
            s.b.skipRun(skipped, width);
            for (int i = skipped ; i < nSamples ; ++i)
            {
                uint32_t in = s.b.get(width);
                if (in & (1u << (width - 1)))
                    in |= 0xFFFFFFFFu << width;
                in &= 0xFFFF;
                if (subType == 0) { last2 = last; last = in; }
                else
                {
                    prvDelta = subType == 1 ? in : ((prvDelta + in) & 0xFFFF);
                    prv = (prv + prvDelta) & 0xFFFF;
                }
            }
            if (subType == 0)
            {
                prv = last;
                prvDelta = (last - last2) & 0xFFFF;
            }
            outIdx += nSamples * inc + fixup;
        }
        first = false;
    }
}

// --- OS93a Type 1 frame (:2831-3032) ---------------------------------------------------------------
// Carried from band to band: the bit position and the previous band's scale code (:2962-2975).  A split
// record holds them in (bitDelta, prv); its "reuse" bit says the frame already ended before this band
// (the 0xFF band-bits code, :2905-2912).
template <class R>
DCS_HD void dcsScan93a(DcsScan<R> &s, const DcsScanTables &tabs, DcsFrameIndex &fi)
{
    const DcsLdsTables &T = *tabs.lds;
    const int hb = s.header[0];
    const uint16_t *bbBook = &T.bandBits93a[(hb & 0x60) >> 1];
    const int numBands = hb & 0x1F;
    const uint32_t frameStart = s.b.bitPos();
    int prvScale = 0x1A;
    int outIdx = 0;
    bool ended = false;

    // An OS93a Type-1 frame has up to 18 bands but no band-type codes: the 16 bandType bytes of its record hold the
    // split records of bands 16 and 17 (same layout as DcsSplit), so that two more lanes can share the frame's tail
    for (int i = 0 ; i < 16 ; ++i)
        fi.bandType[i] = 0;
    for (int band = 0 ; band < numBands ; ++band)
    {
        dcsPutSplit(fi, band, frameStart, s, outIdx, static_cast<uint32_t>(prvScale), 0, 0, ended);
        if (band == 16 || band == 17)
        {
            uint8_t *rec = fi.bandType + (band - 16) * 8;
            const uint32_t bitDelta = s.b.bitPos() - frameStart;
            const uint32_t state = static_cast<uint32_t>(outIdx & 0x1FF) | (ended ? 0x800u : 0u);
            rec[0] = static_cast<uint8_t>(bitDelta); rec[1] = static_cast<uint8_t>(bitDelta >> 8);
            rec[2] = static_cast<uint8_t>(prvScale); rec[3] = static_cast<uint8_t>(static_cast<uint32_t>(prvScale) >> 8);
            rec[4] = 0; rec[5] = 0;
            rec[6] = static_cast<uint8_t>(state); rec[7] = static_cast<uint8_t>(state >> 8);
        }
        if (ended)
            continue;
        if (band >= 18) { dcsFatal(s); return; }
        const int numInputs = T.inputs93a[band];
        const uint32_t e = bbBook[s.b.peek(4)];
        s.b.get(static_cast<int>(e >> 8));
        const int bandBits = static_cast<int>(e & 0xFF);
        if (bandBits == 0xFF)
        {
            ended = true;
            continue;
        }
        outIdx += numInputs * 2;
        if (bandBits == 0)
            continue;
        uint32_t sc = T.scaleCb93a[s.b.peek(4)];
        s.b.get(static_cast<int>((sc >> 8) & 0xF));
        if ((sc & 0xFF) == 0xFF)
        {
            sc = T.scaleCb93a[((sc >> 12) << 4) + s.b.peek(4)];
            s.b.get(static_cast<int>((sc >> 8) & 0xF) - 4);
        }
        int scaleCode = prvScale + static_cast<int>(sc & 0xFF) - 1 + bandBits * 2;
        if (scaleCode > 0x39)
            scaleCode -= 0x36;
        prvScale = scaleCode - bandBits * 2;
        s.b.skipRun(numInputs, bandBits);
    }
}


// Walk one stream.  `sink(f, record)` receives every indexed frame.  Returns the stream summary.
template <class R, class Sink>
DCS_HD DcsStreamInfo dcsScanStream(int os, R reader, const DcsScanTables &tabs, DcsScanMem *mem, Sink &sink)
{
    DcsScan<R> s{ reader, mem->header, mem->bandType };

    // container (InitChannelStream :1433-1463, InitStreamPlayback :1595-1641)
    const int nFrames = static_cast<int>((s.b.byteAt(0) << 8) | s.b.byteAt(1));
    const bool typeBit = (s.b.byteAt(2) & 0x80) != 0;
    const int hdrLen = (os == DCS_OS93A && typeBit) ? 1 : 16;
    for (int i = 0 ; i < 16 ; ++i)
    {
        s.header[i] = i < hdrLen ? static_cast<uint8_t>(s.b.byteAt(2 + static_cast<size_t>(i))) : static_cast<uint8_t>(0);
        s.bandType[i] = 0;
    }
    s.b.setPayload(2 + static_cast<size_t>(hdrLen));
    s.nBands = 0;
    if (os == DCS_OS93A && typeBit)
        s.nBands = s.header[0] & 0x1F;
    else
        while (s.nBands < 16 && (s.header[s.nBands] & 0x7F) != 0x7F)
            ++s.nBands;

    int format;
    if (os == DCS_OS93A)
        format = typeBit ? DCS_FMT_93A_T1 : DCS_FMT_93_T0;
    else if (os == DCS_OS93B)
        format = typeBit ? DCS_FMT_93B_T1 : DCS_FMT_93_T0;
    else if (!typeBit)
        format = DCS_FMT_94_T0;
    else
        format = (((s.header[1] | s.header[2]) & 0x80) == 0) ? DCS_FMT_94_T1_S0 : DCS_FMT_94_T1_S3;

    DcsStreamInfo si = {};
    si.nFrames = nFrames;
    si.formatType = typeBit ? 1 : 0;
    if (os == DCS_OS94 || os == DCS_OS95)       // sic: GetStreamInfo tests header[1] twice (:1517)
        si.formatSubType = ((s.header[1] & 0x80) >> 6) | ((s.header[1] & 0x80) >> 7);
    for (int i = 0 ; i < 16 ; ++i)
        si.header[i] = s.header[i];
    si.format = format;
    si.hdrLen = hdrLen;

    int valid = 0;
    for (int f = 0 ; f < nFrames ; ++f)
    {
        DcsFrameIndex &fi = mem->fi;
        fi = DcsFrameIndex{};
        const uint32_t frameBit = s.b.bitPos();
        fi.bitOff = frameBit;
        fi.nBands = static_cast<uint8_t>(s.nBands);
        s.err = 0;
        switch (format)
        {
        case DCS_FMT_93_T0:
        case DCS_FMT_93B_T1: dcsScan93(s, tabs, fi); break;
        case DCS_FMT_93A_T1: dcsScan93a(s, tabs, fi); break;
        default:             dcsScan94(s, tabs, fi); break;
        }
        fi.nBits = static_cast<uint16_t>(s.b.bitPos() - frameBit);
        fi.flags = static_cast<uint8_t>((s.err << 4) | (s.err != 0 ? DCS_IDX_SERIAL : 0));
        sink(static_cast<uint32_t>(valid), fi);
        ++valid;
        si.payloadBits = s.b.bitPos();
        if (s.err != 0)
            break;                  // the reference stops the channel on the next tick (:95-116)
    }
    si.nValidFrames = valid;
    si.nBytes = static_cast<int32_t>(s.b.bytesFetched());
    return si;
}

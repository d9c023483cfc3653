// dcs_kernels.hip.h -- CDNA4 (gfx950) kernels for batched DCS frame decode.
//
// One wavefront (= one 64-thread workgroup) decodes one CHUNK of up to FPW frames in two phases:
//
//   phase 1, lane-per-frame:  lane s unpacks the bitstream of slot s (Huffman / fixed-width fields),
//            dequantises and mix-accumulates the <=255 frequency-domain words of the frame into row
//            s of an LDS tile.  This is the serial entropy decode (~250 dependent symbol decodes per
//            frame); running FPW frames side by side is what keeps the SIMD lanes busy.
//            [DecoderImpl94x/93/93a::DecompressFrame, DCSDecoderNative.cpp:1679-2261, :2293-2684,
//             :2831-3032; ROMBitPointer, DCSDecoderNative.h:229-289]
//   phase 2, wave-per-frame:  the 64 lanes walk the slots in order and run the fixed-point inverse
//            transform of each row cooperatively (one radix-2 butterfly per lane per stage for the
//            1994+ transform, two for the 1993 one), apply the volume shift, overlap-add with the
//            predecessor's 16-sample tail (kept in LDS) and write 240 int16 PCM samples with
//            coalesced stores.
//            [DecoderImpl94x::TransformFrame :397-576, DecoderImpl93::TransformFrame :614-813]
//
// All arithmetic is the ADSP-2105 fixed-point arithmetic of the reference restated in 32-bit integer
// ops (the reference's 64-bit MR is only ever observed through bits 0..31):
// DCSDecoderNative.h:822-906, .cpp:3447-3580.  No MFMA: this is integer small-transform work.
#pragma once
#include <hip/hip_runtime.h>
#include "dcs_common.h"

namespace dcsk {

constexpr int kRowBytes = 516;          // 256 words + one pad dword: lane-per-frame rows hit distinct LDS banks
constexpr int kScratchBytes = 1040;     // 256 complex points (1993 transform) + pad, 16-byte multiple

__host__ __device__ constexpr int ldsBytes(int fpw)
{
    // tables | tile rows | band types [16][fpw] u16 | header bytes [16][fpw] u8 | tails [fpw][16] i16 | scratch
    return static_cast<int>(sizeof(DcsLdsTables)) + ((fpw * kRowBytes + 15) & ~15) + fpw * 32 + fpw * 16 + fpw * 32 + kScratchBytes;
}

// ------------------------------------------------------------------------------------------------
// L0 arithmetic
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sx16(uint32_t v) { return static_cast<int>(static_cast<int16_t>(v)); }

__device__ __forceinline__ int sat16(int v) { return min(max(v, -32768), 32767); }          // SaturateInt16

// (a*b)<<1 for 16-bit signed operands, as the low word pair of MR (MulSS, .cpp:3556-3567)
__device__ __forceinline__ uint32_t prodSS(int a, int b) { return static_cast<uint32_t>(__mul24(a, b)) << 1; }

// RoundMultiplyResult (.cpp:3503-3514): +0x8000, and clear bit 16 when the LAST PRODUCT's low word
// is exactly 0x8000 (i.e. the un-doubled product has low 15 bits == 0x4000)
__device__ __forceinline__ int roundHi(uint32_t mr, int lastProdUndoubled)
{
    mr += 0x8000u;
    if ((lastProdUndoubled & 0x7FFF) == 0x4000)
        mr &= ~0x10000u;
    return static_cast<int>(mr) >> 16;
}

// complex rotate t = a * (c + i*s) with the reference's operation order (.cpp:500-506, :761-765):
// first term truncating MulSS, second term MultiplyRoundSub / MultiplyRoundAdd
__device__ __forceinline__ void rotate(int are, int aim, int c, int s, int &tre, int &tim)
{
    const int p1 = __mul24(are, c), p2 = __mul24(aim, s);
    tre = roundHi(static_cast<uint32_t>(p1 - p2) << 1, p2);
    const int q1 = __mul24(aim, c), q2 = __mul24(are, s);
    tim = roundHi(static_cast<uint32_t>(q1 + q2) << 1, q2);
}

// CalcExp32 (.cpp:3447-3459)
__device__ __forceinline__ int calcExp32(uint32_t x)
{
    // redundant sign bits: for x >= 0, min(clz(x) - 1, 31) (x == 0 -> 31); for x < 0, clz(~x) - 1
    const uint32_t y = (static_cast<int>(x) < 0) ? ~x : x;
    int n = (y == 0) ? 32 : __clz(static_cast<int>(y));
    n -= 1;
    if (n > 31) n = 31;
    return -n;
}

// ------------------------------------------------------------------------------------------------
// MSB-first bit reader over the blob in global memory; one instance per lane.  64-bit window,
// refilled a big-endian dword at a time, next dword prefetched.  Same VALUES as ROMBitPointer
// (DCSDecoderNative.h:229-289); the reference's byte-granular look-ahead is not observable here.
// ------------------------------------------------------------------------------------------------
struct BitReader
{
    const uint32_t *words;
    uint32_t nWords;
    uint32_t idx;
    uint32_t nxt;
    uint64_t win;
    int cnt;

    __device__ __forceinline__ uint32_t fetch(uint32_t i) const
    {
        return i < nWords ? __builtin_bswap32(words[i]) : 0u;
    }
    __device__ __forceinline__ void init(const uint8_t *blob, uint64_t blobLen, uint64_t bitPos)
    {
        words = reinterpret_cast<const uint32_t *>(blob);
        nWords = static_cast<uint32_t>((blobLen + 3) >> 2);
        idx = static_cast<uint32_t>(bitPos >> 5);
        const int sh = static_cast<int>(bitPos & 31);
        const uint64_t d0 = fetch(idx), d1 = fetch(idx + 1);
        nxt = fetch(idx + 2);
        idx += 3;
        win = ((d0 << 32) | d1) << sh;
        cnt = 64 - sh;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32)
        {
            win |= static_cast<uint64_t>(nxt) << (32 - cnt);
            cnt += 32;
            nxt = fetch(idx++);
        }
    }
    // n in 1..24, after refill()
    __device__ __forceinline__ uint32_t peek(int n) const { return static_cast<uint32_t>(win >> (64 - n)); }
    __device__ __forceinline__ void skip(int n) { win <<= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(int n) { refill(); const uint32_t v = peek(n); skip(n); return v; }
    __device__ __forceinline__ int getSigned(int n)
    {
        refill();
        const int v = static_cast<int>(static_cast<int64_t>(win) >> (64 - n));
        skip(n);
        return v;
    }
};

// prefix code via first-level table + trie (dcs_common.h)
__device__ __forceinline__ int readVlc(BitReader &br, const uint16_t *fast, const uint16_t *trie)
{
    br.refill();
    uint32_t e = fast[br.peek(8)];
    if (e & 0x8000)
    {
        br.skip(static_cast<int>((e >> 8) & 0xF));
        return static_cast<int>(e & 0xFF);
    }
    br.skip(8);
    do
    {
        e = trie[e + br.get(1)];
    }
    while (!(e & 0x8000));
    return static_cast<int>(e & 0xFF);
}

// ------------------------------------------------------------------------------------------------
// per-lane views of the LDS working set
// ------------------------------------------------------------------------------------------------
template <int FPW>
struct Lds
{
    unsigned char *base;
    __device__ __forceinline__ const DcsLdsTables *tables() const { return reinterpret_cast<const DcsLdsTables *>(base); }
    __device__ __forceinline__ uint16_t *row(int s) const
    { return reinterpret_cast<uint16_t *>(base + sizeof(DcsLdsTables) + s * kRowBytes); }
    static constexpr int kSide = static_cast<int>(sizeof(DcsLdsTables)) + ((FPW * kRowBytes + 15) & ~15);
    __device__ __forceinline__ uint16_t *bandTypes() const { return reinterpret_cast<uint16_t *>(base + kSide); }            // [16][FPW]
    __device__ __forceinline__ uint8_t *hdrBytes() const { return base + kSide + FPW * 32; }                                   // [16][FPW]
    __device__ __forceinline__ uint16_t *tails() const { return reinterpret_cast<uint16_t *>(base + kSide + FPW * 48); }     // [FPW][16]
    __device__ __forceinline__ uint32_t *scratch() const { return reinterpret_cast<uint32_t *>(base + kSide + FPW * 80); }   // 260 dwords
};

// the 32-bit "splice" multiply-accumulate of the mixer (.cpp:2244-2250, :2434-2443)
__device__ __forceinline__ void mixAdd(uint16_t *row, int idx, int scaledProduct, uint32_t mixMul)
{
    const uint32_t s = static_cast<uint32_t>(scaledProduct) & 0xFFFFu;
    uint32_t acc = (static_cast<uint32_t>(row[idx]) << 16) | s;
    acc += static_cast<uint32_t>(sx16(s) * static_cast<int>(mixMul));
    row[idx] = static_cast<uint16_t>(acc >> 16);
}
// the same contribution removed again (exact inverse: the MAC is additive modulo 2^16 in the high word)
__device__ __forceinline__ void mixSub(uint16_t *row, int idx, int scaledProduct, uint32_t mixMul)
{
    const uint32_t s = static_cast<uint32_t>(scaledProduct) & 0xFFFFu;
    const uint32_t c = (s + static_cast<uint32_t>(sx16(s) * static_cast<int>(mixMul))) >> 16;
    row[idx] = static_cast<uint16_t>(row[idx] - c);
}

__device__ __forceinline__ uint32_t scaleFactor(const DcsLdsTables *T, int code)
{
    return static_cast<uint32_t>(T->scaleMant[code & 3]) >> (15 - ((code >> 2) & 15));
}

__device__ __forceinline__ void dcFixup(uint16_t *row, uint32_t saved1)
{
    const int delta = sat16(sx16(row[1]) - sx16(saved1));
    row[0] = static_cast<uint16_t>(sat16(delta + sx16(row[0])));
    row[1] = static_cast<uint16_t>(saved1);
}

// ------------------------------------------------------------------------------------------------
// a2: 1994+ frame (DecoderImpl94x::DecompressFrame, .cpp:1679-2261)
// ------------------------------------------------------------------------------------------------
template <int FPW>
__device__ uint32_t unpack94(const Lds<FPW> &L, int lane, BitReader &br, int format, uint32_t mixMul)
{
    const DcsLdsTables *T = L.tables();
    uint16_t *row = L.row(lane);
    uint16_t *bt = L.bandTypes() + lane;            // element b at bt[b * FPW]
    const uint8_t *hdr = L.hdrBytes() + lane;       // element b at hdr[b * FPW]
    const bool type1 = format != DCS_FMT_94_T0;
    const uint32_t saved1 = row[1];
    uint32_t err = 0;

    // scale pre-adjust for bands 0..2 from the PREVIOUS frame's codes (:1744-1773)
    int preAdj0 = 0, preAdj1 = 0, preAdj2 = 0;
    if (type1)
    {
        const uint8_t *map = T->preAdj94 + (format == DCS_FMT_94_T1_S0 ? 0 : 16);
        const uint32_t c0 = bt[0], c1 = bt[FPW], c2 = bt[2 * FPW];
        if ((c0 | c1 | c2) > 15)
            return DCS_FRAME_FATAL | DCS_FRAME_STOP;
        preAdj0 = map[c0]; preAdj1 = map[c1]; preAdj2 = map[c2];
    }

    // frame header: band-type deltas (:1780-1834)
    for (int i = 0 ; i < 16 ; ++i)
    {
        if ((hdr[i * FPW] & 0x7F) == 0x7F)
            break;
        bt[i * FPW] = static_cast<uint16_t>(bt[i * FPW] + readVlc(br, T->fast94, T->trie94) - 16);
    }

    int outIdx = 1;
    bool valid = true;
    for (int band = 0 ; band < 16 ; ++band)
    {
        int hb = hdr[band * FPW] & 0x7F;
        if (hb == 0x7F)
            break;
        int count = band == 0 ? 7 : band == 1 ? 8 : band == 15 ? 32 : 16;          // :1848-1850
        int inc = 1;
        if (hb & 0x40) { inc = 2; count >>= 1; }

        int code = bt[band * FPW];
        if (code == 0)
        {
            outIdx += count;                        // the halved count, not count*inc (:1886)
            continue;
        }
        int scaleCode = hb;
        if (type1)
        {
            if (code > 15) { err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; break; }
            const uint32_t x = T->xlat94[(band < 3 ? 0 : band < 6 ? 16 : 32) + code];
            if (band < 3)
                hb += band == 0 ? preAdj0 : band == 1 ? preAdj1 : preAdj2;
            scaleCode = hb + static_cast<int>(x >> 8);
            code = static_cast<int>(x & 0xFF);
        }
        const int scale = static_cast<int>(scaleFactor(T, scaleCode));

        if (code == 0)
        {
            valid = false; err |= DCS_FRAME_STOP;   // :1985-1991
            outIdx += count * inc;
        }
        else if (code <= 6)
        {
            const uint32_t info = T->cbInfo[code];
            const int maxBits = static_cast<int>(info & 0xF);
            const uint16_t *book = T->cb94 + (info >> 4);
            const int ref = 1 << (code - 1);
            const BitReader bandStart = br;
            const int idxStart = outIdx;
            bool bad = false;
            for (int i = count ; i > 0 ; )
            {
                br.refill();
                const uint32_t e = book[br.peek(maxBits)];
                br.skip(static_cast<int>(e >> 8));
                if (e & 0x80)
                {
                    if (i >= 2) { outIdx += 2 * inc; i -= 2; }
                    else { bad = true; outIdx += inc; i = 0; }              // :2213-2218
                }
                else
                {
                    if (valid)
                        mixAdd(row, outIdx, (static_cast<int>(e & 0xFF) - ref) * scale, mixMul);
                    outIdx += inc; --i;
                }
            }
            if (bad)
            {
                // the reference zeroes the WHOLE band buffer on this error (:2238-2239): take back
                // what this band already contributed by replaying it
                if (valid)
                {
                    BitReader r2 = bandStart;
                    int k = idxStart;
                    for (int i = count ; i > 1 ; )
                    {
                        r2.refill();
                        const uint32_t e = book[r2.peek(maxBits)];
                        r2.skip(static_cast<int>(e >> 8));
                        if (e & 0x80) { k += 2 * inc; i -= 2; }
                        else { mixSub(row, k, (static_cast<int>(e & 0xFF) - ref) * scale, mixMul); k += inc; --i; }
                    }
                }
                valid = false; err |= DCS_FRAME_STOP;
            }
        }
        else
        {
            if (code > 16) { err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; break; }
            for (int i = 0 ; i < count ; ++i, outIdx += inc)
            {
                const int v = br.getSigned(code);
                if (valid)
                    mixAdd(row, outIdx, static_cast<int16_t>(v) * scale, mixMul);
            }
        }
    }

    dcFixup(row, saved1);
    return err;
}

// ------------------------------------------------------------------------------------------------
// a3: 1993 frame, Type 0 and OS93b Type 1 (DecoderImpl93::DecompressFrame + ReadHuff93, .cpp:2293-2684)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mixAdd93(uint16_t *row, int idx, int scaledProduct, uint32_t mixMul)
{
    if (idx < 256)          // words 256..511 of the reference's buffer never reach the output (:714-732 overwrites them)
        mixAdd(row, idx, scaledProduct, mixMul);
}

template <int FPW>
__device__ uint32_t unpack93(const Lds<FPW> &L, int lane, BitReader &br, int format, uint32_t mixMul)
{
    const DcsLdsTables *T = L.tables();
    uint16_t *row = L.row(lane);
    uint16_t *bt = L.bandTypes() + lane;
    const uint8_t *hdr = L.hdrBytes() + lane;
    const bool type1 = format == DCS_FMT_93B_T1;
    const uint32_t saved1 = row[1];
    uint32_t err = 0;

    int subType = type1 ? 0 : 2;
    bool first = true, reuse = false;
    uint32_t prv = 0, prvDelta = 0;         // uint16 semantics: masked on use
    int code = 0;
    int outIdx = 1;

    for (int band = 0 ; band < 16 ; ++band)
    {
        const int hb = hdr[band * FPW] & 0x7F;
        if (hb == 0x7F)
            break;
        const int scale = static_cast<int>(scaleFactor(T, hb));
        const bool strided = (hb >> 6) != 0;

        int nSamples, inc, fixup, stride;
        if (!type1)
        {
            nSamples = 16;
            if (!strided) { inc = 1; fixup = 0; stride = 16; }
            else { ++outIdx; inc = 2; fixup = -1; stride = 31; }
        }
        else
        {
            fixup = 0;
            if (!strided) { inc = 1; nSamples = stride = first ? 15 : 16; }
            else { inc = 2; nSamples = stride = 8; }
        }

        if (reuse)
            reuse = br.get(1) != 0;
        if (!reuse)
        {
            if (!type1)
            {
                if (br.get(1))
                    subType = br.get(1) ? (subType == 2 ? 0 : subType + 1) : (subType == 0 ? 2 : subType - 1);
                code = static_cast<int>(br.get(4));
            }
            else
            {
                int v = readVlc(br, T->fast93, T->trie93);
                if (v < 0x1E)
                    v -= 0x0F;
                else
                {
                    v -= 0x2E;
                    subType = subType != 0 ? 0 : 1;
                }
                const uint32_t nc = (bt[band * FPW] + static_cast<uint32_t>(v)) & 0xFFFFu;
                bt[band * FPW] = static_cast<uint16_t>(nc);
                code = static_cast<int>(nc);
            }
        }

        if (code == 0)
        {
            reuse = true;
            if (subType == 0)
            {
                outIdx += stride;
                prv = 0; prvDelta = 0;
            }
            else if (subType == 1)
            {
                // repeat the previous input; the product's low word is carried, not reloaded (:2513-2534)
                uint32_t low = static_cast<uint32_t>(sx16(prv) * scale) & 0xFFFFu;
                const int mulLow = sx16(low);
                for (int i = 0 ; i < nSamples ; ++i, outIdx += inc)
                {
                    if (outIdx < 256)
                    {
                        uint32_t acc = (static_cast<uint32_t>(row[outIdx]) << 16) | low;
                        acc += static_cast<uint32_t>(mulLow * static_cast<int>(mixMul));
                        row[outIdx] = static_cast<uint16_t>(acc >> 16);
                        low = acc & 0xFFFFu;
                    }
                }
                prvDelta = 0;
                outIdx += fixup;
            }
            else
            {
                for (int i = 0 ; i < nSamples ; ++i, outIdx += inc)
                {
                    prv = (prv + prvDelta) & 0xFFFFu;
                    mixAdd93(row, outIdx, sx16(prv) * scale, mixMul);
                }
                outIdx += fixup;
            }
        }
        else
        {
            const int width = code + (type1 ? 0 : 1);
            if (width > 16) { err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; break; }
            uint32_t last = 0, last2 = 0;
            for (int i = 0 ; i < nSamples ; ++i, outIdx += inc)
            {
                const uint32_t in = static_cast<uint32_t>(br.getSigned(width)) & 0xFFFFu;
                if (subType == 0)
                {
                    mixAdd93(row, outIdx, sx16(in) * scale, mixMul);
                    last2 = last; last = in;
                }
                else
                {
                    prvDelta = (subType == 1) ? in : ((prvDelta + in) & 0xFFFFu);
                    prv = (prv + prvDelta) & 0xFFFFu;
                    mixAdd93(row, outIdx, sx16(prv) * scale, mixMul);
                }
            }
            if (subType == 0)
            {
                prv = last;
                prvDelta = (last - last2) & 0xFFFFu;
            }
            outIdx += fixup;
        }
        first = false;
    }

    dcFixup(row, saved1);
    return err;
}

// ------------------------------------------------------------------------------------------------
// a4: OS93a Type 1 frame (DecoderImpl93a::DecompressFrame, .cpp:2831-3032)
// ------------------------------------------------------------------------------------------------
template <int FPW>
__device__ uint32_t unpack93a(const Lds<FPW> &L, int lane, BitReader &br, uint32_t mixMul, const uint16_t *pairTable)
{
    const DcsLdsTables *T = L.tables();
    uint16_t *row = L.row(lane);
    const int hb = L.hdrBytes()[lane];
    const uint16_t *bbBook = &T->bandBits93a[(hb & 0x60) >> 1];
    const int numBands = hb & 0x1F;
    int prvScale = 0x1A;
    int outIdx = 0;
    uint32_t err = 0;

    for (int band = 0 ; band < numBands ; ++band)
    {
        if (band >= 18) { err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; break; }
        const int numInputs = T->inputs93a[band];

        br.refill();
        const uint32_t e = bbBook[br.peek(4)];
        br.skip(static_cast<int>(e >> 8));
        const int bandBits = static_cast<int>(e & 0xFF);
        if (bandBits == 0xFF)
            break;
        if (bandBits == 0)
        {
            outIdx += numInputs * 2;
            continue;
        }

        br.refill();
        uint32_t sc = T->scaleCb93a[br.peek(4)];
        br.skip(static_cast<int>((sc >> 8) & 0xF));
        if ((sc & 0xFF) == 0xFF)
        {
            br.refill();
            sc = T->scaleCb93a[((sc >> 12) << 4) + br.peek(4)];
            br.skip(static_cast<int>((sc >> 8) & 0xF) - 4);
        }

        int scaleCode = prvScale + static_cast<int>(sc & 0xFF) - 1 + bandBits * 2;
        if (scaleCode > 0x39)
            scaleCode -= 0x36;
        prvScale = scaleCode - bandBits * 2;

        uint32_t sf = 0x8000;
        for (int i = 0 ; i < (scaleCode & 3) ; ++i)
            sf = (sf * 0x9838u) >> 15;
        sf <<= (scaleCode >> 2);
        sf = ((sf >> 16) * mixMul) >> 15;
        const int sfs = sx16(sf);                   // truncated to 16 bits, then read as signed (:2995, :3011)

        const uint16_t *pairBase = pairTable + (2 << bandBits);
        for (int i = 0 ; i < numInputs ; ++i)
        {
            const uint16_t *pair = pairBase + 2 * br.get(bandBits);
            for (int k = 0 ; k < 2 ; ++k, ++outIdx)
            {
                const int p = __mul24(sx16(pair[k]), sfs);
                row[outIdx] = static_cast<uint16_t>(roundHi((static_cast<uint32_t>(row[outIdx]) << 16) + (static_cast<uint32_t>(p) << 1), p));
            }
        }
    }
    return err;
}

// ------------------------------------------------------------------------------------------------
// 1993 transform, DC step: |f0 + i f1| by a 5th-order polynomial square root (.cpp:635-710).
// Scalar per frame, so it runs lane-per-frame at the end of phase 1.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mulRound(int a, int b)     // MultiplyAndRound: returns full MR
{
    const int p = __mul24(a, b);
    uint32_t mr = (static_cast<uint32_t>(p) << 1) + 0x8000u;
    if ((p & 0x7FFF) == 0x4000)
        mr &= ~0x10000u;
    return mr;
}

__device__ void dcMagnitude93(uint16_t *row)
{
    uint32_t ar = row[0];
    const bool neg = sx16(ar) < 0;
    if (neg)
        ar = static_cast<uint32_t>(-sx16(ar)) & 0xFFFFu;
    const int f1 = sx16(row[1]);
    uint32_t sr = prodSS(f1, f1) + prodSS(sx16(ar), sx16(ar));
    int exponent = calcExp32(sr);
    if (exponent < 0)
        sr <<= -exponent;
    ar = sr >> 16;
    if (ar != 0)
    {
        const int x = sx16(ar);
        uint32_t mr = 0x0D490000u;
        mr += static_cast<uint32_t>(0x5D1D * x) << 1;
        int mf = static_cast<int>(mulRound(x, x)) >> 16;
        mr += static_cast<uint32_t>(-22035 * mf) << 1;
        mf = static_cast<int>(mulRound(x, mf)) >> 16;
        mr += static_cast<uint32_t>(0x46D6 * mf) << 1;
        mf = static_cast<int>(mulRound(x, mf)) >> 16;
        mr += static_cast<uint32_t>(-8790 * mf) << 1;
        mf = static_cast<int>(mulRound(x, mf)) >> 16;
        mr += static_cast<uint32_t>(0x072D * mf) << 1;
        if (exponent & 1)
        {
            mr = mulRound(static_cast<int>(mr) >> 16, 0x5A82);
            exponent += 1;
        }
        exponent = exponent / 2 + 1;
        uint32_t sh;
        if (exponent >= 0) sh = mr << exponent;
        else sh = static_cast<uint32_t>(static_cast<int>(mr) >> (-exponent));     // arithmetic for negatives, logical == arithmetic for positives
        ar = sh >> 16;
        if (neg)
            ar = static_cast<uint32_t>(-sx16(ar)) & 0xFFFFu;
    }
    row[0] = static_cast<uint16_t>(ar);
    row[1] = 0;
}

// ------------------------------------------------------------------------------------------------
// phase 2 helpers: complex points are dwords, low half = real, high half = imaginary
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t packC(int re, int im) { return (static_cast<uint32_t>(re) & 0xFFFFu) | (static_cast<uint32_t>(im) << 16); }
__device__ __forceinline__ int reC(uint32_t c) { return sx16(c); }
__device__ __forceinline__ int imC(uint32_t c) { return static_cast<int>(c) >> 16; }

__device__ __forceinline__ int bitrev6(int v) { return static_cast<int>(__brev(static_cast<uint32_t>(v)) >> 26); }

// overlap-add of one sample (.cpp:545-554, :797-801): both products signed x unsigned, sum, +0x8000, high word
__device__ __forceinline__ int overlapMix(int x, uint32_t cx, int o, uint32_t co)
{
    const uint32_t a = static_cast<uint32_t>(x * static_cast<int>(cx)) << 1;
    const uint32_t b = static_cast<uint32_t>(o * static_cast<int>(co)) << 1;
    return static_cast<int>(a + b + 0x8000u) >> 16;
}

struct LaneConsts
{
    uint32_t tw94[6];       // per stage: cos | sin<<16 for this lane's butterfly
    uint32_t twPre94;       // pre-twiddle pair c0 | c1<<16 (.cpp:428-429)
    uint32_t tw93[7][2];    // two butterflies per lane per stage
    uint32_t ovlA, ovlB;    // overlap window: co[2l] | co[2l+1]<<16, co[15-2l] | co[14-2l]<<16 (lanes 0..7)
};

__device__ __forceinline__ int bitrev9(int v) { return static_cast<int>(__brev(static_cast<uint32_t>(v)) >> 23); }

__device__ void loadLaneConsts(const DcsDevTables *G, int lane, LaneConsts &C)
{
    const uint16_t *coef = G->fftCoef;
    for (int st = 0 ; st < 6 ; ++st)
    {
        const int part = lane >> (5 - st);                      // butterfly lane of stage st lives in partition lane / (32 >> st)
        C.tw94[st] = static_cast<uint32_t>(coef[0x80 + part]) | (static_cast<uint32_t>(coef[part]) << 16);
    }
    C.twPre94 = static_cast<uint32_t>(coef[bitrev9(2 + 4 * lane)]) | (static_cast<uint32_t>(coef[bitrev9(4 * lane)]) << 16);
    for (int st = 0 ; st < 7 ; ++st)
        for (int h = 0 ; h < 2 ; ++h)
        {
            const int part = (lane + 64 * h) >> (6 - st);
            C.tw93[st][h] = static_cast<uint32_t>(coef[0x80 + part]) | (static_cast<uint32_t>(coef[part]) << 16);
        }
    const int l = lane & 7;
    C.ovlA = static_cast<uint32_t>(G->ovlCoef[2 * l]) | (static_cast<uint32_t>(G->ovlCoef[2 * l + 1]) << 16);
    C.ovlB = static_cast<uint32_t>(G->ovlCoef[15 - 2 * l]) | (static_cast<uint32_t>(G->ovlCoef[14 - 2 * l]) << 16);
}

// block of one wavefront: this is a fence for the compiler plus s_waitcnt; hipcc drops the s_barrier
// itself when the workgroup is a single wave
__device__ __forceinline__ void waveSync() { __syncthreads(); }

// 1994+ transform of one row into 256 time samples held as S[0..127] (bit-reversed order) (.cpp:397-524)
__device__ void transform94(const uint32_t *rowC, uint32_t *S, int lane, const LaneConsts &C)
{
    // pre-pass 1 + 2 on the pair (point lane, point 128 - lane) (:403-456)
    {
        const uint32_t X = rowC[lane];
        const uint32_t Y = lane == 0 ? 0u : rowC[128 - lane];       // words 0x100/0x101 start at zero
        const int x0 = reC(X), x1 = imC(X), y0 = reC(Y), y1 = imC(Y);
        // MulSS(v, 0x8000) is a wrapping 16-bit negate
        int a0 = sx16(static_cast<uint32_t>(-sat16(x0 + y0)));
        int b0 = sx16(static_cast<uint32_t>(-sat16(x0 - y0)));
        int a1 = sx16(static_cast<uint32_t>(-sat16(x1 - y1)));
        int b1 = sx16(static_cast<uint32_t>(-sat16(x1 + y1)));
        const int c0 = sx16(C.twPre94), c1 = static_cast<int>(C.twPre94) >> 16;
        // prod0 = b1*c1 - b0*c0 ; prod1 = b1*c0 + b0*c1
        const int p1 = __mul24(b1, c1), p2 = __mul24(b0, c0);
        const int prod0 = roundHi(static_cast<uint32_t>(p1 - p2) << 1, p2);
        const int q1 = __mul24(b1, c0), q2 = __mul24(b0, c1);
        const int prod1 = roundHi(static_cast<uint32_t>(q1 + q2) << 1, q2);
        const uint32_t A = packC(sat16(prod1 + a0), sat16(prod0 + a1));
        const uint32_t B = packC(sat16(a0 - prod1), sat16(prod0 - a1));
        S[lane] = A;
        if (lane != 0)
            S[128 - lane] = B;
        else
        {
            // point 64: real part negated, imaginary part unchanged (:403-404)
            const uint32_t M = rowC[64];
            S[64] = packC(sx16(static_cast<uint32_t>(-reC(M))), imC(M));
        }
    }
    waveSync();

    // pre-pass 3 (:458-471): saturating radix-2 across the two halves, then six stages (:480-524)
    uint32_t U = S[lane], A = S[lane + 64];
    {
        const int ur = reC(U), ui = imC(U), ar = reC(A), ai = imC(A);
        U = packC(sat16(ur + ar), sat16(ui + ai));
        A = packC(sat16(ur - ar), sat16(ui - ai));
    }
    S[lane] = U; S[lane + 64] = A;
    waveSync();

#pragma unroll
    for (int st = 0 ; st < 6 ; ++st)
    {
        const int d = 32 >> st;
        const int u = ((lane >> (5 - st)) << (6 - st)) | (lane & (d - 1));
        U = S[u]; A = S[u + d];
        int tre, tim;
        rotate(reC(A), imC(A), sx16(C.tw94[st]), static_cast<int>(C.tw94[st]) >> 16, tre, tim);
        const int ur = reC(U), ui = imC(U);
        S[u] = packC(sat16(ur - tre), sat16(ui - tim));
        S[u + d] = packC(sat16(ur + tre), sat16(ui + tim));
        waveSync();
    }
}

// 1993 transform of one row (DC step already applied) into S[0..255] (.cpp:714-778)
__device__ void transform93(const uint32_t *rowC, uint32_t *S, int lane, const LaneConsts &C)
{
    // expand 128 -> 256 complex points with wrapping adds (:714-732)
    {
        const uint32_t X = rowC[1 + lane], Y = rowC[127 - lane];
        const int xr = reC(X), xi = imC(X), yr = reC(Y), yi = imC(Y);
        S[1 + lane]   = packC(xr + yr, xi - yi);
        S[127 - lane] = packC(xr + yr, yi - xi);
        S[129 + lane] = packC(xr - yr, xi + yi);
        S[255 - lane] = packC(yr - xr, xi + yi);
        if (lane == 0)
        {
            const uint32_t Z = rowC[0];         // (|DC|, 0) from dcMagnitude93
            S[0] = Z; S[128] = Z;
        }
    }
    waveSync();

#pragma unroll
    for (int st = 0 ; st < 7 ; ++st)
    {
        const int d = 64 >> st;
        uint32_t o[2][2];
#pragma unroll
        for (int h = 0 ; h < 2 ; ++h)
        {
            const int k = lane + 64 * h;
            const int u = ((k >> (6 - st)) << (7 - st)) | (k & (d - 1));
            const uint32_t U = S[u], A = S[u + d];
            int tre, tim;
            rotate(reC(A), imC(A), sx16(C.tw93[st][h]), static_cast<int>(C.tw93[st][h]) >> 16, tre, tim);
            o[h][0] = packC(reC(U) - tre, imC(U) - tim);
            o[h][1] = packC(tre + reC(U), tim + imC(U));
        }
        waveSync();
#pragma unroll
        for (int h = 0 ; h < 2 ; ++h)
        {
            const int k = lane + 64 * h;
            const int u = ((k >> (6 - st)) << (7 - st)) | (k & (d - 1));
            S[u] = o[h][0]; S[u + d] = o[h][1];
        }
        waveSync();
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <int FPW>
__global__ void __launch_bounds__(64) dcsDecodeKernel(const DcsKernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lds<FPW> L{ smem };
    const int lane = static_cast<int>(threadIdx.x);
    const uint32_t chunk = blockIdx.x;

    // ---- stage tables, clear the tile --------------------------------------------------------
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&a.tables->lds);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (int i = lane ; i < static_cast<int>(sizeof(DcsLdsTables) / 16) ; i += 64)
            dst[i] = src[i];
        uint4 *tile = reinterpret_cast<uint4 *>(smem + sizeof(DcsLdsTables));
        constexpr int kTile16 = ((FPW * kRowBytes + 15) & ~15) / 16;
        for (int i = lane ; i < kTile16 ; i += 64)
            tile[i] = make_uint4(0, 0, 0, 0);
    }

    LaneConsts C;
    loadLaneConsts(a.tables, lane, C);

    // ---- slot and job of this lane -------------------------------------------------------------
    DcsSlot slot{ 0xFFFFFFFFu, DCS_NO_PREV_SLOT, DCS_SLOT_EMPTY, 0 };
    if (lane < FPW)
        slot = a.slots[static_cast<size_t>(chunk) * FPW + lane];
    const bool live = !(slot.flags & DCS_SLOT_EMPTY);
    DcsFrameJob job{ 0, 0, 0, DCS_XFORM_94, 0, DCS_PREV_NONE, 0 };
    if (live)
        job = a.jobs[slot.job];
    waveSync();

    // ---- phase 1: lane-per-frame unpack ----------------------------------------------------------
    uint32_t err = 0;
    if (live)
    {
        for (int s = 0 ; s < job.nSrc ; ++s)
        {
            const DcsSrcDesc *sd = &a.srcs[job.firstSrc + s];
            const uint64_t streamOff = sd->streamOff;
            const int format = sd->format;
            const int hdrLen = sd->hdrLen;
            const uint32_t mixMul = sd->mixMul;
            // carried band types and the stream header into this lane's LDS columns
            uint16_t *bt = L.bandTypes() + lane;
            uint8_t *hb = L.hdrBytes() + lane;
            for (int i = 0 ; i < 16 ; ++i)
            {
                bt[i * FPW] = sd->bandType[i];
                const uint64_t at = streamOff + 2 + static_cast<uint64_t>(i);
                hb[i * FPW] = (i < hdrLen && at < a.blobLen) ? a.blob[at] : static_cast<uint8_t>(0);
            }
            BitReader br;
            br.init(a.blob, a.blobLen, (streamOff + 2 + static_cast<uint64_t>(hdrLen)) * 8 + sd->bitOff);
            uint32_t e;
            if (format >= DCS_FMT_94_T0)
                e = unpack94<FPW>(L, lane, br, format, mixMul);
            else if (format == DCS_FMT_93A_T1)
                e = unpack93a<FPW>(L, lane, br, mixMul, a.tables->pair93a);
            else
                e = unpack93<FPW>(L, lane, br, format, mixMul);
            err |= e;
        }
        if (job.xform == DCS_XFORM_93)
            dcMagnitude93(L.row(lane));
        if (!(slot.flags & DCS_SLOT_HALO) && a.err != nullptr)
            a.err[slot.job] = err;
    }
    waveSync();

    // ---- phase 2: wave-per-frame transform, overlap, emit -----------------------------------------
    uint32_t *S = L.scratch();
    uint16_t *tails = L.tails();
    for (int s = 0 ; s < FPW ; ++s)
    {
        const uint32_t flags = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(slot.flags), s));
        if (flags & DCS_SLOT_EMPTY)
            break;
        const uint32_t jobIdx = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(slot.job), s));
        const int prevSlot = __builtin_amdgcn_readlane(static_cast<int>(slot.prevSlot), s);
        const int volShift = __builtin_amdgcn_readlane(static_cast<int>(job.volShift), s);
        const int xform = __builtin_amdgcn_readlane(static_cast<int>(job.xform), s);
        const uint32_t prevJob = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(job.prev), s));
        const uint32_t *rowC = reinterpret_cast<const uint32_t *>(L.row(s));

        // predecessor's tail pair for lanes 0..7
        uint32_t tailPair = 0;
        if (lane < 8)
        {
            if (flags & DCS_SLOT_EXT_TAIL)
            {
                if (a.tailsIn != nullptr)
                    tailPair = reinterpret_cast<const uint32_t *>(a.tailsIn)[static_cast<size_t>(prevJob & 0x7FFFFFFFu) * 8 + lane];
            }
            else if (prevSlot != DCS_NO_PREV_SLOT)
                tailPair = reinterpret_cast<const uint32_t *>(tails)[prevSlot * 8 + lane];
        }

        uint32_t first, second;         // sample pairs (2l, 2l+1) and (2l+128, 2l+129)
        if (xform == DCS_XFORM_94)
        {
            transform94(rowC, S, lane, C);
            const int r = bitrev6(lane) * 2;
            const uint32_t P0 = S[r], P1 = S[r + 1];
            first  = packC(reC(P0) >> volShift, imC(P0) >> volShift);         // :532-534
            second = packC(reC(P1) >> volShift, imC(P1) >> volShift);
        }
        else
        {
            transform93(rowC, S, lane, C);
            // sample i = Re(Q[bitrev8(i)]) >> volShift (:782-785); this lane takes samples 2l, 2l+1, 2l+128, 2l+129
            const int r0 = bitrev6(lane) * 2;           // bitrev8(2l)   = bitrev6(l) << 1        (l < 64)
            const uint32_t Q0 = S[r0], Q1 = S[r0 + 128], Q2 = S[r0 + 1], Q3 = S[r0 + 129];
            first  = packC(reC(Q0) >> volShift, reC(Q1) >> volShift);         // bitrev8(2l+1)   = r0 + 128
            second = packC(reC(Q2) >> volShift, reC(Q3) >> volShift);         // bitrev8(2l+128) = r0 + 1
        }

        if (lane < 8)
        {
            // overlap-add with the predecessor's last 16 samples (:538-555, :789-802)
            const int s0 = overlapMix(reC(first), C.ovlA & 0xFFFFu, reC(tailPair), C.ovlB & 0xFFFFu);
            const int s1 = overlapMix(imC(first), C.ovlA >> 16, imC(tailPair), C.ovlB >> 16);
            first = packC(s0, s1);
        }

        // tail for the successor: samples 240..255 = `second` of lanes 56..63 (:569-575, :805-812)
        if (lane >= 56)
            reinterpret_cast<uint32_t *>(tails)[s * 8 + (lane - 56)] = second;

        if (!(flags & DCS_SLOT_HALO))
        {
            uint32_t *out = reinterpret_cast<uint32_t *>(a.pcm) + static_cast<size_t>(jobIdx) * (DCS_FRAME_SAMPLES / 2);
            out[lane] = first;
            if (lane < 56)
                out[64 + lane] = second;
            else if (a.tailsOut != nullptr)
                reinterpret_cast<uint32_t *>(a.tailsOut)[static_cast<size_t>(jobIdx) * 8 + (lane - 56)] = second;
        }
        waveSync();
    }
}

}   // namespace dcsk

// dcs_kernels.hip.h -- CDNA4 (gfx950) kernels for batched DCS frame decode.
//
// One wavefront decodes one CHUNK of up to FPW frames in two phases (four wavefronts form a workgroup and
// share the LDS copy of the tables):
//
//   phase 1, unpack, SUB = 64 / FPW lanes per frame:  the index pass recorded the decoder state at the
//            start of every header band (DcsSplit), so 4, 8 or 16 lanes unpack one frame in parallel, a
//            few bands each: Huffman / fixed-width fields -> dequantise -> mix-accumulate the <=255
//            frequency-domain words into the frame's row of an LDS tile.  All 64 lanes of the wavefront
//            work on the serial entropy decode (16..80 dependent symbol decodes per lane instead of
//            ~250 per frame).  Everything a chunk needs -- slots, descriptor heads, headers, split records,
//            the compressed bytes as an image of the LDS bit pool (dwords in bit order) -- was gathered once
//            per batch into the chunk's PACKAGE by the host (dcsBuildPackages) and is requested with the wavefront's first
//            instructions, so the per-symbol critical path never waits on HBM/L2.
//            [DecoderImpl94x/93/93a::DecompressFrame, DCSDecoderNative.cpp:1679-2261, :2293-2684,
//             :2831-3032; ROMBitPointer, DCSDecoderNative.h:229-289]
//   phase 2, transform, 8 or 16 lanes per frame:  register-resident fixed-point inverse transforms of
//            8 (1994+) or 4 (1993) rows per pass, volume shift, overlap-add with the predecessor's
//            16-sample tail (through LDS inside a chunk, through an epoch-tagged hand-off buffer from the
//            chunk before), 240 int16 PCM samples written per frame.
//            [DecoderImpl94x::TransformFrame :397-576, DecoderImpl93::TransformFrame :614-813]
//
// All arithmetic is the ADSP-2105 fixed-point arithmetic of the reference restated in 32-bit integer
// ops (the reference's 64-bit MR is only ever observed through bits 0..31):
// DCSDecoderNative.h:822-906, .cpp:3447-3580.  No MFMA: this is integer small-transform work.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstddef>
#include "dcs_common.h"

namespace dcsk {

// diagnostic build only (-DDCS_STAMPS): per-chunk cycle stamps at the phase boundaries, written to a debug
// buffer no other code reads (args.debug); never enabled in the shipped library
struct Stamper
{
#ifdef DCS_STAMPS
    unsigned long long *p;          // this chunk's 16 stamps; null on every lane but lane 0
    // (DCS_STAMPS_REALTIME: the 100 MHz clock all compute units share, for a timeline of the launch: tools/timeline.py)
#ifdef DCS_STAMPS_REALTIME
    __device__ __forceinline__ void operator()(int k) const { if (p != nullptr) p[k] = __builtin_amdgcn_s_memrealtime(); }
#else
    __device__ __forceinline__ void operator()(int k) const { if (p != nullptr) p[k] = __builtin_amdgcn_s_memtime(); }
#endif
#else
    __device__ __forceinline__ void operator()(int) const { }
#endif
};
#define DCS_STAMP(k) stamp(k)

// Tails between chunks: a RENDEZVOUS, nobody waits (round 6; rounds 2-5 had the consumer poll for up to 500 ms, which rested on
// workgroups being dispatched in index order).  The last frame of a chunk whose successor lies in another chunk (the producer,
// DCS_SLOT_EXPORT) and that successor (the consumer, DCS_SLOT_IMPORT) each do ONE 64-bit atomic exchange per tail word on the
// producing chunk's row of the hand-off buffer: the producer offers  epoch | 0 | its tail sample (pair),  the consumer
// epoch | 1 | its own first output sample (pair) before the overlap.  Whoever finds the other's word of THIS launch in what the
// exchange returns is the second to arrive, holds both halves, computes the overlap (DCSDecoderNative.cpp:538-555, :789-802) and
// stores the sample into the consumer's PCM row; whoever finds anything else was first and is finished for.  Both sides hold the
// overlap window (one transform per chain), the producer knows the consumer's job from its slot (DcsSlot::nextJob).  No order of
// dispatch, no residency and no other launch on the chip matters, and nothing can be lost.
//   word = epoch (31 bits) << 33 | who << 32 | payload (32 bits; 1993 transform: one 16-bit sample)
__device__ __forceinline__ unsigned long long handoffWord(uint32_t epoch, uint32_t who, uint32_t payload)
{
    return (static_cast<unsigned long long>(epoch) << 33) | (static_cast<unsigned long long>(who) << 32) | payload;
}
// the other side's word of this launch?
__device__ __forceinline__ bool handoffMeets(unsigned long long old, uint32_t epoch, uint32_t whoWanted)
{
    return static_cast<uint32_t>(old >> 32) == ((epoch << 1) | whoWanted);
}
constexpr int kWavesPerBlock = 4;       // wavefronts of a workgroup share one copy of the decode tables in LDS
#ifndef DCS_ROW_BYTES
#define DCS_ROW_BYTES 528
#endif
constexpr int kRowBytes = DCS_ROW_BYTES; // 256 words + 16 bytes: 16-byte aligned rows (the transforms transpose through them
                                        // with 128-bit accesses), rows of neighbouring frames land on different LDS banks

__host__ __device__ constexpr int poolDwords(int fpw) { return static_cast<int>(dcsPoolCapacity(fpw)); }
__host__ __device__ constexpr int subLanes(int fpw) { return 64 / fpw; }     // 16, 8 and 4 lanes per frame for fpw 4, 8, 16

// per wavefront: tile rows | tails [fpw + 1][16] i16 (the last row is a sink for lane groups without a frame) | bit pool
__host__ __device__ constexpr int waveLdsBytes(int fpw) { return fpw * kRowBytes + (fpw + 1) * 32 + poolDwords(fpw) * 4; }
// workgroup layout: tables | the four wavefronts' bit pools | the four wavefronts' (tile rows, tails).  The pools
// come first on purpose, see BitReader.
// ... | OS93a Type-1 pair table (4 KB; only where it fits next to three more workgroups: 4 and 8 frames per wavefront)
__host__ __device__ constexpr bool pairTableInLds(int fpw) { return fpw <= 8; }
__host__ __device__ constexpr int ldsBytes(int fpw)
{
    return DCS_LDS_DECODE_BYTES + kWavesPerBlock * waveLdsBytes(fpw) + (pairTableInLds(fpw) ? 4096 : 0);
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
// MSB-first bit reader over the LDS bit pool; one instance per lane.  The pool holds the frame's
// dwords already byte-swapped, so a dword's MSB is the next bit of the stream.  64-bit window kept as
// two 32-bit registers (hi:lo), refilled a dword at a time.  Same VALUES as ROMBitPointer
// (DCSDecoderNative.h:229-289); the reference's byte-granular look-ahead is not observable here.
// ------------------------------------------------------------------------------------------------
typedef const uint32_t __attribute__((address_space(3))) *LdsDwordPtr;
struct BitReader
{
    static constexpr bool kDirect = false;
    uint32_t pa;            // LDS byte address of the pool dword after `lo`
    uint32_t hi, lo;        // the window: the next 32 bits of the stream are ({hi,lo} >> negpos) & 0xFFFFFFFF
    int negpos;             // 0..31: unread bits of the window that lie below the next 32

    // A reader never runs more than about 1 KB past where it started (16 bands x 32 symbols x 16 bits), and in
    // the workgroup's LDS layout every bit pool is followed by at least 8 KB of other data (more pools, then the
    // tile rows), so even a reader driven by a record that does not match the bytes (only a caller-made one can
    // be) stays inside the allocation without a bound check.
    __device__ __forceinline__ void init(const uint32_t *pool, int bitInDword)
    {
        hi = 0;
        lo = pool[0];
        pa = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsDwordPtr)(pool + 1)));
        negpos = 0;
        skip(bitInDword);
    }
    // the next 32 bits, MSB first
    __device__ __forceinline__ uint32_t cur() const { return __builtin_amdgcn_alignbit(hi, lo, static_cast<uint32_t>(negpos)); }
    __device__ __forceinline__ uint32_t peek(int n) const { return cur() >> (32 - n); }     // n in 1..32
    // Branch-free advance, n in 0..32.  The pool dword a refill would pull in is read every time (an LDS read costs no
    // VALU issue): symbol loops request it at the top of the iteration with prefetch() and it is there when the
    // iteration's last instructions, the advance, need it -- the loop waits for its codebook read in between anyway.  (A
    // third window register that took the dword one refill ahead, so that nothing waited for it, cost one more select
    // per symbol and bought nothing: 20 instead of 19 instructions.)
    __device__ __forceinline__ uint32_t prefetch() const { return *reinterpret_cast<LdsDwordPtr>(static_cast<uintptr_t>(pa)); }
    __device__ __forceinline__ void skip(int n, uint32_t ahead)
    {
        negpos -= n;
        const bool refill = negpos < 0;
        hi = refill ? lo : hi;
        lo = refill ? ahead : lo;
        // the address moves on by 4 bytes on a refill: sign bit of negpos, shifted and added in one instruction
        const uint32_t sign = static_cast<uint32_t>(negpos) >> 31;
        asm("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(pa) : "v"(sign));
        negpos &= 31;
    }
    // the same in six instructions, for the hottest loop: the subtraction's borrow is the refill condition
    // (negpos is kept in 0..31, so "negpos < n" is the unsigned borrow)
    __device__ __forceinline__ void skipTight(uint32_t n, uint32_t ahead)
    {
        uint32_t t;
        asm("v_sub_co_u32 %3, vcc, %3, %5\n\t"
            "v_cndmask_b32 %0, %0, %1, vcc\n\t"
            "v_cndmask_b32 %1, %1, %6, vcc\n\t"
            "v_lshrrev_b32 %4, 31, %3\n\t"
            "v_lshl_add_u32 %2, %4, 2, %2\n\t"
            "v_and_b32 %3, 31, %3"
            : "+v"(hi), "+v"(lo), "+v"(pa), "+v"(negpos), "=&v"(t)
            : "v"(n), "v"(ahead)
            : "vcc");
    }
    __device__ __forceinline__ void skip(int n) { skip(n, prefetch()); }
    __device__ __forceinline__ uint32_t get(int n) { const uint32_t v = peek(n); skip(n); return v; }
    // where the reader stands, as an LDS bit address (lo is the dword at pa - 4 unless nothing of it has been read),
    // and a reader put there: what a band's error path needs to go over the band again
    __device__ __forceinline__ uint32_t bitAddr() const { return (pa - 4u) * 8u - static_cast<uint32_t>(negpos); }
    __device__ __forceinline__ void initAt(uint32_t bits)
    {
        init((const uint32_t *)reinterpret_cast<LdsDwordPtr>(static_cast<uintptr_t>((bits >> 5) << 2)), static_cast<int>(bits & 31u));
    }
};

// The same reader without a window held in registers: only the position is kept, every look reads the two pool
// dwords that hold the next 32 bits.  Five cheap instructions per symbol instead of the queue's eight and one register
// instead of five, at the price of an LDS round trip per look -- which the fixed-width sample loops of the 1993 formats
// take off the critical path, because the position of their next look does not depend on the data.  Measured: the 1993
// layouts gain at every size (1-2 % with 4 frames per wavefront, 4 % with 8), the 1994+ symbol loop, which has the round
// trip on its chain, gains only with 4 frames per wavefront (nothing with 8, 3 % slower with 16): it keeps the window
// in registers there.
struct DirectReader
{
    static constexpr bool kDirect = true;
    uint32_t bp;            // LDS BIT address of the next unread bit, minus one

    __device__ __forceinline__ void init(const uint32_t *pool, int bitInDword)
    {
        bp = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsDwordPtr)pool)) * 8u + static_cast<uint32_t>(bitInDword) - 1u;
    }
    // the next 32 bits, MSB first: with B bits consumed they are {W[j], W[j+1]} >> (-B & 31), j = (B - 1) >> 5
    // (before the first bit of a dword, j is the dword before it and the shift is 0: the funnel returns W[j+1])
    __device__ __forceinline__ uint32_t cur() const
    {
        const LdsDwordPtr w = reinterpret_cast<LdsDwordPtr>(static_cast<uintptr_t>((bp >> 3) & ~3u));
        return __builtin_amdgcn_alignbit(w[0], w[1], ~bp);
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return cur() >> (32 - n); }     // n in 1..32
    __device__ __forceinline__ uint32_t prefetch() const { return 0u; }                     // (nothing to fetch ahead)
    __device__ __forceinline__ void skip(int n) { bp += static_cast<uint32_t>(n); }
    __device__ __forceinline__ void skip(int n, uint32_t) { skip(n); }
    __device__ __forceinline__ void skipTight(uint32_t n, uint32_t) { bp += n; }
    __device__ __forceinline__ uint32_t get(int n) { const uint32_t v = peek(n); skip(n); return v; }
    __device__ __forceinline__ uint32_t bitAddr() const { return bp + 1u; }
    __device__ __forceinline__ void initAt(uint32_t bits) { bp = bits - 1u; }
};

// prefix code via first-level table + trie (dcs_common.h)
template <class BR>
__device__ __forceinline__ int readVlc(BR &br, const uint16_t *fast, const uint16_t *trie)
{
    uint32_t e = fast[br.peek(8)];
    if (e & 0x8000)
    {
        br.skip(static_cast<int>((e >> 8) & 0xF));
        return static_cast<int>(e & 0xFF);
    }
    // the rest of a long code (30 bits at most, dcs_tables.h) bit by bit, from ONE look at the stream: a reader round
    // trip per bit, on top of the trie's, is what made this path cost 2 900 cycles where any lane of the wavefront took it
    br.skip(8);
    uint32_t w = br.cur();
    int n = 0;
    do
    {
        e = trie[e + (w >> 31)];
        w <<= 1;
        ++n;
    }
    while (!(e & 0x8000) && n < 24);
    br.skip(n);
    return static_cast<int>(e & 0xFF);
}

// ------------------------------------------------------------------------------------------------
// per-lane views of the LDS working set
// ------------------------------------------------------------------------------------------------
template <int FPW>
struct Lds
{
    unsigned char *tab;         // workgroup-shared tables
    unsigned char *base;        // this wavefront's tile rows and tails
    unsigned char *poolBase;    // this wavefront's bit pool
    __device__ __forceinline__ const DcsLdsTables *tables() const { return reinterpret_cast<const DcsLdsTables *>(tab); }
    __device__ __forceinline__ uint16_t *row(int s) const { return reinterpret_cast<uint16_t *>(base + s * kRowBytes); }
    __device__ __forceinline__ uint16_t *tails() const { return reinterpret_cast<uint16_t *>(base + FPW * kRowBytes); }      // [FPW][16]
    __device__ __forceinline__ uint32_t *pool() const { return reinterpret_cast<uint32_t *>(poolBase); }
};

// what one sub-lane knows about the frame quarter it unpacks
struct Quarter
{
    uint32_t h0, h1, h2, h3;    // the stream header, 16 bytes (byte b = h[b >> 2] >> 8*(b & 3))
    uint32_t t0, t1, t2, t3;    // DcsFrameIndex.bandType, same packing
    int bandBase;           // first header band of this lane
    int nb;                 // number of bands this lane unpacks (0: idle)
    int outIdx;             // tile word index at the start of bandBase
    uint32_t preAdj;        // 1994+ Type 1, bands 0..2 (4 bits each)
    // 1993 formats: state carried into bandBase
    uint32_t prv, prvDelta;
    int subType;
    bool reuse, first;
    // 1994+, sixteen lanes per frame: band 15 shared by two lanes (dcsMid15).  midEnd: this lane's band 15 ends at the
    // middle; midStart: it starts there, midStraddle: one sample later (a two-zeros code ran across)
    bool midEnd, midStart, midStraddle;
};

// byte b (0..15) of four registers; explicit selects so that nothing is indexed in memory
__device__ __forceinline__ int byteOf(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int b)
{
    const int i = b >> 2;
    const uint32_t lo = (i & 1) ? w1 : w0, hi = (i & 1) ? w3 : w2;
    const uint32_t w = (i & 2) ? hi : lo;
    return static_cast<int>((w >> (8 * (b & 3))) & 0xFFu);
}

// 24 x 24 -> low 32 bits, always the full-rate instruction
__device__ __forceinline__ int mul24(int a, int b)
{
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the 32-bit "splice" multiply-accumulate of the mixer (.cpp:2244-2250, :2434-2443): low word =
// scaled sample, high word = accumulator, add (int16)scaled * (uint16)mixMul, keep the high word.
// FIRST: the row is still zero at idx (first source of the frame; no index is written twice by one
// source), so the read of the accumulator is skipped.
template <bool FIRST>
__device__ __forceinline__ void mixAdd(uint16_t *cell, int scaledProduct, uint32_t mixMul)
{
    // three instructions: the low word of the product is picked (sign- resp. zero-extended) by the operand selects
    uint32_t t, acc;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(t) : "v"(scaledProduct), "v"(mixMul));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(acc) : "v"(t), "v"(scaledProduct));
    if (!FIRST)
        acc += static_cast<uint32_t>(*cell) << 16;
    *cell = static_cast<uint16_t>(acc >> 16);
}
// ... with the cell given as an LDS byte address
typedef uint16_t __attribute__((address_space(3))) *LdsWordPtr;
template <bool FIRST>
__device__ __forceinline__ void mixAddAt(uint32_t cellAddr, int scaledProduct, uint32_t mixMul)
{
    const LdsWordPtr cell = reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(cellAddr));
    uint32_t t, acc;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(t) : "v"(scaledProduct), "v"(mixMul));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(acc) : "v"(t), "v"(scaledProduct));
    if (!FIRST)
        acc += static_cast<uint32_t>(*cell) << 16;
    *cell = static_cast<uint16_t>(acc >> 16);
}
// the same with the low word that is ADDED taken from `addLow` instead of the product (the 1993 "repeat the previous
// input" coding carries it from sample to sample, :2513-2534); returns the 32-bit sum
template <bool FIRST>
__device__ __forceinline__ uint32_t mixAddCarry(uint16_t *cell, int scaledProduct, uint32_t mixMul, uint32_t addLow)
{
    uint32_t t, acc;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(t) : "v"(scaledProduct), "v"(mixMul));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(acc) : "v"(t), "v"(addLow));
    if (!FIRST)
        acc += static_cast<uint32_t>(*cell) << 16;
    *cell = static_cast<uint16_t>(acc >> 16);
    return acc;
}
// ... and with the cell given as an LDS byte address
template <bool FIRST>
__device__ __forceinline__ uint32_t mixAddCarryAt(uint32_t cellAddr, int scaledProduct, uint32_t mixMul, uint32_t addLow)
{
    const LdsWordPtr cell = reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(cellAddr));
    uint32_t t, acc;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(t) : "v"(scaledProduct), "v"(mixMul));
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(acc) : "v"(t), "v"(addLow));
    if (!FIRST)
        acc += static_cast<uint32_t>(*cell) << 16;
    *cell = static_cast<uint16_t>(acc >> 16);
    return acc;
}
// (int16)word 0 of x times a 24-bit signed y, one instruction
__device__ __forceinline__ int mulLowWord(uint32_t x, int y)
{
    int r;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// the same contribution removed again (exact inverse: the MAC is additive modulo 2^16 in the high word)
__device__ __forceinline__ void mixSub(uint16_t *cell, int scaledProduct, uint32_t mixMul)
{
    const uint32_t s = static_cast<uint32_t>(scaledProduct) & 0xFFFFu;
    const uint32_t c = (s + static_cast<uint32_t>(__mul24(sx16(s), static_cast<int>(mixMul)))) >> 16;
    *cell = static_cast<uint16_t>(*cell - c);
}

// scale factor of a band (:1978-1979, :2342): mantissa {0x8000, 0x9838, 0xB505, 0xD745}[code & 3] >> (15 - ((code >> 2) & 15)),
// one read of the 64-entry table (ten instructions as arithmetic on two register constants; the read's latency is hidden
// where several wavefronts share the SIMD and costs nothing measurable where one has it to itself)
__device__ __forceinline__ uint32_t scaleFactor(const DcsLdsTables *T, int code)
{
    return T->scale64[code & 63];
}

__device__ __forceinline__ void dcFixup(uint16_t *row, uint32_t saved1)
{
    const int delta = sat16(sx16(row[1]) - sx16(saved1));
    row[0] = static_cast<uint16_t>(sat16(delta + sx16(row[0])));
    row[1] = static_cast<uint16_t>(saved1);
}

constexpr int kDummyWord = 256;         // the pad word of a tile row: sink for predicated-off stores

// ------------------------------------------------------------------------------------------------
// a2: 1994+ frame (DecoderImpl94x::DecompressFrame, .cpp:1679-2261)
//
// Each lane unpacks Q.nb consecutive header bands.  The frame header (band-type deltas, :1780-1834)
// was already resolved by the index pass: Q.types holds this frame's codes, Q.preAdj the scale
// pre-adjust that depends on the PREVIOUS frame (:1744-1773).  The per-band set-up is ordinary
// divergent code; the symbol loop is branch-free so that lanes with Huffman-coded bands, raw bands
// and different codebooks all execute the same instruction stream.
// ------------------------------------------------------------------------------------------------
template <bool FIRST, class BR, int SUB>
__device__ uint32_t unpack94(const DcsLdsTables *T, uint16_t *row, BR &br, const Quarter &Q,
                             int format, uint32_t mixMul, bool has, const Stamper &stamp)
{
    const bool type1 = format != DCS_FMT_94_T0;
    uint32_t err = 0;
    int nb = has ? Q.nb : 0;
    // the most this lane's bands can advance the output index (samples per band, :1848-1850): with the index
    // pass's own records start + advance <= 256 always holds, so the clamp only bites on a caller-made record
    // that does not belong to the stream and keeps such a lane inside its frame's row
    const int b1 = Q.bandBase + nb;
    const int before = Q.bandBase <= 0 ? 0 : Q.bandBase == 1 ? 7 : 15 + 16 * (Q.bandBase - 2);
    const int upTo = b1 <= 0 ? 0 : b1 == 1 ? 7 : b1 >= 16 ? 255 : 15 + 16 * (b1 - 2);
    // (a lane that starts in the middle of band 15 has sixteen tile words to go at most; one sample fewer -- one word, or
    // two in a strided band, header bit 6 -- behind a straddling code)
    const int midAdvance = 16 - (Q.midStraddle ? 1 + static_cast<int>((Q.h3 >> 30) & 1u) : 0);
    int outIdx = min(Q.outIdx, 256 - ((SUB == 16 && Q.midStart) ? midAdvance : upTo - before));
    bool valid = true;
    const bool owner = has && Q.bandBase == 0;          // the lane that holds band 0 does the DC fix-up
    const uint32_t saved1 = (owner && !FIRST) ? row[1] : 0u;

    stamp(8);
    // The lanes of a wavefront run the symbol loop together, and a lane's bands differ in length (7, 8, 13 x 16 and 32
    // samples), so the loop is run in ROUNDS of at most 7, 9, then 16 samples per lane at a time, and only
    // between rounds does a lane whose band has run out set the next one up (the set-up is some 60 instructions for the
    // whole wavefront however many lanes need it).  A band that is longer than the round simply goes on in the next one.
    // With the bands dealt out as {0, 1, 2} {3, 4} ... {13, 14} {15} (eight lanes per frame, dcsLaneFirstBand) every
    // lane has 31 or 32 samples and the wavefront is through after 7 + 9 + 16 iterations; one band per lane per round
    // (round 1's form) cost 16 + 32.
    int k = 0;                                  // bands this lane has started
    const int keyLim = type1 ? 16 : 17;         // band-type codes beyond this all mean the same (no such code)
    // (cells as LDS byte addresses: 32-bit arithmetic in the loop)
    uint32_t cell = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsWordPtr)row)) + 2u * static_cast<uint32_t>(outIdx);   // where the next sample goes
    uint32_t cellEnd = cell;                    // end of the band in progress; == cell: no band in progress
    uint32_t cellStart = cell;                  // ... its start and where its bits began, for the error path
    uint32_t startBits = 0;
    const uint16_t *book = T->cb94;
    int shPeek = 0, shIdx = 0, scale = 0;
    uint32_t incSh = 1;                         // log2 of the bytes from one sample of the band to the next (strided: 4)
    uint32_t valOff = 0, valWidth = 8;          // where the sample sits in (window's high half | codebook entry)
    for (int round = 0 ; ; ++round)
    {
        // ---- a band that overshot its end: a two-zeros code with one sample left (:2213-2218) --------------------
        if (SUB == 16 && Q.midEnd && cell > cellEnd)
            cellEnd = cell;                     // (not an error: a two-zeros code across the middle of band 15, where this lane stops)
        if (cell > cellEnd)
        {
            // the reference zeroes the WHOLE band buffer on this error (:2238-2239): take back what the
            // band already contributed by replaying it.  (Frames with errors are never split.)
            cell -= 1u << incSh;                // the second zero had no room: the band ends where it should
            if (valid)
            {
                BR r2;
                r2.initAt(startBits);
                uint32_t c2 = cellStart;
                while (c2 + (1u << incSh) < cellEnd)            // (more than one sample left)
                {
                    const uint32_t e = book[r2.cur() >> shIdx];
                    r2.skip(static_cast<int>((e >> 8) & 0x1F));
                    const uint32_t step = e >> 13;
                    if (step == 1)
                        mixSub((uint16_t *)reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(c2)),
                               mul24(static_cast<int>(static_cast<int8_t>(e & 0xFF)), scale), mixMul);
                    c2 += step << incSh;
                }
            }
            valid = false; err |= DCS_FRAME_STOP;
            cellEnd = cell;
        }
        // ---- per-band set-up for the lanes that are between bands ---------------------------------------------
        const bool start = cell == cellEnd && k < nb;
        if (!__any(start || cell < cellEnd))
            break;
        if (start)
        {
            const int band = Q.bandBase + k;
            ++k;
            // what the band-type code means comes out of one table (DcsLdsTables.band94): codebook, look-ahead width,
            // fixed-width or not, scale adjustment, and whether the band is empty or in error
            const int hb = byteOf(Q.h0, Q.h1, Q.h2, Q.h3, band) & 0x7F;
            const int code0 = byteOf(Q.t0, Q.t1, Q.t2, Q.t3, band);
            const int cls = band < 3 ? 0 : band < 6 ? 17 : 34;
            const int key = min(code0, keyLim) + (type1 ? cls : DCS_B94_TYPE0);
            const uint32_t e = T->band94[key];
            // samples of the band (:1848-1850): 7, 8, 16 ... 16, 32; halved for a strided band
            const int count0 = band < 2 ? 7 + band : band == 15 ? 32 : 16;
            const uint32_t strided = (static_cast<uint32_t>(hb) >> 6) & 1u;
            const uint32_t count = static_cast<uint32_t>(count0) >> strided;
            incSh = 1u + strided;
            // Type 1: scale = header byte + pre-adjust (bands 0..2, from the previous frame's codes) + the code's own (:1914-1961)
            const int pre = (type1 && band < 3) ? static_cast<int>((Q.preAdj >> (4 * band)) & 15u) : 0;
            scale = static_cast<int>(scaleFactor(T, hb + pre + static_cast<int>(e >> 25)));
            book = reinterpret_cast<const uint16_t *>(T) + (e & 0x7FFu);
            shPeek = static_cast<int>((e >> 11) & 31u);
            shIdx = static_cast<int>((e >> 16) & 31u);
            const bool isRaw = (e & DCS_B94_RAW) != 0;
            valOff = isRaw ? static_cast<uint32_t>(shPeek) : 0u;
            valWidth = isRaw ? 32u - static_cast<uint32_t>(shPeek) : 8u;
            // (no band-type code stands for sample code 0 -- dcs_tables.cpp checks it --, so the STOP of :1985-1991 cannot
            // happen: a band is empty, in error, or has `count` samples)
            const bool zeroBand = (e & DCS_B94_ZERO) != 0, fatal = (e & DCS_B94_FATAL) != 0;
            uint32_t i = (e & (DCS_B94_ZERO | DCS_B94_FATAL)) == 0 ? count : 0u;      // symbols to decode in this band
            if (SUB == 16 && band == 15 && i != 0)
            {
                // band 15 shared by two lanes: the first stops at the first code boundary with half of the samples done,
                // the second starts there (its output index came with its record)
                if (Q.midEnd) i = count - (count >> 1);
                if (Q.midStart) i = (count >> 1) - (Q.midStraddle ? 1u : 0u);
            }
            // (a band without a code moves on by the halved count, not by count * inc, :1886)
            cell += zeroBand ? 2u * count : 0u;
            if (fatal)
            {
                err |= DCS_FRAME_FATAL | DCS_FRAME_STOP;
                nb = 0;                         // stop: later bands contribute nothing
            }
            if (!valid)
                scale = 0;              // after a STOP the band is still parsed, its samples contribute nothing
            cellStart = cell;
            cellEnd = cell + (i << incSh);
            startBits = br.bitAddr();
        }

        // ---- symbol loop, branch-free ------------------------------------------------------------------
        // One codebook entry drives everything: value, code length and how many samples it stands for
        // (the "two zeros" code, :2200-2212, has value 0 and step 2, so it needs no special store: adding a
        // zero product leaves the accumulator as it was).  A two-zeros code with one sample left (:2213-2218)
        // drives the cell past cellEnd, which is how the error is seen at the top of the next round.
        if (round == 0) stamp(9);
        // (the round is bounded in samples, not iterations: a symbol is at least one sample, and the loop's only test
        // stays the cell against an end)
        const uint32_t roundLen = round == 0 ? 7u : round == 1 ? 9u : 16u;
        const uint32_t roundEnd = min(cellEnd, cell + (roundLen << incSh));
        if (cell < roundEnd)
        {
            do
            {
                const uint32_t ahead = br.prefetch();
                const uint32_t w = br.cur();
                const uint32_t e = book[w >> shIdx];
                // the sample: the entry's low byte, or the window's top `width` bits -- one bit-field extract from
                // (window's high half | entry), at a position and width that are the band's
                const int v = __builtin_amdgcn_sbfe(__builtin_amdgcn_perm(w, e, 0x07060100u), valOff, valWidth);
                const uint32_t step = e >> 13;
                br.skipTight((e >> 8) & 0x1Fu, ahead);
                mixAddAt<FIRST>(cell, mul24(v, scale), mixMul);
                cell += step << incSh;
            }
            while (cell < roundEnd);
        }
        if (round == 0) stamp(10);
    }

    if (owner)
        dcFixup(row, saved1);
    return err;
}

// ------------------------------------------------------------------------------------------------
// a3: 1993 frame, Type 0 and OS93b Type 1 (DecoderImpl93::DecompressFrame + ReadHuff93, .cpp:2293-2684)
//
// The three sample codings of a band (direct / delta / double delta, :2565-2599) are folded into one
// branch-free update:  d = in + (st == 2 ? prvDelta : 0);  p = d + (st == 0 ? 0 : prv);
// prvDelta' = st == 0 ? p - prv : d;  prv' = p.  A code-0 band of sub-type 2 (ramp, :2539-2545) is the
// same update with in = 0, so it rides in the same loop without reading bits.
// ------------------------------------------------------------------------------------------------
template <bool FIRST, class BR>
__device__ uint32_t unpack93(const DcsLdsTables *T, uint16_t *row, BR &br, const Quarter &Q,
                             int format, uint32_t mixMul, bool has, const Stamper &stamp)
{
    const bool type1 = format == DCS_FMT_93B_T1;
    uint32_t err = 0;
    int nb = has ? Q.nb : 0;
    const bool owner = has && Q.bandBase == 0;
    const uint32_t saved1 = (owner && !FIRST) ? row[1] : 0u;

    int subType = Q.subType;
    bool first = Q.first, reuse = Q.reuse;
    uint32_t prv = Q.prv, prvDelta = Q.prvDelta;        // uint16 semantics: masked on use
    int code = 0;
    // where the next sample goes, as an LDS byte address.  Clamped to the row's pad word at every store; once there it
    // stays there (every later update adds at least as much as the one correction below takes back), which is what an
    // unclamped index beyond the row amounts to.
    const uint32_t rowAddr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsWordPtr)row));
    const uint32_t cellLim = rowAddr + 2u * kDummyWord;
    uint32_t cellAddr = rowAddr + 2u * static_cast<uint32_t>(Q.outIdx);

    stamp(8);
    for (int k = 0 ; __any(k < nb) ; ++k)
    {
        const int band = Q.bandBase + k;
        // ---- per-band set-up: straight-line code with selects (lanes without a band end up with nothing to do);
        // only the OS93b Type-1 band-type code, a variable-length code, sits behind a (wave-uniform) branch ----------
        const bool act = k < nb;
        const int hb = byteOf(Q.h0, Q.h1, Q.h2, Q.h3, band) & 0x7F;
        const int scale = static_cast<int>(scaleFactor(T, hb));
        const bool strided = (hb >> 6) != 0;
        // Type 0: 16 samples; a strided band starts one word further on, advances by 2 and ends one word back (:2362-2366).
        // Type 1: 16 samples (15 in the frame's first band), or 8 at stride 2 (:2379-2381)
        const uint32_t incBytes = strided ? 4u : 2u;
        const int nSamples = !type1 ? 16 : strided ? 8 : first ? 15 : 16;
        const int stride = !type1 ? (strided ? 31 : 16) : nSamples;
        const uint32_t fixupBytes = (!type1 && strided) ? 0xFFFFFFFEu : 0u;       // one word back
        cellAddr += (act && !type1 && strided) ? 2u : 0u;

        // the band-type field (:2388-2419): [reuse bit, if the previous band was code 0] then, Type 0:
        // [change sub-type] [direction, if changing] [4-bit code].  Parsed from ONE window read with one advance.
        const uint32_t bw = br.cur();
        const bool hadReuse = reuse;
        const bool reuseNow = hadReuse && (bw >> 31) != 0;
        const bool parse0 = act && !reuseNow && !type1;
        const uint32_t b = bw << (hadReuse ? 1 : 0);
        const bool change = parse0 && (b >> 31) != 0;
        const bool up = ((b >> 30) & 1u) != 0;
        const int subUp = subType == 2 ? 0 : subType + 1, subDown = subType == 0 ? 2 : subType - 1;
        subType = change ? (up ? subUp : subDown) : subType;
        const int fixedBits = change ? 2 : 1;
        code = parse0 ? static_cast<int>((b << fixedBits) >> 28) : code;
        br.skip(!act ? 0 : (hadReuse ? 1 : 0) + (parse0 ? fixedBits + 4 : 0));
        reuse = act ? reuseNow : reuse;
        const bool vlc = act && !reuseNow && type1;
        if (__any(vlc))
        {
            if (vlc)
            {
                int v = readVlc(br, T->fast93, T->trie93);
                if (v < 0x1E)
                    v -= 0x0F;
                else
                {
                    v -= 0x2E;
                    subType = subType != 0 ? 0 : 1;
                }
                code = (byteOf(Q.t0, Q.t1, Q.t2, Q.t3, band) + v) & 0xFFFF;
            }
        }

        const bool zero = act && code == 0;
        reuse = zero ? true : reuse;
        const bool skipBand = zero && subType == 0;                     // nothing coded, previous input forgotten
        // repeat the previous input: rides in the main loop as a ramp with step 0 (no bits read, previous input kept),
        // except that the low word of the product is carried from sample to sample instead of being reloaded (:2513-2534)
        const bool quirk = zero && subType == 1;
        const int w0 = code + (type1 ? 0 : 1);
        const bool fatal = act && code != 0 && w0 > 16;
        const int width = (zero || fatal || !act) ? 0 : w0;            // bits per input (0: ramp, no bits read)
        const int nS = (!act || fatal || skipBand) ? 0 : nSamples;     // samples through the main loop
        cellAddr += skipBand ? 2u * static_cast<uint32_t>(stride) : 0u;
        prv = skipBand ? 0u : prv;
        prvDelta = skipBand ? 0u : prvDelta;
        if (fatal)
        {
            err |= DCS_FRAME_FATAL | DCS_FRAME_STOP;
            nb = 0;
        }
        first = act ? false : first;

        // ---- main sample loop, branch-free -----------------------------------------------------------
        // (three multiply-adds with 0 / 1 / -1 instead of the six operations below were measured: slower)
        const uint32_t m2 = subType == 2 ? 0xFFFFu : 0u;        // add the previous delta
        const uint32_t m0 = subType == 0 ? 0u : 0xFFFFu;        // add the previous value
        // width 0 (a ramp: no bits read) rides along with an all-zero mask instead of a branch
        const int shW = (32 - width) & 31;
        const uint32_t wMask = width != 0 ? 0xFFFFFFFFu : 0u;
        const bool ran = nS > 0;
        if (k == 0) stamp(9);
        uint32_t carry = static_cast<uint32_t>(mulLowWord(prv, scale));      // (only lanes that repeat the previous input use it)
        auto sampleAt = [&](uint32_t in, auto clampTag)
        {
            constexpr bool clamp = decltype(clampTag)::value;
            const uint32_t d = in + (prvDelta & m2);
            const uint32_t p = d + (prv & m0);
            prvDelta = d - (prv & ~m0);
            prv = p;
            const int prod = mulLowWord(p, scale);
            if (clamp)
                cellAddr = min(cellAddr, cellLim);
            carry = mixAddCarryAt<FIRST>(cellAddr, prod, mixMul, quirk ? carry : static_cast<uint32_t>(prod));
            cellAddr += incBytes;
        };
        auto sample = [&](uint32_t in) { sampleAt(in, std::true_type()); };
        // two samples per window read: a sample is at most 16 bits wide, so the next 32 bits always hold two
        int i = 0;
        if constexpr (BR::kDirect)
        {
            // the position of the next pair does not depend on this one's bits: its window is requested an iteration ahead
            uint32_t wNext = br.cur();
            auto pairAt = [&](auto clampTag)
            {
                const uint32_t w = wNext & wMask;
                br.skip(2 * width);
                wNext = br.cur();
                sampleAt(static_cast<uint32_t>(static_cast<int>(w) >> shW), clampTag);
                sampleAt(static_cast<uint32_t>(static_cast<int>(w << width) >> shW), clampTag);
            };
            // Every band of a Type-0 frame has 16 samples: when all the lanes that have a band at all have that many and
            // room for them in their rows (a wave-uniform test; always, with records of the index pass), the eight pairs
            // run as straight-line code, without the loop's bookkeeping and branch and without the clamp of the address
            if (__all((nS == 16 && cellAddr + 16u * incBytes <= cellLim) || nS == 0))
            {
                if (nS == 16)
                {
#pragma unroll
                    for (int u = 0 ; u < 8 ; ++u)
                        pairAt(std::false_type());
                    i = 16;
                }
            }
            else
                for ( ; i + 2 <= nS ; i += 2)
                    pairAt(std::true_type());
        }
        else
        {
            for ( ; i + 2 <= nS ; i += 2)
            {
                const uint32_t ahead = br.prefetch();
                const uint32_t w = br.cur() & wMask;
                sample(static_cast<uint32_t>(static_cast<int>(w) >> shW));
                sample(static_cast<uint32_t>(static_cast<int>(w << width) >> shW));
                br.skipTight(static_cast<uint32_t>(2 * width), ahead);     // last: the prefetched dword has had the whole iteration to arrive
            }
        }
        if (__any(i < nS))
        {
            if (i < nS)
            {
                const uint32_t in = static_cast<uint32_t>(static_cast<int>(br.cur() & wMask) >> shW);
                br.skip(width);
                sample(in);
            }
        }
        prv &= 0xFFFFu; prvDelta &= 0xFFFFu;
        if (k == 0) stamp(10);

        cellAddr += ran ? fixupBytes : 0u;
    }
    stamp(11);

    if (owner)
        dcFixup(row, saved1);
    return err;
}

// ------------------------------------------------------------------------------------------------
// a4: OS93a Type 1 frame (DecoderImpl93a::DecompressFrame, .cpp:2831-3032).  A lane takes the bands
// [bandBase, bandEnd); what it needs from the bands before it -- the previous scale code and whether the
// frame already ended -- comes from the split record.
// ------------------------------------------------------------------------------------------------
template <bool FIRST, class BR, class PairPtr>
__device__ __forceinline__ uint32_t unpack93a(const DcsLdsTables *T, uint16_t *row, BR &br, int hb, uint32_t mixMul,
                                              PairPtr pairTable, int bandBase, int bandEnd, int prvScale, int outIdx)
{
    const uint16_t *bbBook = &T->bandBits93a[(hb & 0x60) >> 1];
    uint32_t err = 0;

    for (int band = bandBase ; band < bandEnd ; ++band)
    {
        if (band >= 18) { err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; break; }
        const int numInputs = T->inputs93a[band];

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

        uint32_t sc = T->scaleCb93a[br.peek(4)];
        br.skip(static_cast<int>((sc >> 8) & 0xF));
        if ((sc & 0xFF) == 0xFF)
        {
            sc = T->scaleCb93a[((sc >> 12) << 4) + br.peek(4)];
            br.skip(static_cast<int>((sc >> 8) & 0xF) - 4);
        }

        int scaleCode = prvScale + static_cast<int>(sc & 0xFF) - 1 + bandBits * 2;
        if (scaleCode > 0x39)
            scaleCode -= 0x36;
        prvScale = scaleCode - bandBits * 2;

        // 0x8000 times (0x9838 / 2^15)^(code & 3), each step truncated (:2986-2990): four constants
        constexpr uint32_t kS1 = (0x8000u * 0x9838u) >> 15, kS2 = (kS1 * 0x9838u) >> 15, kS3 = (kS2 * 0x9838u) >> 15;
        uint32_t sf = (scaleCode & 2) ? ((scaleCode & 1) ? kS3 : kS2) : ((scaleCode & 1) ? kS1 : 0x8000u);
        sf <<= (scaleCode >> 2);
        sf = ((sf >> 16) * mixMul) >> 15;
        const int sfs = sx16(sf);                   // truncated to 16 bits, then read as signed (:2995, :3011)

        const auto pairBase = pairTable + (2 << bandBits);
        for (int i = 0 ; i < numInputs ; ++i)
        {
            // (a pair sits at an even index of the table: one 32-bit read)
            const uint32_t pairWord = *reinterpret_cast<const uint32_t *>(pairBase + 2 * br.get(bandBits));
            for (int k = 0 ; k < 2 ; ++k, ++outIdx)
            {
                const int p = mul24(k == 0 ? sx16(pairWord) : static_cast<int>(pairWord) >> 16, sfs);
                uint16_t *cell = &row[min(outIdx, kDummyWord)];                 // (254 words at most with a matching record)
                // (first source of the frame: the accumulator is still zero, no index is written twice by one source)
                const uint32_t acc = FIRST ? 0u : static_cast<uint32_t>(*cell) << 16;
                *cell = static_cast<uint16_t>(roundHi(acc + (static_cast<uint32_t>(p) << 1), p));
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

// (straight-line code with selects: it runs on one lane per frame while the rest of the wavefront waits)
__device__ __forceinline__ void dcMagnitude93(uint16_t *row)
{
    const uint32_t w = *reinterpret_cast<const uint32_t *>(row);            // row[0] | row[1] << 16 (rows are 16-byte aligned)
    const int f0 = sx16(w), f1 = static_cast<int>(w) >> 16;
    const bool neg = f0 < 0;
    const int a0 = sx16(static_cast<uint32_t>(neg ? -f0 : f0));             // 16-bit negate: -(-32768) stays -32768
    uint32_t sr = prodSS(f1, f1) + prodSS(a0, a0);
    int exponent = calcExp32(sr);                                           // <= 0
    sr <<= -exponent;
    const int x = sx16(sr >> 16);
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
    // odd exponent: one more factor of sqrt(1/2), exponent made even; then exponent / 2 + 1 (C++ division of the
    // reference: exact here, the value is even)
    const bool odd = (exponent & 1) != 0;
    const uint32_t mrOdd = mulRound(static_cast<int>(mr) >> 16, 0x5A82);
    mr = odd ? mrOdd : mr;
    exponent = ((exponent + (odd ? 1 : 0)) >> 1) + 1;
    const uint32_t up = mr << (exponent & 31);
    const uint32_t down = static_cast<uint32_t>(static_cast<int>(mr) >> ((-exponent) & 31));   // arithmetic for negatives
    uint32_t ar = (exponent >= 0 ? up : down) >> 16;
    ar = neg ? static_cast<uint32_t>(-sx16(ar)) & 0xFFFFu : ar;
    ar = x != 0 ? ar : 0u;                                                  // a zero magnitude stays zero (:655)
    *reinterpret_cast<uint32_t *>(row) = ar;                                // row[0] = magnitude, row[1] = 0
}

// ------------------------------------------------------------------------------------------------
// phase 2: register-resident inverse transforms.
//
// A frame's complex points are spread over LPF lanes x 16 registers (a point is one dword, low half
// real, high half imaginary): 8 lanes per frame for the 1994+ transform (128 points), 16 lanes per
// frame for the 1993 one (256 points), so one wavefront transforms G = 8 (or 4) frames per pass.
// In layout A a lane holds the points whose LOW index bits equal its lane number, so the first
// radix-2 stages (partner distance >= 8 resp. 16 points) stay inside the lane; one transpose through
// LDS gives layout B (16 CONSECUTIVE points per lane) for the remaining stages.  Twiddles of the
// layout-A stages are the same for every lane (scalar loads); those of layout B are per-lane constants.
// No cross-lane traffic other than the two LDS transposes, one wave-level sync each.
// ------------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pkAddSat(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pkSubSat(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pkAdd(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, static_cast<u16x2>(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pkSub(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, static_cast<u16x2>(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pkAshr(uint32_t a, uint32_t shiftPair)
{ return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) >> __builtin_bit_cast(s16x2, shiftPair))); }

__device__ __forceinline__ uint32_t packC(int re, int im) { return (static_cast<uint32_t>(re) & 0xFFFFu) | (static_cast<uint32_t>(im) << 16); }
__device__ __forceinline__ int reC(uint32_t c) { return sx16(c); }
__device__ __forceinline__ int imC(uint32_t c) { return static_cast<int>(c) >> 16; }

// ---- complex rotate t = a * (c + i*s) with the reference's rounding (.cpp:500-506, :761-765) -----------------
//
//   t.re = RoundMultiplyResult(MR = 2 a.re c - 2 a.im s),   the rounding quirk keyed on the LAST product, a.im s
//   t.im = RoundMultiplyResult(MR = 2 a.im c + 2 a.re s),   keyed on a.re s
//
// RoundMultiplyResult (.cpp:3503-3514) adds 0x8000 and clears bit 16 when the last product's low word (of the doubled
// product) is exactly 0x8000.  That happens for one product in 32 768.  A transform stage therefore computes its
// butterflies WITHOUT the clear while one instruction per butterfly watches for the condition, and the stage is done
// again with the exact (slower) rotate in the rare case that any lane met it: one wave-uniform branch per stage,
// almost never taken, and the butterflies of a stage stay in one basic block.  "Low word of 2p is 0x8000" is tested
// on 2p + 0x8000 (for one of the two products a term the sum needs anyway): its low word is zero; the watch register
// keeps the minimum of all those low words (it starts at 0xFFFF and is set back to that when a stage is repeated).
// running minimum of the low words seen so far: zero as soon as one product met the condition
__device__ __forceinline__ void quirkWatch(uint32_t &watch, uint32_t a, uint32_t b)
{
    // (two two-operand minima: v_min3_u16 would do it in one instruction but runs at half rate on gfx950 -- 8.2 cycles
    // against 2 x 2.2 where several wavefronts share the SIMD, tools/valu_latency.hip)
    asm("v_min_u16 %0, %0, %1\n\tv_min_u16 %0, %0, %2" : "+v"(watch) : "v"(a), "v"(b));
}
__device__ __forceinline__ bool quirkSeen(uint32_t watch) { return __any((watch & 0xFFFFu) == 0); }
// 16 x 16 -> 32 signed products of selected halves of packed registers, one instruction each
#define DCS_MUL_SEL(dst, x, xsel, y, ysel) \
    asm("v_mul_i32_i24_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" xsel " src1_sel:" ysel : "=v"(dst) : "v"(x), "v"(y))

// registers every butterfly needs: 0x8000 and 0x10000 in VGPRs (VOP3 takes no literal) and the watch register
struct BflyRegs
{
    uint32_t k8000, k10000; uint32_t watch;
    uint32_t k4000, watchB;         // layout-B stages: rounding constant of the two-product sums, their watch (see rotateDotB)
};

// twiddles of the layout-A stages (entries 2..7 of the table; 0 and 1 are exact and need no multiplier): the same for
// every lane, held doubled (DcsDevTables.twA), which makes the products come out as the reference's doubled MR terms
struct TwScalar { int c2, s2, ns2; };
struct TwA { TwScalar t[8]; };
// The six of them are constants of the format (cos, sin of entries 2..7 of the reference's twiddle table, in 1.15,
// doubled; dcs_tables.cpp checks them against the table it uploads): written into the code they need no load, no
// registers across phase 1 and no spills -- as eighteen scalar loads at the kernel's start they were spilled to
// vector-register lanes and read back, forty instructions per wavefront.
constexpr int kTwADoubled[8][2] = { { -65536, 0 }, { 0, -65536 }, { -46340, -46340 }, { 46340, -46340 },
                                    { -60548, -25080 }, { 25080, -60548 }, { -25080, -60548 }, { 60548, -25080 } };
__device__ __forceinline__ void loadTwA(const DcsDevTables *, TwA &W)
{
#pragma unroll
    for (int k = 2 ; k < 8 ; ++k)
        W.t[k] = TwScalar{ kTwADoubled[k][0], kTwADoubled[k][1], -kTwADoubled[k][1] };
}

// exact rotate, every case handled in line: 15 VALU
__device__ __forceinline__ uint32_t rotateExact(uint32_t A, int c, int s)
{
    const int are = reC(A), aim = imC(A);
    const int p2 = __mul24(aim, s);
    const int dR = __mul24(are, c) - p2;
    const int q2 = __mul24(are, s);
    const int dI = __mul24(aim, c) + q2;
    uint32_t mrR = (static_cast<uint32_t>(dR) << 1) + 0x8000u;
    uint32_t mrI = (static_cast<uint32_t>(dI) << 1) + 0x8000u;
    if ((p2 & 0x7FFF) == 0x4000) mrR &= ~0x10000u;
    if ((q2 & 0x7FFF) == 0x4000) mrI &= ~0x10000u;
    return __builtin_amdgcn_perm(mrI, mrR, 0x07060302u);       // (mrR >> 16) | (mrI & 0xFFFF0000)
}

// fast rotates: the result without the clear, `quirk` collects the lanes that would have needed it.
// Doubled twiddle in registers common to all lanes: 2 unpack + 4 multiply-adds + 1 watch + 1 pack = 8 instructions
__device__ __forceinline__ uint32_t rotateFastA(uint32_t A, const TwScalar &t, BflyRegs &R)
{
    const int are = reC(A), aim = imC(A);
    const uint32_t nP = static_cast<uint32_t>(__mul24(aim, t.ns2)) + R.k8000;         // 0x8000 - 2 a.im s
    const uint32_t mrR = static_cast<uint32_t>(__mul24(are, t.c2)) + nP;
    const uint32_t qK = static_cast<uint32_t>(__mul24(are, t.s2)) + R.k8000;          // 0x8000 + 2 a.re s
    const uint32_t mrI = static_cast<uint32_t>(__mul24(aim, t.c2)) + qK;
    quirkWatch(R.watch, nP, qK);
    return __builtin_amdgcn_perm(mrI, mrR, 0x07060302u);
}
// (a << 1) + b in one instruction, kept as written (the compiler would re-associate the sums below into one more add)
__device__ __forceinline__ uint32_t shl1Add(uint32_t a, uint32_t b)
{
    uint32_t r = (a << 1) + b;
    asm("" : "+v"(r));              // (no instruction: only hides the sum from re-association)
    return r;
}
// per-lane twiddle, packed cos | sin << 16: 4 products + 4 adds + 1 watch + 1 pack = 10 instructions.
// TIGHT keeps the sums exactly as written (one instruction fewer); measured, that pays where several wavefronts share
// a SIMD (the saturating 1994+ stages use it), while a wavefront alone on its SIMD -- the 4 096-frame 1993 batch --
// runs 2 % faster with the compiler's own association of the sums.
template <bool TIGHT>
__device__ __forceinline__ uint32_t rotateFastB(uint32_t A, uint32_t tw, BflyRegs &R)
{
    int p1, p2, q1, q2;
    DCS_MUL_SEL(p2, A, "WORD_1", tw, "WORD_1");         // a.im s
    DCS_MUL_SEL(p1, A, "WORD_0", tw, "WORD_0");         // a.re c
    DCS_MUL_SEL(q2, A, "WORD_0", tw, "WORD_1");         // a.re s
    DCS_MUL_SEL(q1, A, "WORD_1", tw, "WORD_0");         // a.im c
    // MR.re = 2 p1 - 2 p2 + 0x8000 = (2 p1 + 0x10000) - (2 p2 + 0x8000);  MR.im = 2 q1 + (2 q2 + 0x8000): the terms in
    // parentheses are the ones the watch looks at
    if (TIGHT)
    {
        const uint32_t tR = shl1Add(static_cast<uint32_t>(p2), R.k8000);
        const uint32_t mrR = shl1Add(static_cast<uint32_t>(p1), R.k10000) - tR;
        const uint32_t qK = shl1Add(static_cast<uint32_t>(q2), R.k8000);
        const uint32_t mrI = shl1Add(static_cast<uint32_t>(q1), qK);
        quirkWatch(R.watch, tR, qK);
        return __builtin_amdgcn_perm(mrI, mrR, 0x07060302u);
    }
    const uint32_t mrR = (static_cast<uint32_t>(p1 - p2) << 1) + R.k8000;
    const uint32_t qK = (static_cast<uint32_t>(q2) << 1) + R.k8000;
    const uint32_t mrI = (static_cast<uint32_t>(q1) << 1) + qK;
    quirkWatch(R.watch, (static_cast<uint32_t>(p2) << 1) + R.k8000, qK);
    return __builtin_amdgcn_perm(mrI, mrR, 0x07060302u);
}

// The same rotate with the two-product sums of v_dot2_i32_i16: 7 instructions.  The twiddle (c, s) is needed in three
// arrangements, derived once per twiddle of a stage:  (c, -s) and (s, c) for the real and imaginary sums, (2s, 2s) for
// the watch.  MR = 2 (sum + 0x4000); the sums cannot overflow (|c| + |s| <= 1.42, so |sum| < 2^31).
//   watch: the low words of 2 a.re s and 2 a.im s from ONE packed 16-bit multiply; a low word of exactly 0x8000 is the
//   smallest signed 16-bit number, so a packed signed minimum keeps it (the watch register starts at 0x7FFF7FFF).
// One twiddle of these stages has s = -1.0, whose negative does not fit 16 bits: entry 1 of the table, (0, -1.0), in one
// lane per stage.  There -s wraps to -1.0, the real sum comes out 2^16 a.im too small, and exactly that is added back
// through the sum's constant operand (`fix`: the register's high word in that lane, 0 in the others).
struct TwDot { uint32_t re, im, s2; };
__device__ __forceinline__ TwDot deriveTwDot(uint32_t tw)
{
    TwDot d;
    d.im = __builtin_amdgcn_alignbit(tw, tw, 16);                                                                  // (s, c)
    d.re = __builtin_bit_cast(uint32_t, static_cast<u16x2>(__builtin_bit_cast(u16x2, tw) * __builtin_bit_cast(u16x2, 0xFFFF0001u)));   // (c, -s)
    asm("v_pk_add_u16 %0, %1, %1 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(d.s2) : "v"(tw));                           // (2s, 2s)
    return d;
}
template <bool FIX>
__device__ __forceinline__ uint32_t rotateDotB(uint32_t A, const TwDot &t, BflyRegs &R, uint32_t fixMask)
{
    uint32_t xR, xI, lows;
    const uint32_t cR = FIX ? ((A & fixMask) | R.k4000) : R.k4000;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(xR) : "v"(A), "v"(t.re), "v"(cR));
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(xI) : "v"(A), "v"(t.im), "v"(R.k4000));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(lows) : "v"(A), "v"(t.s2));
    asm("v_pk_min_i16 %0, %0, %1" : "+v"(R.watchB) : "v"(lows));
    return __builtin_amdgcn_perm(xI << 1, xR << 1, 0x07060302u);
}
__device__ __forceinline__ bool quirkSeenB(uint32_t w) { return __any((w & 0xFFFFu) == 0x8000u || (w >> 16) == 0x8000u); }

// radix-2 butterfly outputs u - t, u + t (saturating for the 1994+ transform, wrapping for 1993)
template <bool SAT>
__device__ __forceinline__ void addSub(uint32_t u, uint32_t T, uint32_t &lo, uint32_t &hi)
{
    if (SAT) { lo = pkSubSat(u, T); hi = pkAddSat(u, T); }
    else     { lo = pkSub(u, T);    hi = pkAdd(u, T); }
}

// The first two entries of the twiddle table are exact: entry 0 = (cos, sin) = (-1.0, 0), entry 1 = (0, -1.0)
// (0x8000 is -1.0 in 1.15; dcs_tables.cpp asserts the values).  Multiplying by them needs no multiplier and
// never triggers the rounding quirk (the last product is 0 or a multiple of 0x8000), so the reference's result
// reduces to  t = (-a.re, -a.im)  resp.  t = (a.im, -a.re)  with 16-bit wrap-around (-(-32768) = -32768).
__device__ __forceinline__ uint32_t rotateByMinusOne(uint32_t A) { return pkSub(0u, A); }
__device__ __forceinline__ uint32_t rotateByMinusI(uint32_t A)
{
    return __builtin_bit_cast(uint32_t, static_cast<u16x2>(__builtin_bit_cast(u16x2, __builtin_amdgcn_alignbit(A, A, 16))
                                                            * __builtin_bit_cast(u16x2, 0xFFFF0001u)));
}

// One stage on the 16 registers of a lane, partner distance D registers.
// Layout A: the twiddle index of the pair (r, r + D) is r >> SH, known at compile time.
template <bool SAT, int D, int SH>
__device__ __forceinline__ void stageA(uint32_t (&x)[16], const TwA &W, BflyRegs &R)
{
    uint32_t y[16];
#pragma unroll
    for (int r = 0 ; r < 16 ; ++r)
        if (!(r & D))
        {
            const int idx = r >> SH;
            const uint32_t T = idx == 0 ? rotateByMinusOne(x[r + D]) : idx == 1 ? rotateByMinusI(x[r + D])
                                                                                 : rotateFastA(x[r + D], W.t[idx], R);
            addSub<SAT>(x[r], T, y[r], y[r + D]);
        }
    if (__builtin_expect(quirkSeen(R.watch), 0))
    {
        R.watch = 0xFFFFu;              // (start watching again: only this stage is repeated)
#pragma unroll
        for (int r = 0 ; r < 16 ; ++r)
            if (!(r & D) && (r >> SH) >= 2)
            {
                int c2 = W.t[r >> SH].c2, s2 = W.t[r >> SH].s2;
                asm volatile("" : "+v"(c2), "+v"(s2));          // (halved HERE, in the rare path, not ahead of the pass loop)
                addSub<SAT>(x[r], rotateExact(x[r + D], c2 >> 1, s2 >> 1), y[r], y[r + D]);
            }
    }
#pragma unroll
    for (int r = 0 ; r < 16 ; ++r)
        x[r] = y[r];
}
// Layout B: per-lane twiddles tw[(r >> SH)] from the lane constants.  DOT selects the rotate: the two-product-sum form
// (1993 transform) or the four-product form (1994+, where deriving three registers per twiddle for fewer butterflies
// per twiddle does not pay; measured).
template <bool SAT, int D, int SH, bool DOT = !SAT>
__device__ __forceinline__ void stageB(uint32_t (&x)[16], const uint32_t *tw, BflyRegs &R)
{
    uint32_t y[16];
    bool redo;
    if constexpr (DOT)
    {
        constexpr int kTw = 8 / D;          // distinct twiddles of the stage
        TwDot td[kTw];
#pragma unroll
        for (int k = 0 ; k < kTw ; ++k)
            td[k] = deriveTwDot(tw[k]);
        // the twiddle that can be (0, -1.0): part 1 = kTw * lane + k, i.e. k = 1 of lane 0 (the stage's only one: k = 0 of lane 1)
        constexpr int kFix = kTw == 1 ? 0 : 1;
        const uint32_t fixMask = tw[kFix] == 0x80000000u ? 0xFFFF0000u : 0u;
#pragma unroll
        for (int r = 0 ; r < 16 ; ++r)
            if (!(r & D))
            {
                const uint32_t T = (r >> SH) == kFix ? rotateDotB<true>(x[r + D], td[r >> SH], R, fixMask)
                                                     : rotateDotB<false>(x[r + D], td[r >> SH], R, 0u);
                addSub<SAT>(x[r], T, y[r], y[r + D]);
            }
        redo = quirkSeenB(R.watchB);
        if (redo)
            R.watchB = 0x7FFF7FFFu;         // (start watching again: only this stage is repeated)
    }
    else
    {
#pragma unroll
        for (int r = 0 ; r < 16 ; ++r)
            if (!(r & D))
                addSub<SAT>(x[r], rotateFastB<SAT>(x[r + D], tw[r >> SH], R), y[r], y[r + D]);
        redo = quirkSeen(R.watch);
        if (redo)
            R.watch = 0xFFFFu;              // (start watching again: only this stage is repeated)
    }
    if (__builtin_expect(redo, 0))
    {
#pragma unroll
        for (int r = 0 ; r < 16 ; ++r)
            if (!(r & D))
                addSub<SAT>(x[r], rotateExact(x[r + D], sx16(tw[r >> SH]), static_cast<int>(tw[r >> SH]) >> 16), y[r], y[r + D]);
    }
#pragma unroll
    for (int r = 0 ; r < 16 ; ++r)
        x[r] = y[r];
}

__device__ __forceinline__ int bitrev9(int v) { return static_cast<int>(__brev(static_cast<uint32_t>(v)) >> 23); }
__device__ __forceinline__ int bitrevN(int v, int bits) { return static_cast<int>(__brev(static_cast<uint32_t>(v)) >> (32 - bits)); }

// overlap-add of one sample (.cpp:545-554, :797-801): both products signed x unsigned, sum, +0x8000, high word
__device__ __forceinline__ int overlapMix(int x, uint32_t cx, int o, uint32_t co)
{
    const uint32_t a = static_cast<uint32_t>(x * static_cast<int>(cx)) << 1;
    const uint32_t b = static_cast<uint32_t>(o * static_cast<int>(co)) << 1;
    return static_cast<int>(a + b + 0x8000u) >> 16;
}

typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));      // 16 bytes at any dword address

// the lane's constants for one of the two transforms (DcsDevTables.lane94 / lane93): six 16-byte loads
struct LaneConsts { uint32_t k[DCS_LANE_CONSTS]; };
__device__ __forceinline__ void loadLaneConsts(const DcsDevTables *G, int lane, int xform, LaneConsts &C)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(xform == DCS_XFORM_94 ? G->lane94[lane] : G->lane93[lane]);
#pragma unroll
    for (int i = 0 ; i < DCS_LANE_CONSTS / 4 ; ++i)
    {
        const uint4 v = src[i];
        C.k[4 * i] = v.x; C.k[4 * i + 1] = v.y; C.k[4 * i + 2] = v.z; C.k[4 * i + 3] = v.w;
    }
}

// Ordering of LDS traffic between the lanes of ONE wavefront.  A wavefront's LDS instructions execute
// in program order, so no hardware wait is needed; this only stops the compiler from moving LDS
// accesses across the point.  (A workgroup barrier must not be used here: the four wavefronts of a
// workgroup work on independent chunks with different trip counts.)
__device__ __forceinline__ void waveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// per-lane description of the frame a lane works on in a transform pass
struct PassLane
{
    uint32_t *rowC;         // the frame's spectrum row in the tile; reused as transpose scratch once it is in registers
    int l;                  // lane inside the group
    uint32_t shiftPair;     // volShift | volShift << 16
    bool paced;             // a wavefront of the launch's last generation (see the kernel): it lowers its priority as the transform gets on
#ifdef DCS_STAMPS_XFORM
    Stamper stamp;
#endif
};
#ifdef DCS_STAMPS_XFORM
#define DCS_XSTAMP(k) P.stamp(k)
#else
#define DCS_XSTAMP(k)
#endif

__device__ __forceinline__ uint4 ldsRead4(const uint32_t *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ void ldsWrite4(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{ *reinterpret_cast<uint4 *>(p) = make_uint4(a, b, c, d); }

// The transposes go through the frame's own tile row viewed as 8 rows x 16 dwords.  The four quads of a
// row are XOR-swizzled with the row number so that 128-bit accesses of the 8 lanes of a frame spread
// over all banks: dword index of (row, pos) = row*16 + (((pos >> 2) ^ (row >> 1)) & 3)*4 + (pos & 3).
__device__ __forceinline__ int swzQuad(int row, int quad) { return row * 16 + ((quad ^ (row >> 1)) & 3) * 4; }
__device__ __forceinline__ int swzPos(int row, int pos) { return swzQuad(row, pos >> 2) + (pos & 3); }

// 1994+ transform of 8 frames, 8 lanes each (DecoderImpl94x::TransformFrame, .cpp:397-534).
// On return x[r'] holds point 16*l + r' = output sample pair m = 8*bitrev4(r') + bitrev3(l), shifted.
__device__ __forceinline__ void transform94x8(const PassLane &P, const TwA &W, const LaneConsts &C, BflyRegs &R, uint32_t (&x)[16])
{
    const int l = P.l;
    uint32_t *S = P.rowC;
    // ---- pre-passes 1 + 2 on the pairs (i, 128 - i), i = l + 8j (:403-456) ------------------------
    uint32_t An[8], Bn[8];
    {
        uint32_t a_[8], b_[8];
#pragma unroll
        for (int j = 0 ; j < 8 ; ++j)
        {
            const int i = l + 8 * j;
            const uint32_t X = P.rowC[i];
            uint32_t Y = P.rowC[(128 - i) & 127];
            if (i == 0) Y = 0;                                  // words 0x100/0x101 start at zero
            // MulSS(v, 0x8000) = wrapping negate
            const uint32_t Sm = pkAddSat(X, Y), Df = pkSubSat(X, Y);
            a_[j] = pkSub(0u, __builtin_amdgcn_perm(Df, Sm, 0x07060100u));     // (-(x0+y0), -(x1-y1))
            b_[j] = pkSub(0u, __builtin_amdgcn_perm(Sm, Df, 0x07060100u));     // (-(x0-y0), -(x1+y1))
        }
        // prod0 = b1*c1 - b0*c0 (rounding keyed on b0*c0) ; prod1 = b1*c0 + b0*c1 (keyed on b0*c1); b = (b0, b1), pre94 = (c0, c1)
        auto finish = [&](int j, uint32_t Pr)                   // Pr = (prod1, prod0)
        {
            const uint32_t a = a_[j];
            An[j] = pkAddSat(Pr, a);                                        // (prod1 + a0, prod0 + a1)
            const uint32_t t = pkSubSat(a, Pr);                             // (a0 - prod1, a1 - prod0)
            const uint32_t t2 = pkSubSat(Pr, a);                            // (prod1 - a0, prod0 - a1)
            Bn[j] = __builtin_amdgcn_perm(t2, t, 0x07060100u);              // (a0 - prod1, prod0 - a1)
        };
#pragma unroll
        for (int j = 0 ; j < 8 ; ++j)
        {
            int p1, p2, q1, q2;
            DCS_MUL_SEL(p2, b_[j], "WORD_0", C.k[DCS_K94_PRE + j], "WORD_0");
            DCS_MUL_SEL(p1, b_[j], "WORD_1", C.k[DCS_K94_PRE + j], "WORD_1");
            DCS_MUL_SEL(q2, b_[j], "WORD_0", C.k[DCS_K94_PRE + j], "WORD_1");
            DCS_MUL_SEL(q1, b_[j], "WORD_1", C.k[DCS_K94_PRE + j], "WORD_0");
            const uint32_t tR = shl1Add(static_cast<uint32_t>(p2), R.k8000);
            const uint32_t m0 = shl1Add(static_cast<uint32_t>(p1), R.k10000) - tR;
            const uint32_t qK = shl1Add(static_cast<uint32_t>(q2), R.k8000);
            const uint32_t m1 = shl1Add(static_cast<uint32_t>(q1), qK);
            quirkWatch(R.watch, tR, qK);
            finish(j, __builtin_amdgcn_perm(m0, m1, 0x07060302u));
        }
        if (__builtin_expect(quirkSeen(R.watch), 0))
        {
            R.watch = 0xFFFFu;
#pragma unroll
            for (int j = 0 ; j < 8 ; ++j)
            {
                const int b0 = reC(b_[j]), b1 = imC(b_[j]);
                const int c0 = sx16(C.k[DCS_K94_PRE + j]), c1 = static_cast<int>(C.k[DCS_K94_PRE + j]) >> 16;
                const int p2 = __mul24(b0, c0), q2 = __mul24(b0, c1);
                uint32_t m0 = (static_cast<uint32_t>(__mul24(b1, c1) - p2) << 1) + 0x8000u;
                uint32_t m1 = (static_cast<uint32_t>(__mul24(b1, c0) + q2) << 1) + 0x8000u;
                if ((p2 & 0x7FFF) == 0x4000) m0 &= ~0x10000u;
                if ((q2 & 0x7FFF) == 0x4000) m1 &= ~0x10000u;
                finish(j, __builtin_amdgcn_perm(m0, m1, 0x07060302u));
            }
        }
    }
    // point 64: real part negated, imaginary part unchanged (:403-404); the pair i = 0 of lane 0 has no
    // partner, so lane 0 places it where B[0] would go
    if (l == 0)
    {
        const uint32_t M = P.rowC[64];
        Bn[0] = __builtin_amdgcn_perm(M, pkSub(0u, M), 0x07060100u);
    }
    waveSync();         // every lane has read its part of the row: the row becomes scratch
    // transpose into layout A: point p sits at row (p & 7), position (p >> 3).  A points of lane l fill
    // positions 0..7 of row l; its B points (128 - i) fill positions 15 - j of row (8 - l) & 7, except
    // lane 0 whose B points 128 - 8j sit at position 16 - j of row 0 (j = 0 holds point 64 -> position 8)
    {
        ldsWrite4(S + swzQuad(l, 0), An[0], An[1], An[2], An[3]);
        ldsWrite4(S + swzQuad(l, 1), An[4], An[5], An[6], An[7]);
        const int rb = (8 - l) & 7;
        if (l != 0)
        {
            ldsWrite4(S + swzQuad(rb, 2), Bn[7], Bn[6], Bn[5], Bn[4]);
            ldsWrite4(S + swzQuad(rb, 3), Bn[3], Bn[2], Bn[1], Bn[0]);
        }
        else
        {
            ldsWrite4(S + swzQuad(rb, 2), Bn[0], Bn[7], Bn[6], Bn[5]);
            ldsWrite4(S + swzQuad(rb, 3), Bn[4], Bn[3], Bn[2], Bn[1]);
        }
    }
    waveSync();
#pragma unroll
    for (int c = 0 ; c < 4 ; ++c)
    {
        const uint4 v = ldsRead4(S + swzQuad(l, c));
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
    waveSync();         // the row is reused by the second transpose
    if (P.paced) __builtin_amdgcn_s_setprio(2);
    // ---- layout A: point p = 8r + l.  pre-pass 3 (:458-471) then stages d = 32, 16, 8 (:480-524) -------
#pragma unroll
    for (int r = 0 ; r < 8 ; ++r)
    {
        const uint32_t u = x[r], a = x[r + 8];
        x[r] = pkAddSat(u, a); x[r + 8] = pkSubSat(u, a);
    }
    stageA<true, 4, 3>(x, W, R);
    stageA<true, 2, 2>(x, W, R);
    stageA<true, 1, 1>(x, W, R);
    if (P.paced) __builtin_amdgcn_s_setprio(1);
    // ---- transpose to layout B: point p = 16 l' + r' lives in row (p >> 4), position (p & 15) ---------------
#pragma unroll
    for (int r = 0 ; r < 16 ; ++r)
        S[swzPos(r >> 1, 8 * (r & 1) + l)] = x[r];
    waveSync();
#pragma unroll
    for (int c = 0 ; c < 4 ; ++c)
    {
        const uint4 v = ldsRead4(S + swzQuad(l, c));
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
    // ---- stages d = 4, 2, 1 ------------------------------------------------------------------------------
    stageB<true, 4, 3>(x, C.k + DCS_K94_TWB, R);
    if (P.paced) __builtin_amdgcn_s_setprio(0);
    stageB<true, 2, 2>(x, C.k + DCS_K94_TWB + 2, R);
    stageB<true, 1, 1>(x, C.k + DCS_K94_TWB + 6, R);
    // volume shift (:532-534); at full volume there is none
    if (__any(P.shiftPair != 0))
    {
#pragma unroll
        for (int r = 0 ; r < 16 ; ++r)
            x[r] = pkAshr(x[r], P.shiftPair);
    }
}

// 1993 transform of 4 frames, 16 lanes each (DecoderImpl93::TransformFrame, .cpp:714-785; the DC
// magnitude step ran in phase 1).  On return x[r'] holds point 16*l + r'; its real part is output
// sample i = 16*bitrev4(r') + bitrev4(l), shifted.
__device__ __forceinline__ void transform93x4(const PassLane &P, const TwA &W, const LaneConsts &C, BflyRegs &R, uint32_t (&x)[16])
{
    const int l = P.l;
    uint32_t *S = P.rowC;
    // ---- expand 128 -> 256 points (:714-732): S[p] = (A.re + B.re, A.im - B.im), S[128 + p] = (A.re - B.re, A.im + B.im)
    // with A = row[p], B = row[128 - p]; layout A: point p = 16 r + l, so S[p] -> x[r], S[128 + p] -> x[r + 8]
#pragma unroll
    for (int r = 0 ; r < 8 ; ++r)
    {
        const int p = 16 * r + l;
        const uint32_t A = P.rowC[p];
        const uint32_t B = P.rowC[(128 - p) & 127];
        const uint32_t Sm = pkAdd(A, B), Df = pkSub(A, B);
        x[r] = __builtin_amdgcn_perm(Df, Sm, 0x07060100u);          // (sum.re, diff.im)
        x[r + 8] = __builtin_amdgcn_perm(Sm, Df, 0x07060100u);      // (diff.re, sum.im)
        if (p == 0)
        {
            x[0] = A; x[8] = A;                                     // (|DC|, 0) from dcMagnitude93 (:709-710)
        }
    }
    DCS_XSTAMP(8);
    if (P.paced) __builtin_amdgcn_s_setprio(2);
    // ---- stages d = 64, 32, 16 (wrapping) (:742-778) ------------------------------------------------------
    stageA<false, 4, 3>(x, W, R);
    stageA<false, 2, 2>(x, W, R);
    stageA<false, 1, 1>(x, W, R);
    // ---- transpose: point 16 r + l  ->  lane r, register l.  The row holds 8 x 16 dwords, so two rounds:
    // registers 0..7 feed lanes 0..7, registers 8..15 feed lanes 8..15 ---------------------------------------------
    DCS_XSTAMP(9);
    waveSync();         // every lane has read its part of the row: the row becomes scratch
    uint32_t y[16];
#pragma unroll
    for (int h = 0 ; h < 2 ; ++h)
    {
#pragma unroll
        for (int r = 0 ; r < 8 ; ++r)
            S[swzPos(r, l)] = x[8 * h + r];
        waveSync();
        if ((l >> 3) == h)
        {
#pragma unroll
            for (int c = 0 ; c < 4 ; ++c)
            {
                const uint4 v = ldsRead4(S + swzQuad(l & 7, c));
                y[4 * c] = v.x; y[4 * c + 1] = v.y; y[4 * c + 2] = v.z; y[4 * c + 3] = v.w;
            }
        }
        waveSync();
    }
#pragma unroll
    for (int r = 0 ; r < 16 ; ++r)
        x[r] = y[r];
    DCS_XSTAMP(10);
    if (P.paced) __builtin_amdgcn_s_setprio(1);
    // ---- stages d = 8, 4, 2, 1 -----------------------------------------------------------------------------
    stageB<false, 8, 4>(x, C.k + DCS_K93_TWB, R);
    stageB<false, 4, 3>(x, C.k + DCS_K93_TWB + 1, R);
    if (P.paced) __builtin_amdgcn_s_setprio(0);
    stageB<false, 2, 2>(x, C.k + DCS_K93_TWB + 3, R);
    stageB<false, 1, 1>(x, C.k + DCS_K93_TWB + 7, R);
    DCS_XSTAMP(11);
    // volume shift of the real parts (:782-785); at full volume there is none
    if (__any(P.shiftPair != 0))
    {
#pragma unroll
        for (int r = 0 ; r < 16 ; ++r)
            x[r] = pkAshr(x[r], P.shiftPair);
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <int FPW>
#ifndef DCS_MIN_WAVES
#define DCS_MIN_WAVES 4
#endif
// The arguments are those of DcsKernelArgs, as separate kernel parameters: the first 16 dwords of the kernel-argument
// segment -- everything the prologue and the single-source path need -- are then preloaded into scalar registers
// when the wavefront is launched (-amdgpu-kernarg-preload-count, Makefile), instead of being fetched by the kernel's
// first instructions with the package loads waiting behind that fetch.
__global__ void __launch_bounds__(64 * kWavesPerBlock, DCS_MIN_WAVES)
dcsDecodeKernel(uint8_t *kPackages, const DcsDevTables *kTables, uint32_t kNChunks, uint32_t kFlags, uint32_t kEpoch, uint32_t kNJobs,
                int16_t *kPcm, unsigned long long *kHandoff, uint32_t *kErr, int16_t *kTailsOut,
                const uint8_t *kBlob, uint64_t kBlobLen, const DcsSrcDesc *kSrcs, const int16_t *kTailsIn, unsigned long long *kDebug)
{
    DcsKernelArgs a;
    a.blob = kBlob; a.blobLen = kBlobLen; a.srcs = kSrcs; a.packages = kPackages; a.nChunks = kNChunks; a.nJobs = kNJobs;
    a.pcm = kPcm; a.err = kErr; a.tailsIn = kTailsIn; a.tailsOut = kTailsOut; a.tables = kTables; a.debug = kDebug;
    a.handoff = kHandoff; a.epoch = kEpoch; a.flags = kFlags;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    constexpr int kPoolBytes = poolDwords(FPW) * 4, kTileBytes = FPW * kRowBytes + (FPW + 1) * 32;
    static_assert(kPoolBytes % 16 == 0 && kWavesPerBlock * kTileBytes >= 8192, "LDS layout");
    const Lds<FPW> L{ smem, smem + DCS_LDS_DECODE_BYTES + kWavesPerBlock * kPoolBytes + wave * kTileBytes,
                      smem + DCS_LDS_DECODE_BYTES + wave * kPoolBytes };
    // (DCS_BATCH_XCD_RANGES, dcs_common.h: workgroup i runs on XCD i % 8; XCD j takes the logical workgroups [j R, (j + 1) R) in order)
    uint32_t blk = blockIdx.x;
    if (a.flags & DCS_BATCH_XCD_RANGES)
        blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const uint32_t chunk = blk * kWavesPerBlock + static_cast<uint32_t>(wave);
#ifdef DCS_STAMPS
    const Stamper stamp{ (lane == 0 && a.debug != nullptr && chunk < a.nChunks) ? a.debug + static_cast<size_t>(chunk) * 16 : nullptr };
#else
    const Stamper stamp{};
#endif
    DCS_STAMP(0);
    // Priorities (DCS_BATCH_PACED, set by launches with more than one wavefront per SIMD).  A SIMD serves its OLDEST wavefront first, so
    // the four of a SIMD finish one after the other, which is fine while fresh wavefronts keep coming (a staggered finish hides the
    // newcomers' wait for their packages) and costs at the END of the launch: the last ones finish alone, each on a SIMD it cannot
    // fill.  So the LAST generation -- the workgroups that find no successors, gridDim - CUs x 4 onwards -- runs least progress
    // first where it counts, at the end: every wavefront starts at priority 3 and those lower it as the transform gets on (2 behind
    // the pre-passes, 1 at the second transpose, 0 for the last stages and the stores); among equals the older one still goes first,
    // so earlier generations, which stay at 3, are never held up by newcomers.  A launch of one generation is all "last".
    // Measured (NOTES 43): survey3_65536 32.8 -> 31.1 us, dcs94_65536 32.6 -> 30.8, realistic_65536 36.7 -> 35.4, 16 384 mixed
    // frames 15.1 -> 14.3, sixteen generations 444 -> 442; steps earlier in the wavefront's life (the unpack rounds) or later (the
    // stores) measured worse.
    const bool pacedLaunch = (a.flags & DCS_BATCH_PACED) != 0;
    const bool paced = pacedLaunch && blockIdx.x + ((a.flags >> DCS_BATCH_CUS8_SHIFT) & 0xFFu) * 32u >= gridDim.x;
    if (pacedLaunch) __builtin_amdgcn_s_setprio(3);
    constexpr int SUB = subLanes(FPW);              // lanes that unpack one frame together
#ifndef DCS_DIRECT_MAX_FPW
#define DCS_DIRECT_MAX_FPW 4
#endif
    // the bit readers (measured per variant, tools/ab.sh): the 1993 layouts always read position-only; the 1994+ symbol
    // loop, whose next position depends on the symbol just read, does so only with 4 frames per wavefront and keeps the
    // window in registers with 8 and 16 (see DirectReader)
    using BR94 = typename std::conditional<(FPW <= DCS_DIRECT_MAX_FPW), DirectReader, BitReader>::type;
    using BR93 = DirectReader;
    static_assert(SUB * FPW == 64 && SUB <= 16, "every lane unpacks; a frame has at most 16 split lanes");
    const int s = lane % FPW;                       // slot of this lane
    const int q = lane / FPW;                       // which part of the frame it unpacks
    const bool unpacker = q < SUB;

    // ---- everything unpack round 0 needs comes from the chunk's package (dcs_common.h): slot, descriptor head, stream
    // header, this lane's split record, the pool image.  All of it is requested here, before anything else, in one
    // go; the padding wavefronts of the last workgroup read package 0 and drop out after the barrier.
    // (the packages' pool image is imgDw dwords long, as much as the batch's fullest chunk needs: flags bits 16..25, dcs_common.h)
    const uint32_t layout = (a.flags >> DCS_BATCH_IMG_SHIFT) & DCS_BATCH_IMG_MASK;
    const int imgDw = static_cast<int>(dcsPkgImgDw(layout));
    const bool split4 = (layout & DCS_PKG_SPLIT4) != 0;     // every source a 1994+ frame: 4-byte split records
    const uint32_t offPool = split4 ? dcsPkgOffPool(FPW, DCS_PKG_SPLIT4) : dcsPkgOffPool(FPW, 0);
    const uint8_t *pkg = a.packages + (chunk < a.nChunks ? static_cast<size_t>(chunk) * (offPool + static_cast<uint32_t>(imgDw) * 4u) : 0);
    // The head of the package (per slot 80 bytes: slot, descriptor head, header) is per-slot data: it is
    // fetched ONCE per wavefront, 16 bytes per lane, and handed to the lanes through LDS below (fewer vector-memory
    // instructions in the burst at the start of a kernel, where every wavefront of the chip issues its loads at once).
    constexpr int kHeadVec = FPW * 5, kHeadLoads = (kHeadVec + 63) / 64;
    static_assert(kHeadLoads <= 2, "package head: two 16-byte loads per lane at most");
    uint4 phead0, phead1 = make_uint4(0, 0, 0, 0);      // (named registers: an array of two ended up in scratch memory)
    uint2 psplit;
    constexpr int kPoolPieces = (poolDwords(FPW) + 255) / 256;
    uint4 pimg[kPoolPieces];
    {
        static_assert(sizeof(DcsSlot) == 32 && offsetof(DcsSlot, runStartDw) == 16 && offsetof(DcsSlot, nextJob) == 24 && DCS_PKG_SLOT_BYTES == 80, "DcsSlot / package layout");
        phead0 = reinterpret_cast<const uint4 *>(pkg)[min(lane, kHeadVec - 1)];
        if (kHeadLoads > 1)
            phead1 = reinterpret_cast<const uint4 *>(pkg)[min(lane + 64, kHeadVec - 1)];
        if (split4)
        {
            // (4-byte records, lane l's at l x 4: one dword -- ADVICE r5: an 8-byte load there was 4-byte aligned on odd lanes and
            // read past the split area on the last one; a uniform branch, the layout word comes with the kernel's arguments)
            const uint32_t raw = *reinterpret_cast<const uint32_t *>(pkg + dcsPkgOffSplit(FPW) + static_cast<uint32_t>(lane) * 4u);
            psplit = make_uint2(raw & 0xFFFFu, raw & 0xFFFF0000u);
        }
        else
            psplit = *reinterpret_cast<const uint2 *>(pkg + dcsPkgOffSplit(FPW) + static_cast<uint32_t>(lane) * 8u);
#pragma unroll
        for (int t = 0 ; t < kPoolPieces ; ++t)
        {
            // (unconditional: a predicated load would make the compiler wait for every load above before the tables
            // are even requested; lanes past the image re-read its last 16 bytes and store nothing)
            const int i = min(lane * 4 + 256 * t, imgDw - 4);
            pimg[t] = *reinterpret_cast<const uint4 *>(pkg + offPool + i * 4);
        }
    }
    TwA W;
    loadTwA(a.tables, W);

    // ---- stage the shared tables (whole workgroup), clear this wavefront's tile ------------------------
    {
        // one 16-byte piece of the tables per thread, requested (unconditionally, see above) before the tile is cleared:
        // the clearing needs nothing from memory and runs while all these loads are in flight
        constexpr int kTableVec = DCS_LDS_DECODE_BYTES / 16;
        static_assert(kTableVec <= 64 * kWavesPerBlock, "one piece of the tables per thread");
        const uint4 piece = reinterpret_cast<const uint4 *>(&a.tables->lds)[min(static_cast<int>(threadIdx.x), kTableVec - 1)];
        // batches with OS93a Type-1 frames: the 4 KB sample-pair table, one 16-byte piece per thread (a uniform branch)
        const bool stagePairs = pairTableInLds(FPW) && (a.flags & DCS_BATCH_HAS_93A_T1);
        uint4 pairPiece = make_uint4(0, 0, 0, 0);
        if (stagePairs)
            pairPiece = reinterpret_cast<const uint4 *>(a.tables->pair93a)[threadIdx.x];
        uint4 *tile = reinterpret_cast<uint4 *>(L.base);
#ifndef DCS_CLEAR_FIRST_MAX_FPW
#define DCS_CLEAR_FIRST_MAX_FPW 8
#endif
        // (with 16 frames per wavefront, where the clearing is four times as long and other workgroups of the CU are busy
        // anyway, tables first measured 1 % better)
        constexpr bool kClearFirst = FPW <= DCS_CLEAR_FIRST_MAX_FPW;
        if (kClearFirst)
            for (int i = lane ; i < FPW * kRowBytes / 16 ; i += 64)
                tile[i] = make_uint4(0, 0, 0, 0);
        if (static_cast<int>(threadIdx.x) < kTableVec)
            reinterpret_cast<uint4 *>(smem)[threadIdx.x] = piece;
        if (stagePairs)
            reinterpret_cast<uint4 *>(smem + ldsBytes(FPW) - 4096)[threadIdx.x] = pairPiece;
        if (!kClearFirst)
            for (int i = lane ; i < FPW * kRowBytes / 16 ; i += 64)
                tile[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();                                // the only workgroup barrier: tables are in place
    if (chunk >= a.nChunks)
        return;                                     // padding wavefront of the last workgroup

    DCS_STAMP(1);

    // hand the package head to the lanes: through this wavefront's bit pool (filled with the image right after)
    struct { uint32_t job; uint32_t prevSlot, flags, nSrc, shiftXform; uint32_t firstSrc, prevJob, poolOff, bpl; } slot;
    uint32_t exportNext;                                // where the chunk's tail for another chunk is due (a scalar: see below)
    uint4 pd0, pd1, phdr;
    uint2 pd2;
    {
        uint4 *scratch = reinterpret_cast<uint4 *>(L.pool());
        static_assert(kHeadVec * 16 <= poolDwords(FPW) * 4, "the package head fits in the bit pool");
        if (lane < kHeadVec)
            scratch[lane] = phead0;
        if (kHeadLoads > 1 && lane + 64 < kHeadVec)
            scratch[lane + 64] = phead1;
        waveSync();
        const uint4 *sp5 = scratch + 5 * s;             // (dcs_common.h: five pieces per slot)
        const uint4 s0 = sp5[0];
        pd0 = sp5[1]; pd1 = sp5[2];
        const uint4 d2v = sp5[3];
        pd2 = make_uint2(d2v.x, d2v.y);
        phdr = sp5[4];
        waveSync();
        slot.job = s0.x;
        slot.prevSlot = s0.y & 0xFFu; slot.flags = (s0.y >> 8) & 0xFFu; slot.nSrc = (s0.y >> 16) & 0xFFu; slot.shiftXform = s0.y >> 24;
        slot.firstSrc = s0.z; slot.prevJob = s0.w;
        slot.poolOff = d2v.z & 0xFFFFu;
        slot.bpl = (d2v.z >> 16) & 0xFFu;
        // (DCS_SLOT_EXPORT: the job whose first samples this frame's tail overlaps into.  A chunk has at most one such frame, its
        // last one, so this is the same for every lane: taken into a scalar register here, no vector register lives through phase 1)
        const unsigned long long exportSlots = __ballot(lane < FPW && !(slot.flags & DCS_SLOT_EMPTY) && (slot.flags & DCS_SLOT_EXPORT) != 0);
        exportNext = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(d2v.w), exportSlots != 0 ? static_cast<int>(__builtin_ctzll(exportSlots)) : 0));
    }

    // the lane's transform constants for the chunk's first frame (the whole chunk, normally): requested now, needed in
    // phase 2.  (Requesting them with the first loads of the kernel, which a batch flag could allow when every job runs
    // the same transform, makes that first round trip longer: 4 % slower on the 4 096-frame batch.)
    int constsXform = __builtin_amdgcn_readfirstlane(static_cast<int>(slot.shiftXform >> 4)) == DCS_XFORM_94 ? DCS_XFORM_94 : DCS_XFORM_93;
    LaneConsts C;
    loadLaneConsts(a.tables, lane, constsXform, C);
    // the bit pool of round 0: a straight copy of the package's image
#pragma unroll
    for (int t = 0 ; t < kPoolPieces ; ++t)
    {
        const int i = lane * 4 + 256 * t;
        const bool in = i < imgDw;                  // (behind the image the pool is zero, as it was when the image had the pool's length)
        if (i < poolDwords(FPW))
            ldsWrite4(L.pool() + i, in ? pimg[t].x : 0u, in ? pimg[t].y : 0u, in ? pimg[t].z : 0u, in ? pimg[t].w : 0u);
    }

    // ---- job of this lane's slot ---------------------------------------------------------------------------
    const bool live = !(slot.flags & DCS_SLOT_EMPTY);
    struct { uint32_t firstSrc; int nSrc; int volShift; int xform; uint32_t prev; } job;
    job.firstSrc = slot.firstSrc; job.nSrc = slot.nSrc; job.volShift = slot.shiftXform & 15;
    job.xform = slot.shiftXform >> 4; job.prev = slot.prevJob;

    // ---- phase 1: unpack, one round per source index -------------------------------------------------
    uint32_t err = 0;
    {
        uint32_t *pool = L.pool();
        const DcsLdsTables *T = L.tables();
        uint16_t *row = L.row(s);
        const uint32_t *blobW = reinterpret_cast<const uint32_t *>(a.blob);
        const uint32_t blobWords = static_cast<uint32_t>((a.blobLen + 3) >> 2);
        const int myNSrc = live ? job.nSrc : 0;

        // one unpack round = the r-th source of every frame of the chunk.  R0 (the first source, in most batches the
        // only one): the planner has put into the slot record where the compressed bytes, the stream header and
        // the lane's split record lie, so all of it is requested in ONE memory round trip together with the
        // descriptor; for further sources (multi-channel mixes) the addresses follow from the descriptor.
        auto unpackRound = [&](auto r0tag, const int r)
        {
            constexpr bool R0 = decltype(r0tag)::value;
            const bool has = r < myNSrc;
            // the descriptor: identical addresses within a slot's sub-lanes
            const uint4 *sdp = reinterpret_cast<const uint4 *>(&a.srcs[has ? job.firstSrc + r : 0]);
            uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
            uint2 d2 = make_uint2(0, 0);
            uint2 sp = make_uint2(0, 0);
            const int bplSlot = slot.bpl;
            if (R0)
            {
                if (has) { d0 = pd0; d1 = pd1; d2 = pd2; sp = psplit; }
            }
            else if (has) { d0 = sdp[0]; d1 = sdp[1]; d2 = *reinterpret_cast<const uint2 *>(sdp + 2); }
            // DcsSrcDesc: [0] streamOff lo, [1] streamOff hi, [2] mixMul | format<<16 | hdrLen<<24,
            // idx at byte 12: [3] bitOff, [4] nBits | hdrBits<<16, [5..8] bandType, [9] preAdj | nBands<<16 | flags<<24,
            // [10..39] split[15], two dwords each
            const uint64_t streamOff = static_cast<uint64_t>(d0.x) | (static_cast<uint64_t>(d0.y) << 32);
            const uint32_t mixMul = d0.z & 0xFFFFu;
            const int format = static_cast<int>((d0.z >> 16) & 0xFFu);
            const int hdrLen = has ? static_cast<int>(d0.z >> 24) : 16;
            const uint32_t bitOff = d0.w;
            const uint32_t nBits = d1.x & 0xFFFFu, hdrBits = d1.x >> 16;
            const int nBands = static_cast<int>((d2.y >> 16) & 0xFFu);
            const uint32_t flags = d2.y >> 24;
            const bool serial = R0 ? bplSlot == 0 : ((flags & DCS_IDX_SERIAL) != 0 || SUB == 1);

            Quarter Q;
            Q.t0 = d1.y; Q.t1 = d1.z; Q.t2 = d1.w; Q.t3 = d2.x;
            Q.preAdj = d2.y & 0xFFFFu;

            const uint64_t bitPos = (streamOff + 2 + static_cast<uint64_t>(hdrLen)) * 8 + bitOff;
            // ---- stage the compressed bytes into the bit pool, byte-swapped so that bit 31 of a dword is the next
            // stream bit.  All loads are issued before the first store, so their latencies overlap.
            uint32_t off;                                                   // pool dword of this lane's frame
            bool fits;
            if (R0)
            {
                // (the pool was filled from the package's image)
                off = min(static_cast<uint32_t>(slot.poolOff), static_cast<uint32_t>(poolDwords(FPW) - 1));
                fits = true;
            }
            else
            {
                // further sources: one coalesced run of dwords per slot, positions from a prefix sum over the slots
                const uint32_t startDw = static_cast<uint32_t>(bitPos >> 5);
                const uint32_t nDw = (has && q == 0) ? dcsPoolDwords(streamOff, static_cast<uint32_t>(hdrLen), bitOff, nBits) : 0u;
                uint32_t incl = nDw;
#pragma unroll
                for (int d = 1 ; d < 64 ; d <<= 1)
                {
                    const uint32_t up = __shfl_up(incl, d);
                    if (lane >= d) incl += up;
                }
                const uint32_t offMine = incl - nDw;
                off = __shfl(offMine, s);                                   // from the slot's q = 0 lane
                const uint32_t nDwSlot = __shfl(nDw, s);
                fits = off + nDwSlot <= static_cast<uint32_t>(poolDwords(FPW));
                constexpr int kStageUnroll = FPW < 8 ? FPW : 8;
                for (int t0 = 0 ; t0 < FPW ; t0 += kStageUnroll)
                {
                    uint32_t v[kStageUnroll], dst[kStageUnroll];
                    bool any64 = false;
#pragma unroll
                    for (int u = 0 ; u < kStageUnroll ; ++u)
                    {
                        const uint32_t n = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(nDw), t0 + u));
                        const uint32_t st = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(startDw), t0 + u));
                        const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(offMine), t0 + u));
                        const bool room = o + n <= static_cast<uint32_t>(poolDwords(FPW));
                        const uint32_t w = st + static_cast<uint32_t>(lane);
                        const bool mine = room && static_cast<uint32_t>(lane) < n;
                        v[u] = (mine && w < blobWords) ? blobW[w] : 0u;
                        dst[u] = mine ? o + static_cast<uint32_t>(lane) : 0xFFFFFFFFu;
                        any64 = any64 || (room && n > 64);
                    }
#pragma unroll
                    for (int u = 0 ; u < kStageUnroll ; ++u)
                        if (dst[u] != 0xFFFFFFFFu)
                            pool[dst[u]] = __builtin_bswap32(v[u]);
                    if (any64)
                    {
                        // frames longer than 256 bytes: the rest, slot by slot
                        for (int u = 0 ; u < kStageUnroll ; ++u)
                        {
                            const uint32_t n = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(nDw), t0 + u));
                            const uint32_t st = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(startDw), t0 + u));
                            const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(offMine), t0 + u));
                            if (o + n > static_cast<uint32_t>(poolDwords(FPW)))
                                continue;
                            for (uint32_t i = static_cast<uint32_t>(lane) + 64 ; i < n ; i += 64)
                            {
                                const uint32_t w = st + i;
                                pool[o + i] = w < blobWords ? __builtin_bswap32(blobW[w]) : 0u;
                            }
                        }
                    }
                }
            }

            if (R0) DCS_STAMP(2);
            // the stream header: 16 bytes at streamOff + 2 (round 0: from the package, aligned and masked there)
            if (R0)
            {
                Q.h0 = has ? phdr.x : 0u; Q.h1 = has ? phdr.y : 0u; Q.h2 = has ? phdr.z : 0u; Q.h3 = has ? phdr.w : 0u;
            }
            else
            {
                const uint32_t hw = static_cast<uint32_t>((streamOff + 2) >> 2);
                const uint32_t sh = static_cast<uint32_t>((streamOff + 2) & 3);
                uint32_t hdrW[5];
#pragma unroll
                for (int i = 0 ; i < 5 ; ++i)
                    hdrW[i] = (has && hw + i < blobWords) ? blobW[hw + i] : 0u;
                Q.h0 = __builtin_amdgcn_alignbyte(hdrW[1], hdrW[0], sh);
                Q.h1 = __builtin_amdgcn_alignbyte(hdrW[2], hdrW[1], sh);
                Q.h2 = __builtin_amdgcn_alignbyte(hdrW[3], hdrW[2], sh);
                Q.h3 = __builtin_amdgcn_alignbyte(hdrW[4], hdrW[3], sh);
                if (hdrLen == 1)
                {
                    Q.h0 &= 0xFFu; Q.h1 = Q.h2 = Q.h3 = 0;
                }
            }
            waveSync();

            if (R0) DCS_STAMP(3);
            // ---- which part of the frame this lane unpacks, and from which decoder state ----------------
            const bool ok = has && fits && unpacker;
            if (has && !fits && q == 0)
                err |= DCS_FRAME_FATAL | DCS_FRAME_STOP;            // cannot happen with the library's planner
            if (has && q == 0)
                err |= flags >> 4;                                  // errors the index pass met before band 0 (:1771-1773)
            uint32_t relBits = hdrBits;                             // band 0 starts behind the 1994+ frame header
            Q.bandBase = 0;
            Q.outIdx = 1;
            Q.prv = 0; Q.prvDelta = 0;
            Q.subType = (format == DCS_FMT_93B_T1) ? 0 : 2;
            Q.reuse = false; Q.first = true;
            Q.midEnd = Q.midStart = Q.midStraddle = false;
            if (serial)
                Q.nb = (q == 0) ? nBands : 0;
            else
            {
                const int nb16 = min(nBands, 16);
                if (R0)
                {
                    // the packer dealt the bands out (dcsLaneFirstBand): this lane's first
                    // band comes with its split record, its last one is where the next lane of the frame starts
                    const bool f93a = format == DCS_FMT_93A_T1;
                    const int nbEnd = f93a ? min(nBands, 18) : nb16;
                    const int myBase = (sp.x & 0x8000u) ? nbEnd
                                     : static_cast<int>(sp.y >> 28) + ((f93a && (sp.y & (DCS_SPLIT_BASE16 << 16)) != 0) ? 16 : 0);
                    const int nextBase = __shfl(myBase, lane + FPW);
                    Q.bandBase = myBase;
                    Q.nb = max((q == SUB - 1 ? nbEnd : nextBase) - myBase, 0);
                    if (SUB == 16)
                    {
                        // the frame's last lane may hold the second half of band 15; the lane before it then stops there
                        const bool mid = format >= DCS_FMT_94_T0 && q == SUB - 1 && myBase == 15 && (sp.y & (DCS_SPLIT_MID15 << 16)) != 0;
                        const bool nextMid = __shfl(static_cast<int>(mid), lane + FPW) != 0;
                        Q.midStart = mid;
                        Q.midStraddle = mid && (sp.y & (DCS_MID15_STRADDLE << 16)) != 0;
                        Q.midEnd = q == SUB - 2 && nextMid;
                        if (Q.midEnd)
                            Q.nb = 16 - myBase;
                    }
                }
                else
                {
                    // further sources of a frame: the deal worked out here, bpl = ceil(nBands / SUB) (OS93a Type 1: bands in
                    // order, the tail handled below)
                    const int bpl = max((nb16 + SUB - 1) / SUB, 1);
                    if (format == DCS_FMT_93A_T1)
                    {
                        Q.bandBase = min(q * bpl, nb16);
                        Q.nb = min(max(nb16 - Q.bandBase, 0), bpl);
                    }
                    else
                    {
                        Q.bandBase = dcsLaneFirstBand(format, q, bpl, nb16);
                        Q.nb = (q == SUB - 1 ? nb16 : dcsLaneFirstBand(format, q + 1, bpl, nb16)) - Q.bandBase;
                    }
                }
                if (q != 0 && Q.nb != 0)
                {
                    if (!R0)
                        sp = reinterpret_cast<const uint2 *>(sdp)[5 + Q.bandBase - 1];
                    const uint32_t sp0 = sp.x, sp1 = sp.y;
                    relBits = sp0 & 0x7FFFu;
                    Q.prv = sp0 >> 16;
                    Q.prvDelta = sp1 & 0xFFFFu;
                    const uint32_t st = sp1 >> 16;
                    Q.outIdx = static_cast<int>(st & 0x1FFu);
                    Q.subType = static_cast<int>((st >> 9) & 3u);
                    Q.reuse = (st & 0x800u) != 0;
                    Q.first = false;
                }
            }
            const uint32_t inPool = static_cast<uint32_t>(bitPos & 31) + relBits;
            const uint32_t *brAt = pool + (ok ? off + (inPool >> 5) : 0u);
            const int brBit = static_cast<int>(inPool & 31);

            if (R0) DCS_STAMP(12);
            // every lane enters the unpackers (their symbol loops are wave-convergent); lanes without a
            // source of that family are masked off inside
            const bool is94 = ok && format >= DCS_FMT_94_T0;
            // OS93a Type 1: up to 18 bands, the lane that holds band 15 also takes 16 and 17; a lane whose split
            // record says the frame ended earlier has nothing to do
            const bool is93a = ok && format == DCS_FMT_93A_T1 && Q.nb != 0 && !(Q.bandBase != 0 && Q.reuse);
            const bool is93 = ok && format < DCS_FMT_93A_T1;
            if (__any(is94))
            {
                BR94 br;
                br.init(brAt, brBit);
                err |= unpack94<R0, BR94, SUB>(T, row, br, Q, format, mixMul, is94, stamp);
            }
            BR93 br;
            br.init(brAt, brBit);
            if (__any(is93))
                err |= unpack93<R0, BR93>(T, row, br, Q, format, mixMul, is93, stamp);
            if (is93a)
            {
                // Round 0: the packer dealt all eighteen bands out (dcsLaneFirstBand), a lane walks [bandBase, bandBase + nb).
                // Further sources of a frame (the deal made above, bands 0..15 in order): with 16 lanes per frame bands 16
                // and 17 go to the lanes of bands 0 and 1, the two shortest (their split records travel in the frame record's
                // bandType bytes, dcs_scan.h); with fewer lanes per frame the lane that holds band 15 takes them as well.
                constexpr bool kSpreadTail = SUB == 16 && !R0;
                const int end = Q.bandBase + Q.nb;
                const int end2 = (!R0 && SUB != 16 && end >= 16 && nBands > 16) ? nBands : end;
                const int prv0 = Q.bandBase == 0 ? 0x1A : sx16(Q.prv), out0 = Q.bandBase == 0 ? 0 : Q.outIdx;
                const int hb0 = static_cast<int>(Q.h0 & 0xFFu);
                if (pairTableInLds(FPW))
                    err |= unpack93a<R0, BR93>(T, row, br, hb0, mixMul, reinterpret_cast<const uint16_t *>(smem + ldsBytes(FPW) - 4096),
                                         Q.bandBase, end2, prv0, out0);
                else
                    err |= unpack93a<R0, BR93>(T, row, br, hb0, mixMul, a.tables->pair93a, Q.bandBase, end2, prv0, out0);
                if (kSpreadTail && pairTableInLds(FPW) && q < 2 && 16 + q < nBands)
                {
                    const uint32_t r0 = q == 0 ? Q.t0 : Q.t2, r1 = q == 0 ? Q.t1 : Q.t3;       // DcsSplit of band 16 + q
                    if (!((r1 >> 16) & 0x800u))                                                   // (the frame had not ended)
                    {
                        const uint32_t inPool2 = static_cast<uint32_t>(bitPos & 31) + (r0 & 0xFFFFu);
                        BR93 br2;
                        br2.init(pool + off + (inPool2 >> 5), static_cast<int>(inPool2 & 31));
                        err |= unpack93a<R0, BR93>(T, row, br2, hb0, mixMul, reinterpret_cast<const uint16_t *>(smem + ldsBytes(FPW) - 4096),
                                             16 + q, 17 + q, sx16(r0 >> 16), static_cast<int>((r1 >> 16) & 0x1FFu));
                    }
                }
            }
            waveSync();
        };

        {
            if (__any(myNSrc > 0))
                unpackRound(std::true_type{}, 0);
            for (int r = 1 ; __any(r < myNSrc) ; ++r)
                unpackRound(std::false_type{}, r);
        }

        DCS_STAMP(4);
        // a frame's error bits = OR over its sub-lanes
        if (__any(err != 0))            // (rare: skip the exchange when no lane of the wavefront has anything to report)
        {
#pragma unroll
            for (int m = FPW ; m < 64 ; m <<= 1)
                err |= __shfl_xor(err, m);
        }
        if (live && q == 0)
        {
            if (job.xform == DCS_XFORM_93)
                dcMagnitude93(row);
            if (!(slot.flags & DCS_SLOT_HALO) && a.err != nullptr)
                a.err[slot.job] = err;
        }
    }
    waveSync();

    DCS_STAMP(5);
    // ---- phase 2: transform passes (8 frames x 8 lanes, or 4 frames x 16 lanes), overlap, emit ------------
    uint32_t *tails = reinterpret_cast<uint32_t *>(L.tails());         // [slot][8] dwords = 16 samples
#ifdef DCS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DCS_STAMP(7);
#endif
    const int nSlots = __popcll(__ballot(live && lane < FPW));          // padding slots are trailing
    const int slotJob = static_cast<int>(slot.job);
    const int slotWord = static_cast<int>(slot.flags | (slot.prevSlot << 8) | (static_cast<uint32_t>(job.volShift) << 16));
    const int jobXform = job.xform, jobPrev = static_cast<int>(job.prev);


    const unsigned long long slots94 = __ballot(live && lane < FPW && jobXform == DCS_XFORM_94);
    const bool oneXform = slots94 == 0 || slots94 == __ballot(live && lane < FPW);
    for (int s0 = 0 ; s0 < nSlots ; )
    {
        // a pass takes the run of slots from s0 on that want the same transform (8 frames at most for 1994+, 4 for 1993)
        const int xf = __builtin_amdgcn_readlane(jobXform, s0);
        const int G = (xf == DCS_XFORM_94) ? 8 : 4;
#ifndef DCS_RUN_BALLOT_MIN_FPW
#define DCS_RUN_BALLOT_MIN_FPW 8
#endif
        int n = 1;
        if constexpr (FPW >= DCS_RUN_BALLOT_MIN_FPW)
        {
            // the run length as a count of trailing zeros over a ballot
            const unsigned long long sameXf = (xf == DCS_XFORM_94) ? slots94 : ~slots94;
            n = min(min(static_cast<int>(__builtin_ctzll(~(sameXf >> s0))), G), nSlots - s0);
        }
        else if (oneXform)
            n = min(G, nSlots - s0);            // the usual case: every frame of the chunk wants the same transform
        else
        {
            // (with 4 slots the loop is as short and measured faster)
            while (n < G && s0 + n < nSlots && __builtin_amdgcn_readlane(jobXform, s0 + n) == xf)
                ++n;
        }

        if (xf != constsXform)
        {
            constsXform = xf;                       // a chunk that mixes decoders of both families
            loadLaneConsts(a.tables, lane, constsXform, C);
        }
        const int lpfShift = (xf == DCS_XFORM_94) ? 3 : 4;
        const int g = lane >> lpfShift;
        const bool active = g < n;
        const int mySlot = s0 + (active ? g : 0);
        const int myWord = __shfl(slotWord, mySlot);
        const int myFlags = myWord & 0xFF, myPrevSlot = (myWord >> 8) & 0xFF, myShift = myWord >> 16;
        const uint32_t myJob = static_cast<uint32_t>(__shfl(slotJob, mySlot));
        const uint32_t myPrevJob = static_cast<uint32_t>(__shfl(jobPrev, mySlot));
        const int lr = (xf == DCS_XFORM_94) ? bitrevN(lane & 7, 3) : bitrevN(lane & 15, 4);   // the output sample (pair) this lane overlaps

        // a tail handed in by the caller (DCS_PREV_EXT, streaming use): requested before the transform, so that the
        // overlap below never waits for memory
        uint32_t extTail = 0;
        if (__any(active && (myFlags & DCS_SLOT_EXT_TAIL) != 0))
        {
            if (active && (myFlags & DCS_SLOT_EXT_TAIL) != 0 && a.tailsIn != nullptr)
            {
                const size_t k = static_cast<size_t>(myPrevJob & 0x7FFFFFFFu);
                extTail = (xf == DCS_XFORM_94) ? reinterpret_cast<const uint32_t *>(a.tailsIn)[k * 8 + lr]
                                               : static_cast<uint32_t>(static_cast<uint16_t>(a.tailsIn[k * 16 + lr]));
            }
        }

        // lane groups beyond the pass's frames run the same instruction stream on a dummy row (the bit
        // pool is dead in phase 2) and store nothing
        PassLane P;
        P.rowC = active ? reinterpret_cast<uint32_t *>(L.row(mySlot)) : L.pool();
#ifdef DCS_STAMPS_XFORM
        P.stamp = stamp;
#endif
        P.l = lane & ((1 << lpfShift) - 1);
        P.shiftPair = static_cast<uint32_t>(myShift) * 0x00010001u;
        P.paced = paced && s0 + n >= nSlots;            // (the chunk's last pass)

        uint32_t x[16];
        BflyRegs R;
        R.k8000 = 0x8000u; R.k10000 = 0x10000u; R.watch = 0xFFFFu;
        R.k4000 = 0x4000u; R.watchB = 0x7FFF7FFFu;
        asm volatile("" : "+v"(R.k8000), "+v"(R.k10000), "+v"(R.k4000));    // keep them in vector registers (VOP3 takes no literal operand)
        if (xf == DCS_XFORM_94)
            transform94x8(P, W, C, R, x);
        else
            transform93x4(P, W, C, R, x);
        if (s0 == 0) DCS_STAMP(14);

        // tail for the successor = output samples 240..255 (:569-575, :805-812): register 15 of every lane.  Lane
        // groups without a frame write to a spare row of the tail array.
        {
            const int ts = active ? mySlot : FPW;
            if (xf == DCS_XFORM_94)
                tails[ts * 8 + lr] = x[15];
            else
                reinterpret_cast<uint16_t *>(tails)[ts * 16 + lr] = static_cast<uint16_t>(x[15]);
        }
        waveSync();

        if (s0 == 0) DCS_STAMP(15);
        const bool emit = active && !(myFlags & DCS_SLOT_HALO);
        const bool hasPrev = myPrevSlot != DCS_NO_PREV_SLOT;
        // Where the successor or the predecessor lies in another chunk, the frame goes to the rendezvous (see the top of this file):
        // its tail for a successor elsewhere (DCS_SLOT_EXPORT), its own first sample (pair), NOT overlapped, for a tail from
        // elsewhere (DCS_SLOT_IMPORT; such a frame leaves its first 16 samples to whoever arrives second).  The exchanges are issued
        // half way through the PCM stores -- the registers their results take have just come free, and the other half of the stores
        // covers their way to memory and back -- and looked at behind the last store.
        const bool exporter = active && (myFlags & DCS_SLOT_EXPORT) != 0;
        const bool deferred = active && (myFlags & DCS_SLOT_IMPORT) != 0;
#ifndef DCS_RDV_SPLIT_FPW4
#define DCS_RDV_SPLIT_FPW4 12
#endif
        // (stores issued in front of the exchanges: with 4 frames per wavefront -- sixteen lanes unpack a frame, the longest-lived
        // register set -- the exchanges' results only fit behind twelve of them)
        constexpr int kSplit = FPW == 4 ? DCS_RDV_SPLIT_FPW4 : 8;
        // (the lane's output position once more, opaque to the compiler: with 4 frames per wavefront every pass runs the same transform,
        // and the addresses behind the transform, hoisted out of the pass loop as 64-bit pairs, cost the transform its registers)
        int lrA = lr;
        if (FPW == 4)
            asm volatile("" : "+v"(lrA));
        unsigned long long metAsProducer = 0, metAsConsumer = 0;
        auto rendezvous = [&](uint32_t mine0, uint32_t tailOut)
        {
            if (exporter)
                metAsProducer = __hip_atomic_exchange(a.handoff + static_cast<size_t>(chunk) * 16 + lrA, handoffWord(a.epoch, 0u, tailOut),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (deferred)
                metAsConsumer = __hip_atomic_exchange(a.handoff + static_cast<size_t>(myPrevJob) * 16 + lrA, handoffWord(a.epoch, 1u, mine0),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // (The PCM leaves as 15 two- resp. four-byte stores per lane.  Putting a frame's samples in order in its dead tile
        // row first and storing 16 bytes per lane -- 2 resp. 4 store instructions -- was measured twice: 5 % slower, the
        // extra LDS round trip is on the critical path and the narrow stores are not.  Swapping registers r and r + 8
        // with lane l ^ 8 by DPP and storing sample PAIRS, 8 stores instead of 15 for a 1993 frame: no difference.)
        if (xf == DCS_XFORM_94)
        {
            // overlap-add on sample pair m = bitrev3(l) (register 0) (:538-555)
            uint32_t tailPair = tails[(hasPrev ? myPrevSlot : mySlot) * 8 + lr];
            tailPair = hasPrev ? tailPair : extTail;
            const uint32_t mixed = packC(overlapMix(reC(x[0]), C.k[DCS_K94_OVLA] & 0xFFFFu, reC(tailPair), C.k[DCS_K94_OVLB] & 0xFFFFu),
                                         overlapMix(imC(x[0]), C.k[DCS_K94_OVLA] >> 16, imC(tailPair), C.k[DCS_K94_OVLB] >> 16));
            x[0] = deferred ? x[0] : mixed;                             // (a deferred frame keeps the raw pair for the rendezvous)
            uint32_t *out = reinterpret_cast<uint32_t *>(a.pcm) + static_cast<size_t>(myJob) * (DCS_FRAME_SAMPLES / 2) + lrA;
            if (emit)
            {
                if (!deferred)
                    out[0] = x[0];
#pragma unroll
                for (int r = 1 ; r < kSplit ; ++r)
                    out[8 * bitrevN(r, 4)] = x[r];                      // pair 8*bitrev4(r) + bitrev3(l)
            }
            rendezvous(x[0], x[15]);
            if (emit)
            {
#pragma unroll
                for (int r = kSplit ; r < 15 ; ++r)
                    out[8 * bitrevN(r, 4)] = x[r];
                if (a.tailsOut != nullptr && (myFlags & DCS_SLOT_KEEP_TAIL))
                    reinterpret_cast<uint32_t *>(a.tailsOut)[static_cast<size_t>(myJob) * 8 + lrA] = x[15];
            }
            // the second to arrive finishes the consumer's first sample pair
            if (deferred && handoffMeets(metAsConsumer, a.epoch, 0u))
            {
                const uint32_t theirTail = static_cast<uint32_t>(metAsConsumer);
                out[0] = packC(overlapMix(reC(x[0]), C.k[DCS_K94_OVLA] & 0xFFFFu, reC(theirTail), C.k[DCS_K94_OVLB] & 0xFFFFu),
                               overlapMix(imC(x[0]), C.k[DCS_K94_OVLA] >> 16, imC(theirTail), C.k[DCS_K94_OVLB] >> 16));
            }
            if (exporter && handoffMeets(metAsProducer, a.epoch, 1u))
            {
                const uint32_t theirs = static_cast<uint32_t>(metAsProducer);
                reinterpret_cast<uint32_t *>(a.pcm)[static_cast<size_t>(exportNext) * (DCS_FRAME_SAMPLES / 2) + lrA] =
                    packC(overlapMix(reC(theirs), C.k[DCS_K94_OVLA] & 0xFFFFu, reC(x[15]), C.k[DCS_K94_OVLB] & 0xFFFFu),
                          overlapMix(imC(theirs), C.k[DCS_K94_OVLA] >> 16, imC(x[15]), C.k[DCS_K94_OVLB] >> 16));
            }
        }
        else
        {
            // overlap-add on sample i = bitrev4(l) (register 0) (:789-802)
            int tailSample = static_cast<int16_t>(reinterpret_cast<const uint16_t *>(tails)[(hasPrev ? myPrevSlot : mySlot) * 16 + lr]);
            tailSample = hasPrev ? tailSample : sx16(extTail);
            const uint32_t mixed = static_cast<uint32_t>(overlapMix(reC(x[0]), C.k[DCS_K93_OVL] & 0xFFFFu, tailSample, C.k[DCS_K93_OVL] >> 16)) & 0xFFFFu;
            x[0] = deferred ? (x[0] & 0xFFFFu) : mixed;
            int16_t *out = a.pcm + static_cast<size_t>(myJob) * DCS_FRAME_SAMPLES + lrA;
            if (emit)
            {
                if (!deferred)
                    out[0] = static_cast<int16_t>(x[0]);
#pragma unroll
                for (int r = 1 ; r < kSplit ; ++r)
                    out[16 * bitrevN(r, 4)] = static_cast<int16_t>(x[r]);          // sample 16*bitrev4(r) + bitrev4(l)
            }
            rendezvous(x[0], x[15] & 0xFFFFu);
            if (emit)
            {
#pragma unroll
                for (int r = kSplit ; r < 15 ; ++r)
                    out[16 * bitrevN(r, 4)] = static_cast<int16_t>(x[r]);
                if (a.tailsOut != nullptr && (myFlags & DCS_SLOT_KEEP_TAIL))
                    a.tailsOut[static_cast<size_t>(myJob) * 16 + lrA] = static_cast<int16_t>(x[15]);
            }
            if (deferred && handoffMeets(metAsConsumer, a.epoch, 0u))
                out[0] = static_cast<int16_t>(overlapMix(reC(x[0]), C.k[DCS_K93_OVL] & 0xFFFFu, sx16(static_cast<uint32_t>(metAsConsumer)), C.k[DCS_K93_OVL] >> 16));
            if (exporter && handoffMeets(metAsProducer, a.epoch, 1u))
                a.pcm[static_cast<size_t>(exportNext) * DCS_FRAME_SAMPLES + lrA] =
                    static_cast<int16_t>(overlapMix(reC(static_cast<uint32_t>(metAsProducer)), C.k[DCS_K93_OVL] & 0xFFFFu, sx16(x[15]), C.k[DCS_K93_OVL] >> 16));
        }
        waveSync();
        s0 += n;
    }

    DCS_STAMP(6);
}

}   // namespace dcsk

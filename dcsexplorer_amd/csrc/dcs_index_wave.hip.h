// dcs_index_wave.hip.h -- the index pass on the device, ONE WAVEFRONT PER STREAM.
//
// The walk that finds where every frame of a stream starts (DCSDecoderNative::GetStreamInfo, DCSDecoderNative.cpp:1486-1537)
// is serial from frame to frame and from band to band, but not inside a band: a band's codebook is fixed once its
// band-type code is known, so the 64 lanes look up the code that WOULD start at each of the next 64 bit positions at
// once, and the chain through those candidates -- position += length of the code found there -- is followed on the
// scalar unit with one v_readlane and one s_add per symbol (no memory or LDS access on the chain).  The same scheme
// takes the frame header's band-type delta codes (:1780-1834), and the delta / double-delta sample runs of the 1993
// layouts (:2565-2599) become two wavefront sums.  Everything else of the walk is control flow on values all lanes
// share: it runs once per wavefront on the scalar unit, the stream's bits come out of a register window (64 + 64
// dwords of the stream held one per lane, a third block in flight) through v_readlane.
//
// What it writes is what the one-lane walker of dcs_scan.h writes on the host (dcs_index_stream): the same
// DcsFrameIndex per frame, the same DcsStreamInfo including the reference reader's byte pointer (nBytes, computed like
// WinBits / the former DevBits do: the maximum over all looks of position + width).  tests/ and tools/fuzz_parity.py
// hold the two against each other on every layout, on damaged and on truncated streams.
#pragma once
#include <hip/hip_runtime.h>
#include "dcs_common.h"
#include "dcs_scan.h"

namespace dcsidx {

// diagnostic build only (-DDCS_IDX_STAMPS): shader cycles by phase, summed over the streams of a launch
#ifdef DCS_IDX_STAMPS
__device__ unsigned long long g_idxStamps[16];
#define IDX_T0(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#define IDX_ACC(b, k, t0) (b).acc[k] += __builtin_amdgcn_s_memtime() - (t0)
#define IDX_CNT(b, k, n) (b).acc[k] += (n)
#else
#define IDX_T0(name)
#define IDX_ACC(b, k, t0)
#define IDX_CNT(b, k, n)
#endif

constexpr int kWaves = 4;               // wavefronts (= streams) per workgroup; they share the tables in LDS

// Fair shares of a SIMD.  A SIMD serves its OLDEST wavefront first: of eight walks on one SIMD the five oldest run at the pace of a
// lone wavefront and leave (a lone walk uses a fifth of the vector unit), the three youngest get what is left and then finish on a
// SIMD they cannot fill -- 8 192 streams took 3.9 ms where the instructions issued account for 2.4.  So every walk tells its XCD
// how many frames it has left (one atomic add every kPaceFrames frames on a word in the XCD's own L2, the answer read a frame
// later), hears the sum and the number of walks alive, and sets its priority (s_setprio) by where it stands against their mean:
// the walks with the most left to do go first, all of a launch finish together, and streams of unequal length (a corpus) have
// their long ones started on at once.  The words are advice only (no result depends on them) and return to zero by themselves:
// every walk takes back exactly what it has added.  Every launch has words of its own (`paceSlot`): launches that overlap -- lists
// in flight -- are NOT paced against each other, between them the older wavefront goes first as before, so the older list's walk
// finishes first and its planner, packer and decode overlap the younger lists' walks (one set of words for all: eight lists in
// flight 8.5 -> 7.9 x 10^10 samples/s; per launch: see NOTES 42).  Measured (NOTES 42): 8 192 / 6 144 / 4 096 streams x 256 frames 3.96 / 3.24 /
// 2.49 -> 3.20 / 2.65 / 2.20 ms, one list unchanged (1.89); a report every 2 to 16 frames, bands of 1 to 16 frames and two
// levels instead of four all measure within 2 % of each other (variant builds: DCS_EXP_PACE_MODE 1 = two levels, 3 = off).
#ifndef DCS_EXP_PACE_FRAMES
#define DCS_EXP_PACE_FRAMES 8
#endif
#ifndef DCS_EXP_PACE_BAND
#define DCS_EXP_PACE_BAND 8
#endif
#ifndef DCS_EXP_PACE_MODE
#define DCS_EXP_PACE_MODE 0
#endif
constexpr uint32_t kPaceFrames = DCS_EXP_PACE_FRAMES;
constexpr uint32_t kPaceBand = DCS_EXP_PACE_BAND;       // frames ahead of / behind the mean that change the priority by one step
constexpr uint32_t kPaceSlots = 16;     // launches in flight that keep words of their own (a launch's number modulo this)
__device__ unsigned long long g_idxPace[kPaceSlots][8 * 16];    // per launch slot and XCD (128 bytes apart): frames left | walks alive << 32
constexpr int kRingDw = 256;            // per wavefront: LDS mirror of the register window, for per-lane gathers
constexpr int kRecDw = 40;              // staging of one record (37 dwords) or one stream summary (12)
static_assert(sizeof(DcsFrameIndex) == 148 && sizeof(DcsStreamInfo) == 48 && sizeof(DcsFrameDigest) == 8, "record layouts");

struct IndexLds
{
    DcsLdsTables T;
    uint16_t fast94[256];
    uint16_t trie94[DCS_TRIE94_MAX];
    uint8_t multi94[6][1 << DCS_IDX_MULTI_BITS];
    uint32_t ring[kWaves][kRingDw + 4];     // (slots 0..2 once more behind the end: a gather reads four neighbours without wrapping)
    uint32_t rec[kWaves][kRecDw];
};

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(lane))); }
// v_writelane: two scalar operands are one too many for gfx950's constant bus, and this compiler has no builtin that would
// route the lane number through M0; a compare and a select do the same
__device__ __forceinline__ uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t wl(uint32_t old, uint32_t lane, uint32_t val) { return laneId() == lane ? val : old; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// lanes exchange data through LDS in program order; the fence keeps the compiler from moving accesses across
__device__ __forceinline__ void waveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the sixteen lanes of a row (all lanes of the row receive it): quad swap, quad-pair swap, half mirror, mirror
__device__ __forceinline__ uint32_t rowSum16(uint32_t x)
{
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}

// inclusive prefix sum over the sixteen lanes of a row
__device__ __forceinline__ uint32_t rowScan16(uint32_t x)
{
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xF, 0xF, true));   // row_shr:1
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xF, 0xF, true));   // row_shr:2
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xF, 0xF, true));   // row_shr:4
    x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xF, 0xF, true));   // row_shr:8
    return x;
}

// ---------------------------------------------------------------------------------------------------------
// The stream's bits.  All state but W0..W2 is the same on every lane (the compiler keeps it in scalar registers).
// Positions count bits from the dword that holds the stream's first byte, so they fit 32 bits whatever the address.
// Window: lane l of W0 / W1 holds dword B + l / B + 64 + l as 32 MSB-first stream bits (bytes outside the stream
// zero), W2 the block after, requested when the window last moved and not waited for until it moves again; `ring`
// mirrors W0 and W1 in LDS for the gathers.  A 64-bit scalar window (win, have) serves the single-field reads.
// The reference reader's byte pointer (ROMBitPointer, DCSDecoderNative.h:229-289: Peek(n) pulls whole bytes while
// nBits <= n) is computed, not kept: after Peek(n) at bit position p it stands at floor((p + n) / 8) + 1.
// ---------------------------------------------------------------------------------------------------------
struct WaveBits
{
    static constexpr bool kAnalytic = true;
    const uint32_t __attribute__((address_space(1))) *sBase;      // the dword that holds the stream's first byte (global memory: a generic
                                                                   // pointer would make every load a flat one, which LDS waits wait for too)
    uint32_t endByte;           // bytes from sBase to the stream's end
    uint32_t nDwValid;          // dwords that hold stream bytes
    uint32_t payBit = 0;        // first payload bit
    uint32_t pos = 0;           // payload bits consumed
    uint32_t hi = 0;            // max over looks of pos + n
    bool any = false;
    uint32_t B = 0;
    uint32_t posLim = 0;        // the position's dword lies in W0 while pos < posLim (= ((B + 64) << 5) - payBit: one compare per look)
    uint32_t W0 = 0, W1 = 0, W2 = 0;
    uint32_t *ring;
    uint32_t lane;
    uint64_t win = 0;
    int have = 0;
#ifdef DCS_IDX_STAMPS
    unsigned long long acc[12] = { 0 };     // 0 walk, 1 header deltas, 2 huffRun, 3 its chains, 4 record out, 5 slides, 6 runs, 7 symbols, 8 frames, 9 per-band set-up of scan94
#endif

    // dword d of the stream as stream bits; nothing behind the stream's last dword is touched (a damaged stream may
    // announce frames far beyond its bytes; the caller's buffer ends somewhere behind the stream's last dword)
    __device__ __forceinline__ uint32_t load(uint32_t d) const
    {
        uint32_t raw = 0;
        if (d < nDwValid)
            raw = sBase[d];
        const uint32_t lo = d * 4;
        if (lo + 4 > endByte)
        {
            const uint32_t keep = lo >= endByte ? 0u : endByte - lo;        // 0..3 bytes
            raw &= keep == 0 ? 0u : (0xFFFFFFFFu >> (32 - 8 * keep));
        }
        return __builtin_bswap32(raw);
    }
    __device__ __forceinline__ void ringPut(uint32_t d, uint32_t v)
    {
        const uint32_t slot = d & (kRingDw - 1);
        ring[slot] = v;
        if (slot < 3)
            ring[kRingDw + slot] = v;
    }
    __device__ __forceinline__ void reload()
    {
        B = (payBit + pos) >> 5;
        posLim = ((B + 64) << 5) - payBit;
        W0 = load(B + lane);
        W1 = load(B + 64 + lane);
        W2 = load(B + 128 + lane);
        waveSync();
        ringPut(B + lane, W0);
        ringPut(B + 64 + lane, W1);
        waveSync();
    }
    __device__ __forceinline__ void setPayload(uint32_t skewPlusOff)
    {
        payBit = skewPlusOff * 8;
        pos = 0; hi = 0; any = false; have = 0;
        reload();
    }
    // the dword of the current position lies in W0
    __device__ __forceinline__ void ensure()
    {
        if (pos < posLim)
            return;
        uint32_t j = ((payBit + pos) >> 5) - B;
        while (j >= 64)
        {
            IDX_T0(tSlide);
            if (j >= 192)
            {
                reload();
                return;
            }
            W0 = W1; W1 = W2; B += 64; j -= 64;
            posLim += 64 * 32;
            waveSync();
            ringPut(B + 64 + lane, W1);
            waveSync();
            W2 = load(B + 128 + lane);
            IDX_ACC(*this, 5, tSlide);
        }
    }
    __device__ __forceinline__ uint32_t windowDword(uint32_t j) const       // j < 128, the same on every lane
    {
        const uint32_t a = rl(W0, j & 63), b = rl(W1, j & 63);
        return j < 64 ? a : b;
    }
    __device__ __forceinline__ void refill()
    {
        ensure();
        const uint32_t a = payBit + pos;
        const uint32_t j = (a >> 5) - B;
        const uint32_t sh = a & 31;
        win = ((static_cast<uint64_t>(windowDword(j)) << 32) | windowDword(j + 1)) << sh;
        have = 64 - static_cast<int>(sh);
    }
    __device__ __forceinline__ uint32_t look(int n)
    {
        if (have < n)
            refill();
        return n == 0 ? 0u : static_cast<uint32_t>(win >> 32) >> (32 - n);
    }
    __device__ __forceinline__ uint32_t peek(int n)
    {
        any = true;
        hi = umax(hi, pos + static_cast<uint32_t>(n));
        return look(n);
    }
    __device__ __forceinline__ void consume(int n)
    {
        win <<= n;
        have -= n;
        pos += static_cast<uint32_t>(n);
    }
    __device__ __forceinline__ void took(int n)
    {
        consume(n);
        any = true;
        hi = umax(hi, pos);
    }
    __device__ __forceinline__ uint32_t get(int n)
    {
        const uint32_t r = peek(n);
        consume(n);
        return r;
    }
    // `count` fields of `width` bits whose values nobody needs (the last field's look reaches exactly its own end)
    __device__ __forceinline__ void skipRun(int count, int width)
    {
        if (count <= 0 || width <= 0)
            return;
        any = true;
        const uint32_t total = static_cast<uint32_t>(count) * static_cast<uint32_t>(width);
        pos += total;
        hi = umax(hi, pos);
        if (total <= static_cast<uint32_t>(have))
        {
            win = total >= 64 ? 0 : win << total;
            have -= static_cast<int>(total);
        }
        else
            have = 0;
    }
    // the next 32 bits at `off` bits behind the current position, per lane (ensure() first; off < 1 900)
    __device__ __forceinline__ uint32_t gather32(uint32_t off) const
    {
        const uint32_t a = payBit + pos + off;
        const uint32_t d = a >> 5;
        const uint32_t *p = ring + (d & (kRingDw - 1));
        const uint32_t h = p[0], l = p[1];
        return static_cast<uint32_t>((((static_cast<uint64_t>(h) << 32) | l) << (a & 31)) >> 32);
    }
    // the same for TWO positions per lane: `off` and `off` + 64 bits behind the current position (128 candidates a look)
    __device__ __forceinline__ void gather32x2(uint32_t off, uint32_t &lo, uint32_t &hi64) const
    {
        const uint32_t a = payBit + pos + off;
        const uint32_t d = a >> 5;
        const uint32_t *p = ring + (d & (kRingDw - 1));
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
        const uint32_t sh = a & 31;
        lo = static_cast<uint32_t>((((static_cast<uint64_t>(w0) << 32) | w1) << sh) >> 32);
        hi64 = static_cast<uint32_t>((((static_cast<uint64_t>(w2) << 32) | w3) << sh) >> 32);
    }
    __device__ __forceinline__ uint32_t bitPos() const { return pos; }
    __device__ __forceinline__ uint32_t bytesFetched(uint32_t payOff) const { return any ? payOff + (hi >> 3) + 1 : payOff; }
};

// One walk: the stream's constants, what the decoder carries from frame to frame, and the record under construction.
struct Walk
{
    WaveBits b;
    const IndexLds *L;
    uint32_t lane;
    int os = 0, format = 0, nBands = 0;
    bool type1 = false, sub0 = false;
    uint32_t vHeader = 0;       // lane i < 16: header byte i
    uint32_t vBandType = 0;     // lane i < 16: AudioStream::bandTypeBuf[i]
    uint32_t vCount = 0, vInc = 0;              // 1994+, lane i < 16: samples of band i (:1848-1862) and their spacing
    uint32_t vBB = 0, vSc1 = 0, vInputs = 0;    // OS93a Type 1: the stream's band-bits codebook, the first level of the scale
                                                // codebook, inputs per band -- one entry per lane
    uint32_t err = 0;
    unsigned long long *pace = nullptr;         // this XCD's word of g_idxPace
    // the record
    uint32_t vSplitLo = 0, vSplitHi = 0;        // lane k < 15: split[k] = bitDelta | prv << 16, prvDelta | state << 16
    uint32_t vRecBT = 0;                        // lane i < 16: byte i of the record's bandType field
    uint32_t hdrBits = 0, preAdj = 0;

    __device__ __forceinline__ void fatal() { err |= DCS_FRAME_FATAL | DCS_FRAME_STOP; }
    __device__ __forceinline__ void putSplit(int band, uint32_t bitDelta, uint32_t prv, uint32_t prvDelta, uint32_t state)
    {
        if (band < 1 || band > 15)
            return;
        vSplitLo = wl(vSplitLo, static_cast<uint32_t>(band - 1), (bitDelta & 0xFFFFu) | (prv << 16));
        vSplitHi = wl(vSplitHi, static_cast<uint32_t>(band - 1), (prvDelta & 0xFFFFu) | (state << 16));
    }
};

// The chain through the candidates: state = bits walked | samples left - 1 << 16, v = per lane the length of the code
// that starts there minus, in the upper half, the samples it stands for.  Ends when the samples run out (sign bit) or
// the walk leaves the candidates (bit 6 or 7); `se` = the last candidate taken.  Written out because the loop the
// compiler makes of it pays a taken branch per symbol (62 cycles a symbol measured, against ~20 for this form: eight
// steps in a row whose exits are branches NOT taken).  (An exit through the CARRY of the add -- entries that leave the
// candidates poisoned, the sample count running up to an overflow -- saves the s_and per step and was measured in round 4:
// 2.21 against 2.16 ms for one list, nothing at 8 192 streams; the poisoning costs the look what the steps save.)
#define DCS_CHAIN_STEP(MASK)                            \
        "v_readlane_b32 %[se], %[v], %[st]\n\t"          \
        "s_add_u32 %[st], %[st], %[se]\n\t"              \
        "s_and_b32 %[t], %[st], " MASK "\n\t"            \
        "s_cbranch_scc1 2f\n\t"
#define DCS_CHAIN_BODY(MASK)                                                                        \
    uint32_t t;                                                                                     \
    asm volatile("s_nop 0\n"                                                                        \
                 "1:\n\t"                                                                           \
                 DCS_CHAIN_STEP(MASK) DCS_CHAIN_STEP(MASK) DCS_CHAIN_STEP(MASK) DCS_CHAIN_STEP(MASK) \
                 DCS_CHAIN_STEP(MASK) DCS_CHAIN_STEP(MASK) DCS_CHAIN_STEP(MASK) DCS_CHAIN_STEP(MASK) \
                 "s_branch 1b\n"                                                                    \
                 "2:\n\t"                                                                           \
                 : [st] "+s"(state), [se] "=&s"(se), [t] "=&s"(t)                                   \
                 : [v] "v"(v)                                                                       \
                 : "scc");
// (v_readlane takes the lane from the low six bits of `state`: candidates 0..63 of the look)
__device__ __forceinline__ void chain(uint32_t v, uint32_t &state, uint32_t &se) { DCS_CHAIN_BODY("0x80000040") }
// the same through candidates 64..127 of a two-candidates-per-lane look (`v` = lane l: candidate 64 + l): ends when the
// samples run out or the walk leaves the 128 candidates (bit 7)
__device__ __forceinline__ void chainHi(uint32_t v, uint32_t &state, uint32_t &se) { DCS_CHAIN_BODY("0x80000080") }
#undef DCS_CHAIN_BODY
#undef DCS_CHAIN_STEP

// A run of Huffman-coded samples (:2186-2225): symbols from the current position until the samples are accounted for (a
// two-zeros code counts for two).  `book` = the codebook's direct look-up table on the next 32 - `shift` bits, `multi` = its
// several-codes table (DcsDevTables::multi94): while more than DCS_IDX_MULTI_SAMPLES samples are to go a step of the chain
// takes all the codes of such an entry, the band's last codes go one by one (so the two-zeros rule and the run's last look,
// the one that can decide nBytes, are the reference's).
// The run's state is ONE word from its first look to its last (round 5: the bookkeeping between the chains was 2/3 of the
// walk's scalar instructions): S = bits walked in this look | (samples to go - 1) << 16 -- the single-code chains' state as
// it is, the several-codes chains' after subtracting DCS_IDX_MULTI_SAMPLES << 16.  In goes (samples - 1) << 16; out comes
// 0xFFFF0000 (all samples accounted for) or 0xFFFE0000 (the last code was a two-zeros code with one sample to go).
__device__ __forceinline__ uint32_t huffRunS(WaveBits &b, const uint16_t *book, uint32_t shift, const uint8_t *multi, uint32_t S)
{
    constexpr uint32_t kM = static_cast<uint32_t>(DCS_IDX_MULTI_SAMPLES) << 16;
    static_assert((DCS_IDX_MULTI_SAMPLES & (DCS_IDX_MULTI_SAMPLES - 1)) == 0, "the test below masks samples-to-go with a power of two");
    const uint32_t maxBits = 32u - shift;
    b.any = true;
    IDX_T0(tRun);
    IDX_CNT(b, 6, 1);
    IDX_CNT(b, 7, (S >> 16) + 1);
    do
    {
        b.ensure();
        // lane l: the code(s) that would start l bits and 64 + l bits from here -- length, and minus the samples in the upper
        // half.  128 candidates a look: a band of sixteen samples is 80 to 90 bits on average, so most runs need one look
        // (with 64 candidates the second look's two LDS round trips were a quarter of a run)
        uint32_t wLo, wHi;
        b.gather32x2(b.lane, wLo, wHi);
        const uint32_t eLo = book[wLo >> shift], eHi = book[wHi >> shift];
        const uint32_t mLo = multi[wLo >> (32 - DCS_IDX_MULTI_BITS)], mHi = multi[wHi >> (32 - DCS_IDX_MULTI_BITS)];
        asm volatile("" :: "v"(eLo), "v"(mLo), "v"(eHi), "v"(mHi));     // all four reads on their way before anything waits
        const uint32_t vSingleLo = ((eLo >> 8) & 0x1Fu) - ((eLo >> 13) == 2 ? 0x20000u : 0x10000u);
        const uint32_t vSingleHi = ((eHi >> 8) & 0x1Fu) - ((eHi >> 13) == 2 ? 0x20000u : 0x10000u);
        const uint32_t vMultiLo = (mLo & 15u) - ((mLo >> 4) << 16);
        const uint32_t vMultiHi = (mHi & 15u) - ((mHi >> 4) << 16);
        // A chain ends when the samples run out (sign) or the walk leaves its 64 candidates (bit 6 for the first half, bit 7
        // for the second).
        uint32_t se = 0;
        IDX_T0(tChain);
        if (S >= kM)                                        // more than DCS_IDX_MULTI_SAMPLES samples to go
        {
            S -= kM;
            chain(vMultiLo, S, se);
            if (static_cast<int32_t>(S) >= 0)               // (left the first 64 candidates with samples to go)
                chainHi(vMultiHi, S, se);
            S += kM;
        }
        if ((S & ((0xFFFF0000u & ~(kM - 0x10000u)) | 0xFF80u)) == 0)     // at most DCS_IDX_MULTI_SAMPLES to go, and inside the 128 candidates
        {
            if ((S & 64u) == 0)
                chain(vSingleLo, S, se);
            if (static_cast<int32_t>(S) >= 0)               // (the first half left, or never entered)
                chainHi(vSingleHi, S, se);
            b.hi = umax(b.hi, b.pos + maxBits + (S & 0xFFFFu) - (se & 0xFFFFu));       // the last symbol's look
        }
        IDX_ACC(b, 3, tChain);
        const uint32_t off = S & 0xFFFFu;
        b.pos += off;
        S -= off;
    }
    while (static_cast<int32_t>(S) >= 0);
    b.have = 0;
    IDX_ACC(b, 2, tRun);
    return S;
}
// the form band 15's two halves use: `rem` samples -> what is left: 0, or -1 when the last code was a two-zeros code with one sample to go
__device__ __forceinline__ int huffRun(WaveBits &b, const uint16_t *book, uint32_t maxBits, const uint8_t *multi, int rem)
{
    return (static_cast<int32_t>(huffRunS(b, book, 32u - maxBits, multi, static_cast<uint32_t>(rem - 1) << 16)) >> 16) + 1;
}

// The frame header of a 1994+ frame: one band-type delta code per populated band (:1780-1834), lane k < nBands
// receives code k's payload (delta + 16).  Candidates through the first-level table; the chain takes sixteen steps
// without a test (it does not move on from a candidate beyond 55, so it stays inside the 64) and only marks where it has been;
// which code is whose follows from the marks afterwards (a lane's rank among them), and the payloads travel to
// their bands' lanes in one ds_permute.  A code longer than eight bits (rare) sends the rest of the header through the
// trie on the scalar unit.
__device__ __forceinline__ uint32_t headerDeltas94(Walk &s)
{
    WaveBits &b = s.b;
    const uint32_t lane = s.lane;
    uint32_t vDelta = 16;
    int band = 0;
    while (band < s.nBands)
    {
        b.ensure();
        const uint32_t e = s.L->fast94[b.gather32(lane) >> 24];
        const bool leaf = (e & 0x8000u) != 0;
        // (the chain does not get past a long code, nor past candidate 55: a code is at most eight bits, so sixteen steps that
        // never start beyond 55 stay inside the 64 candidates without a test)
        const uint32_t vLen = (leaf && lane < 56) ? (e >> 8) & 0xFu : 0u;
        unsigned long long marks = 0;
        uint32_t off = 0, t;
#define DCS_HDR_STEP                                    \
            "s_bitset1_b64 %[m], %[off]\n\t"             \
            "v_readlane_b32 %[t], %[v], %[off]\n\t"       \
            "s_add_u32 %[off], %[off], %[t]\n\t"
        asm volatile("s_nop 0\n\t"
                     DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP
                     DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP DCS_HDR_STEP
                     : [m] "+s"(marks), [off] "+s"(off), [t] "=&s"(t)
                     : [v] "v"(vLen)
                     : "scc");
#undef DCS_HDR_STEP
        // the codes this round takes: those that start below candidate 56, as many as there are bands left
        const int want = s.nBands - band;
        const int low = __builtin_popcountll(marks & 0x00FFFFFFFFFFFFFFull);
        const int n = want < low ? want : low;
        const bool marked = ((marks >> lane) & 1) != 0;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(marks >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(marks), 0u));
        const bool mine = marked && static_cast<int>(rank) < n;
        if (__ballot(mine && !leaf) != 0)
        {
            // a long code among them: this and the remaining codes one by one
            b.have = 0;
            for ( ; band < s.nBands ; ++band)
                vDelta = wl(vDelta, static_cast<uint32_t>(band), static_cast<uint32_t>(dcsReadVlcFast(b, s.L->fast94, s.L->trie94)));
            break;
        }
        const uint32_t got = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>((mine ? static_cast<uint32_t>(band) + rank : 63u) << 2),
                                                                                 static_cast<int>(e & 0xFFu)));
        if (static_cast<int>(lane) >= band && static_cast<int>(lane) < band + n)
            vDelta = got;
        // where the next code starts: behind the sixteenth, or at the mark of rank n (which may be the one the chain stopped at)
        const uint32_t adv = n == 16 ? off : static_cast<uint32_t>(__builtin_ctzll(__ballot(marked && static_cast<int>(rank) == n)));
        b.pos += adv;
        b.have = 0;
        band += n;
    }
    if (s.nBands > 0)
    {
        b.any = true;
        b.hi = umax(b.hi, b.pos);       // (a code's one-bit looks reach its last bit)
    }
    return vDelta;
}

// --- 1994+ frame (:1679-2261; the walk of dcsScan94, dcs_scan.h) ---------------------------------------------------
// Once the frame header is read, everything about a band but the length of its Huffman-coded samples follows from its
// band-type code and the stream header: lane b works that out for band b (one look-up in the decode kernel's set-up
// table, DcsLdsTables::band94), output indices and the bits of the fixed-width bands become prefix sums over the lanes,
// and what is left to do one after the other are the Huffman-coded bands.
__device__ void scan94(Walk &s)
{
    WaveBits &b = s.b;
    const DcsLdsTables &T = s.L->T;
    const uint32_t lane = s.lane;
    const uint32_t frameStart = b.pos;

    // Type 1 indexes its pre-adjust map with the previous frame's codes of bands 0..2 (:1744-1773)
    if (s.type1)
    {
        if (__ballot(lane < 3 && s.vBandType > 15) != 0) { s.fatal(); return; }
        const uint32_t p = lane < 3 ? T.preAdj94[(s.sub0 ? 0u : 16u) + (s.vBandType & 15u)] : 0u;
        s.preAdj = rl(p, 0) | (rl(p, 1) << 4) | (rl(p, 2) << 8);
    }

    IDX_T0(tHdr);
    const uint32_t vDelta = headerDeltas94(s);
    IDX_ACC(b, 1, tHdr);
    IDX_T0(tSetup);
    if (static_cast<int>(lane) < s.nBands)
        s.vBandType = (s.vBandType + vDelta - 16u) & 0xFFFFu;
    const uint32_t hdrBits = b.pos - frameStart;
    s.hdrBits = hdrBits & 0xFFFFu;
    s.vRecBT = s.vBandType > 255u ? 255u : s.vBandType;

    // lane b: band b (:1848-2005 through the table: code translation, codebook, fixed width, no such code)
    const uint32_t raw = s.vBandType;
    const uint32_t at = s.type1 ? (lane < 3 ? 0u : lane < 6 ? 17u : 34u) + (raw < 16u ? raw : 16u)
                                : static_cast<uint32_t>(DCS_B94_TYPE0) + (raw < 17u ? raw : 17u);
    const uint32_t d = T.band94[lane < 16 ? at : 0u];
    const bool active = static_cast<int>(lane) < s.nBands;
    const unsigned long long fatalBands = __ballot(active && (d & DCS_B94_FATAL) != 0);
    const uint32_t procEnd = fatalBands != 0 ? static_cast<uint32_t>(__builtin_ctzll(fatalBands)) : static_cast<uint32_t>(s.nBands);
    const bool coded = lane < procEnd && raw != 0;
    const bool rawBand = coded && (d & DCS_B94_RAW) != 0;
    const bool huffBand = coded && (d & DCS_B94_RAW) == 0;
    const uint32_t width = 32u - ((d >> 11) & 31u);             // fixed sample width, or the codebook's look-ahead
    const uint32_t adv = lane >= procEnd ? 0u : raw == 0 ? s.vCount : s.vCount * s.vInc;      // (the halved count, :1886)
    const uint32_t fixedBits = rawBand ? s.vCount * width : 0u;
    const uint32_t advIncl = rowScan16(adv), fixedIncl = rowScan16(fixedBits);
    const uint32_t outIdxB = 1u + advIncl - adv;                // output index at the band's start
    const uint32_t fixedBefore = fixedIncl - fixedBits;         // bits of the fixed-width bands before it

    // What a Huffman-coded band's walk needs, a word each (the scalar unit is the walk's busiest at saturation, the vector unit is
    // not: five v_readlane a run instead of two and six scalar instructions that took their fields apart): the bits of the
    // fixed-width bands before it, its state word (samples - 1 << 16), its codebook (byte offset in the tables), its several-codes
    // table (of the six: look-aheads 2, 3, 5, 7, 8, 9, :2005) and 32 - its look-ahead
    const uint32_t vRunS = (s.vCount - 1u) << 16;
    const uint32_t vRunBook = (d & 0x7FFu) * 2u;
    const uint32_t vRunMulti = (width <= 3u ? width - 2u : width == 5u ? 2u : width - 4u) << DCS_IDX_MULTI_BITS;
    const uint32_t vRunShift = 32u - width;

    // The Huffman-coded bands, one after the other.  The walk carries q = position - bits of the fixed-width bands before it
    // (the runs are contiguous but for those), lane h keeps q behind band h's run; where every band begins follows from
    // that afterwards (a maximum over the lanes below: q only grows).
    const uint32_t base = frameStart + hdrBits;
    uint32_t q = base, vQ = 0, runEnds = 0xFFFFFFFFu;
    uint32_t midBit = 0, midIdx = 0;
    IDX_ACC(b, 9, tSetup);
    IDX_T0(tLoop);
    const uint32_t huffMask = static_cast<uint32_t>(__ballot(huffBand));
    for (uint32_t left = huffMask & 0x7FFFu ; left != 0 ; left &= left - 1)
    {
        const uint32_t h = static_cast<uint32_t>(__builtin_ctz(left));
        const uint32_t G = rl(fixedBefore, h);
        const uint16_t *book = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(&T) + rl(vRunBook, h));
        const uint8_t *multi = &s.L->multi94[0][0] + rl(vRunMulti, h);
        b.pos = q + G;
        runEnds &= huffRunS(b, book, rl(vRunShift, h), multi, rl(vRunS, h));
        q = b.pos - G;
        vQ = lane == h ? q : vQ;
    }
    if ((runEnds >> 16) != 0xFFFFu)
        s.err |= DCS_FRAME_STOP;                                // two zeros with one slot left (:2213-2218)
    if ((huffMask & 0x8000u) != 0)
    {
        // band 15 in two halves; where the second one starts is recorded (dcsPutMid15, dcs_scan.h)
        const uint32_t G = rl(fixedBefore, 15);
        const uint16_t *book = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(&T) + rl(vRunBook, 15));
        const uint8_t *multi = &s.L->multi94[0][0] + rl(vRunMulti, 15);
        const uint32_t maxBits = 32u - rl(vRunShift, 15);
        const int count = static_cast<int>(rl(s.vCount, 15));
        b.pos = q + G;
        const int lim = count / 2;
        const int inc = static_cast<int>(rl(s.vInc, 15));
        int i = lim + huffRun(b, book, maxBits, multi, count - lim);
        midBit = (b.pos - frameStart) & 0xFFFFu;
        midIdx = ((rl(outIdxB, 15) + static_cast<uint32_t>((count - i) * inc)) & 0x1FFu) | (i < lim ? DCS_MID15_STRADDLE : 0u);
        if (i > 0 && huffRun(b, book, maxBits, multi, i) < 0)
            s.err |= DCS_FRAME_STOP;
        q = b.pos - G;
    }
    // lane b: bits of the Huffman-coded bands before band b = the largest q among the lanes below, from `base`
    uint32_t vHuffBefore;
    {
        uint32_t m = vQ;
        m = umax(m, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(m), 0x111, 0xF, 0xF, true)));     // row_shr:1
        m = umax(m, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(m), 0x112, 0xF, 0xF, true)));     // row_shr:2
        m = umax(m, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(m), 0x114, 0xF, 0xF, true)));     // row_shr:4
        m = umax(m, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(m), 0x118, 0xF, 0xF, true)));     // row_shr:8
        m = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(m), 0x111, 0xF, 0xF, true));             // exclusive
        vHuffBefore = umax(m, base) - base;
    }
    const uint32_t huffBits = q - base;
    IDX_ACC(b, 10, tLoop);
    IDX_T0(tTail);
    b.pos = base + rl(fixedIncl, 15) + huffBits;
    b.have = 0;
    // the fixed-width bands' looks: the last one's reaches furthest, exactly to its end
    const uint32_t bitDelta = hdrBits + fixedBefore + vHuffBefore;
    const uint32_t rawBands = static_cast<uint32_t>(__ballot(rawBand));
    if (rawBands != 0)
    {
        const uint32_t r = 31u - static_cast<uint32_t>(__builtin_clz(rawBands));
        b.any = true;
        b.hi = umax(b.hi, frameStart + rl(bitDelta, r) + rl(fixedBits, r));
    }
    if (lane == 15 && rawBand)
    {
        const uint32_t second = s.vCount / 2;
        midBit = (bitDelta + (s.vCount - second) * width) & 0xFFFFu;
        midIdx = (outIdxB + (s.vCount - second) * s.vInc) & 0x1FFu;
    }
    // split[b - 1] = the state at the start of band b: every band walked, and the one that was fatal
    const bool recorded = lane >= 1 && static_cast<int>(lane) < s.nBands && lane <= procEnd;
    uint32_t lo = recorded ? (bitDelta & 0xFFFFu) : 0u;
    uint32_t hiw = recorded ? (outIdxB & 0x1FFu) << 16 : 0u;
    if (lane == 15 && coded)
    {
        lo |= midBit << 16;
        hiw |= midIdx;
    }
    s.vSplitLo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), 0x101, 0xF, 0xF, true));     // row_shl:1: lane k takes lane k + 1
    s.vSplitHi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hiw), 0x101, 0xF, 0xF, true));
    if (fatalBands != 0)
        s.fatal();
    IDX_ACC(b, 11, tTail);
}

// --- 1993 frame, Type 0 and OS93b Type 1 (:2293-2615; dcsScan93) ------------------------------------------------------
__device__ void scan93(Walk &s)
{
    WaveBits &b = s.b;
    const DcsLdsTables &T = s.L->T;
    const uint32_t lane = s.lane;
    const bool type1 = s.type1;
    const uint32_t frameStart = b.pos;
    bool first = true, reuse = false;
    int code = 0;
    int subType = type1 ? 0 : 2;
    uint32_t prv = 0, prvDelta = 0;
    int outIdx = 1;

    s.vRecBT = s.vBandType > 255u ? 255u : s.vBandType;

    for (int band = 0 ; band < s.nBands ; ++band)
    {
        s.putSplit(band, b.pos - frameStart, prv, prvDelta,
                   (static_cast<uint32_t>(outIdx) & 0x1FFu) | (static_cast<uint32_t>(subType) << 9) | (reuse ? 0x800u : 0u));
        const uint32_t hb = rl(s.vHeader, static_cast<uint32_t>(band)) & 0x7Fu;
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
            reuse = b.get(1) != 0;
        if (!reuse)
        {
            if (!type1)
            {
                if (b.get(1))
                    subType = b.get(1) ? (subType + 1) % 3 : (subType + 2) % 3;     // :2402-2414
                code = static_cast<int>(b.get(4));
            }
            else
            {
                int v = dcsReadVlcFast(b, T.fast93, T.trie93);
                if (v < 0x1E)
                    v -= 0x0F;                                      // :2668-2681
                else
                {
                    v -= 0x2E;
                    subType = subType != 0 ? 0 : 1;
                }
                const uint32_t bt = (rl(s.vBandType, static_cast<uint32_t>(band)) + static_cast<uint32_t>(v)) & 0xFFFFu;
                s.vBandType = wl(s.vBandType, static_cast<uint32_t>(band), bt);
                code = static_cast<int>(bt);
            }
        }

        if (code == 0)
        {
            reuse = true;                                           // :2455
            if (subType == 0) { outIdx += stride; prv = 0; prvDelta = 0; }
            else if (subType == 1) { prvDelta = 0; outIdx += nSamples * inc + fixup; }
            else
            {
                prv = (prv + static_cast<uint32_t>(nSamples) * prvDelta) & 0xFFFFu;
                outIdx += nSamples * inc + fixup;
            }
        }
        else
        {
            const int width = code + (type1 ? 0 : 1);
            if (width > 16) { s.fatal(); return; }
            if (subType == 0)
            {
                // directly coded samples: only the last two are carried on (:2565-2599)
                uint32_t last = 0, last2 = 0;
                const int skipped = nSamples > 2 ? nSamples - 2 : 0;
                b.skipRun(skipped, width);
                for (int i = skipped ; i < nSamples ; ++i)
                {
                    uint32_t in = b.get(width);
                    if (in & (1u << (width - 1)))
                        in |= 0xFFFFFFFFu << width;
                    in &= 0xFFFF;
                    last2 = last; last = in;
                }
                prv = last;
                prvDelta = (last - last2) & 0xFFFF;
            }
            else
            {
                // delta (1) and double-delta (2) coding: lane i reads sample i; what is carried on are sums
                b.ensure();
                uint32_t x = b.gather32(lane * static_cast<uint32_t>(width)) >> (32 - width);
                if (x & (1u << (width - 1)))
                    x |= 0xFFFFFFFFu << width;
                if (static_cast<int>(lane) >= nSamples)
                    x = 0;
                const uint32_t sum = rl(rowSum16(x), 0);
                if (subType == 1)
                {
                    prvDelta = rl(x, static_cast<uint32_t>(nSamples - 1)) & 0xFFFFu;
                    prv = (prv + sum) & 0xFFFFu;
                }
                else
                {
                    const uint32_t weighted = rl(rowSum16(x * (static_cast<uint32_t>(nSamples) - lane)), 0);
                    prv = (prv + static_cast<uint32_t>(nSamples) * prvDelta + weighted) & 0xFFFFu;
                    prvDelta = (prvDelta + sum) & 0xFFFFu;
                }
                b.any = true;
                b.pos += static_cast<uint32_t>(nSamples * width);
                b.hi = umax(b.hi, b.pos);
                b.have = 0;
            }
            outIdx += nSamples * inc + fixup;
        }
        first = false;
    }
}

// --- OS93a Type 1 frame (:2831-3032; dcsScan93a) -------------------------------------------------------------------------
__device__ void scan93a(Walk &s)
{
    WaveBits &b = s.b;
    const DcsLdsTables &T = s.L->T;
    const int numBands = s.nBands;
    const uint32_t frameStart = b.pos;
    int prvScale = 0x1A;
    int outIdx = 0;
    bool ended = false;

    s.vRecBT = 0;
    for (int band = 0 ; band < numBands ; ++band)
    {
        const uint32_t state = (static_cast<uint32_t>(outIdx) & 0x1FFu) | (ended ? 0x800u : 0u);
        s.putSplit(band, b.pos - frameStart, static_cast<uint32_t>(prvScale) & 0xFFFFu, 0, state);
        if (band == 16 || band == 17)
        {
            // the records of bands 16 and 17 travel in the record's 16 bandType bytes
            const uint32_t at = static_cast<uint32_t>(band - 16) * 8;
            const uint32_t bitDelta = b.pos - frameStart;
            s.vRecBT = wl(s.vRecBT, at + 0, bitDelta & 0xFFu);
            s.vRecBT = wl(s.vRecBT, at + 1, (bitDelta >> 8) & 0xFFu);
            s.vRecBT = wl(s.vRecBT, at + 2, static_cast<uint32_t>(prvScale) & 0xFFu);
            s.vRecBT = wl(s.vRecBT, at + 3, (static_cast<uint32_t>(prvScale) >> 8) & 0xFFu);
            s.vRecBT = wl(s.vRecBT, at + 6, state & 0xFFu);
            s.vRecBT = wl(s.vRecBT, at + 7, (state >> 8) & 0xFFu);
        }
        if (ended)
            continue;
        if (band >= 18) { s.fatal(); return; }
        const int numInputs = static_cast<int>(rl(s.vInputs, static_cast<uint32_t>(band)));
        const uint32_t e = rl(s.vBB, b.peek(4));
        b.get(static_cast<int>(e >> 8));
        const int bandBits = static_cast<int>(e & 0xFF);
        if (bandBits == 0xFF)
        {
            ended = true;
            continue;
        }
        outIdx += numInputs * 2;
        if (bandBits == 0)
            continue;
        uint32_t sc = rl(s.vSc1, b.peek(4));
        b.get(static_cast<int>((sc >> 8) & 0xF));
        if ((sc & 0xFF) == 0xFF)
        {
            sc = uni(T.scaleCb93a[((sc >> 12) << 4) + b.peek(4)]);
            b.get(static_cast<int>((sc >> 8) & 0xF) - 4);
        }
        int scaleCode = prvScale + static_cast<int>(sc & 0xFF) - 1 + bandBits * 2;
        if (scaleCode > 0x39)
            scaleCode -= 0x36;
        prvScale = scaleCode - bandBits * 2;
        b.skipRun(numInputs, bandBits);
    }
}

// the frames of one stream, one after the other: KIND 0 = 1993 Type 0 / OS93b Type 1, 1 = OS93a Type 1, 2 = 1994+
template <int KIND>
__device__ __forceinline__ void walkFrames(Walk &s, uint32_t nFrames, DcsFrameIndex *outRec, DcsFrameDigest *outDigest, uint32_t &valid, uint32_t &payloadBits)
{
    const uint32_t lane = s.lane;
    // (the walk's place among the others of its XCD: see g_idxPace)
    if (lane == 0)
        __hip_atomic_fetch_add(s.pace, static_cast<unsigned long long>(nFrames) | (1ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long vOthers = 0;
    uint32_t credited = 0;
    for (uint32_t f = 0 ; f < nFrames ; ++f)
    {
        if ((f & (kPaceFrames - 1)) == 0 && f != 0)
        {
            if (lane == 0)
                vOthers = __hip_atomic_fetch_add(s.pace, 0ull - kPaceFrames, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            credited = f;
        }
        else if ((f & (kPaceFrames - 1)) == 1 && f != 1)
        {
            const uint32_t sum = rl(static_cast<uint32_t>(vOthers), 0), alive = rl(static_cast<uint32_t>(vOthers >> 32), 0);
            const int32_t ahead = static_cast<int32_t>((nFrames - f) * alive - sum);       // > 0: more left than the mean
            const int32_t band = static_cast<int32_t>(alive * kPaceBand);
#if DCS_EXP_PACE_MODE == 0
            if (ahead >= band) __builtin_amdgcn_s_setprio(3);
            else if (ahead >= 0) __builtin_amdgcn_s_setprio(2);
            else if (ahead >= -band) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#elif DCS_EXP_PACE_MODE == 1
            (void)band;
            if (ahead >= 0) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#else
            (void)ahead; (void)band;
#endif
        }
        const uint32_t frameBit = s.b.pos;
        IDX_T0(tWalk);
        IDX_CNT(s.b, 8, 1);
        s.err = 0;
        s.vSplitLo = 0; s.vSplitHi = 0; s.vRecBT = 0; s.hdrBits = 0; s.preAdj = 0;
        if (KIND == 0) scan93(s);
        else if (KIND == 1) scan93a(s);
        else scan94(s);
        IDX_ACC(s.b, 0, tWalk);
        IDX_T0(tOut);
        const uint32_t nBits = (s.b.pos - frameBit) & 0xFFFFu;
        const uint32_t flags = ((s.err << 4) | (s.err != 0 ? DCS_IDX_SERIAL : 0u)) & 0xFFu;
        // the record, every lane its part (nothing waits for these stores)
        {
            uint8_t *const r8 = reinterpret_cast<uint8_t *>(outRec + valid);
            uint32_t *const r32 = reinterpret_cast<uint32_t *>(r8);
            if (lane < 16)
                r8[8 + lane] = static_cast<uint8_t>(s.vRecBT);
            if (lane < 15)
            {
                r32[7 + 2 * lane] = s.vSplitLo;
                r32[8 + 2 * lane] = s.vSplitHi;
            }
            if (lane == 0)
            {
                r32[0] = frameBit;
                r32[1] = nBits | (s.hdrBits << 16);
                r32[6] = (s.preAdj & 0xFFFFu) | (static_cast<uint32_t>(s.nBands) << 16) | (flags << 24);
            }
        }
        if (outDigest != nullptr && lane == 0)
            outDigest[valid] = DcsFrameDigest{ frameBit, static_cast<uint16_t>(nBits), static_cast<uint8_t>(s.nBands), static_cast<uint8_t>(flags) };
        ++valid;
        IDX_ACC(s.b, 4, tOut);
        payloadBits = s.b.pos;
        if (s.err != 0)
            break;                  // the reference stops the channel on the next tick (:95-116)
    }
    if (lane == 0)
        __hip_atomic_fetch_add(s.pace, 0ull - (static_cast<unsigned long long>(nFrames - credited) | (1ull << 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// where a stream's results go when the streams of one launch belong to different owners (the pipeline's lists, each
// with buffers of its own)
struct StreamOut
{
    DcsFrameIndex *records;
    DcsFrameDigest *digest;     // may be null
    DcsStreamInfo *info;
};

// stream k of the launch is walked by wavefront k
// (eight wavefronts per SIMD: the walk's scalar state would otherwise take 106 SGPRs, and a gfx9 SIMD's 800 hold only seven such
// wavefronts -- a launch of 8 192 streams, eight per SIMD, then ran in two generations; three more spilled SGPRs are cheaper)
__global__ __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void dcsIndexWaveKernel(uintptr_t blobBase, const DcsStreamLoc *locs, uint32_t nStreams,
                                                                  const DcsDevTables *tables, DcsFrameIndex *out, DcsStreamInfo *infos,
                                                                  DcsFrameDigest *digest, const StreamOut *outs, uint32_t paceSlot)
{
    __shared__ IndexLds L;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&tables->lds);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&L.T);
        for (uint32_t i = threadIdx.x ; i < sizeof(DcsLdsTables) / 4 ; i += blockDim.x)
            dst[i] = src[i];
        for (uint32_t i = threadIdx.x ; i < 256 ; i += blockDim.x)
            L.fast94[i] = tables->fast94[i];
        for (uint32_t i = threadIdx.x ; i < DCS_TRIE94_MAX ; i += blockDim.x)
            L.trie94[i] = tables->trie94[i];
        for (uint32_t i = threadIdx.x ; i < sizeof(L.multi94) / 4 ; i += blockDim.x)
            reinterpret_cast<uint32_t *>(L.multi94)[i] = reinterpret_cast<const uint32_t *>(tables->multi94)[i];
    }
    __syncthreads();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * kWaves + wave;
    if (k >= nStreams)
        return;
    const DcsStreamLoc loc = locs[k];
    const int os = loc.os;
    const uint8_t *stream = reinterpret_cast<const uint8_t *>(blobBase + loc.off);     // (blobBase 0: the locations are device addresses)
    const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stream) & 3);
    const uint32_t len = loc.len;

    Walk s;
    s.L = &L;
    s.lane = lane;
    s.os = os;
    s.b.sBase = (const uint32_t __attribute__((address_space(1))) *)(reinterpret_cast<uintptr_t>(stream) - skew);
    s.b.endByte = skew + len;
    s.b.nDwValid = (skew + len + 3) / 4;
    s.b.ring = L.ring[wave];
    s.b.lane = lane;
    s.pace = &g_idxPace[paceSlot & (kPaceSlots - 1)][(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u) * 16];        // hwreg(HW_REG_XCC_ID, 0, 4)
    uint32_t *rec = L.rec[wave];

    // container (InitChannelStream :1433-1463, InitStreamPlayback :1595-1641): lane l looks at byte l of the stream
    const uint32_t vByte = (lane < 18 && lane < len) ? stream[lane] : 0u;
    const uint32_t nFrames = (rl(vByte, 0) << 8) | rl(vByte, 1);
    const bool typeBit = (rl(vByte, 2) & 0x80u) != 0;
    const uint32_t hdrLen = (os == DCS_OS93A && typeBit) ? 1u : 16u;
    s.vHeader = (lane < hdrLen && lane + 2 < len) ? stream[lane + 2] : 0u;
    s.b.setPayload(skew + 2 + hdrLen);
    const uint32_t h0 = rl(s.vHeader, 0), h1 = rl(s.vHeader, 1), h2 = rl(s.vHeader, 2);
    if (os == DCS_OS93A && typeBit)
        s.nBands = static_cast<int>(h0 & 0x1F);
    else
    {
        const unsigned long long ends = __ballot(lane < 16 && (s.vHeader & 0x7Fu) == 0x7Fu);
        s.nBands = ends != 0 ? static_cast<int>(__builtin_ctzll(ends)) : 16;
    }
    int format;
    if (os == DCS_OS93A)
        format = typeBit ? DCS_FMT_93A_T1 : DCS_FMT_93_T0;
    else if (os == DCS_OS93B)
        format = typeBit ? DCS_FMT_93B_T1 : DCS_FMT_93_T0;
    else if (!typeBit)
        format = DCS_FMT_94_T0;
    else
        format = (((h1 | h2) & 0x80u) == 0) ? DCS_FMT_94_T1_S0 : DCS_FMT_94_T1_S3;
    s.format = format;
    s.type1 = typeBit;
    s.sub0 = ((h1 | h2) & 0x80u) == 0;
    {
        const uint32_t full = lane == 0 ? 7u : lane == 1 ? 8u : lane == 15 ? 32u : 16u;        // :1848-1850
        const bool strided = (s.vHeader & 0x40u) != 0;                                             // :1858-1862
        s.vCount = lane < 16 ? (strided ? full / 2 : full) : 0u;
        s.vInc = strided ? 2u : 1u;
    }
    if (format == DCS_FMT_93A_T1)
    {
        s.vBB = L.T.bandBits93a[((h0 & 0x60u) >> 1) + (lane & 15)];
        s.vSc1 = L.T.scaleCb93a[lane & 15];
        s.vInputs = L.T.inputs93a[lane < 24 ? lane : 0];
    }

    // (one array for the launch, or per stream)
    DcsFrameIndex *outRec;
    DcsFrameDigest *outDigest;
    DcsStreamInfo *outInfo;
    if (outs != nullptr)
    {
        const StreamOut o = outs[k];
        outRec = o.records; outDigest = o.digest; outInfo = o.info;
    }
    else
    {
        outRec = out + loc.firstRecord;
        outDigest = digest != nullptr ? digest + loc.firstRecord : nullptr;
        outInfo = infos + k;
    }
    // (the layout is the stream's: one frame loop per family, so that no frame pays for joining three walks' registers)
    uint32_t valid = 0, payloadBits = 0;
    switch (format)
    {
    case DCS_FMT_93_T0:
    case DCS_FMT_93B_T1: walkFrames<0>(s, nFrames, outRec, outDigest, valid, payloadBits); break;
    case DCS_FMT_93A_T1: walkFrames<1>(s, nFrames, outRec, outDigest, valid, payloadBits); break;
    default:             walkFrames<2>(s, nFrames, outRec, outDigest, valid, payloadBits); break;
    }

    // the stream's summary (GetStreamInfo :1486-1537)
    waveSync();
    if (lane < 16)
        reinterpret_cast<uint8_t *>(rec)[16 + lane] = static_cast<uint8_t>(s.vHeader);
    if (lane == 0)
    {
        rec[0] = nFrames;
        rec[1] = s.b.bytesFetched(2 + hdrLen);
        rec[2] = typeBit ? 1u : 0u;
        rec[3] = (os == DCS_OS94 || os == DCS_OS95) ? (((h1 & 0x80u) >> 6) | ((h1 & 0x80u) >> 7)) : 0u;     // sic (:1517)
        rec[8] = static_cast<uint32_t>(format);
        rec[9] = hdrLen;
        rec[10] = valid;
        rec[11] = payloadBits;
    }
    waveSync();
    if (lane < 12)
        reinterpret_cast<uint32_t *>(outInfo)[lane] = rec[lane];
#ifdef DCS_IDX_STAMPS
    if (lane == 0)
        for (int i = 0 ; i < 12 ; ++i)
            atomicAdd(&g_idxStamps[i], s.b.acc[i]);
#endif
}

}   // namespace dcsidx

// dcs_tables.cpp -- expands the canonical code lists of dcs_tables.h into the lookup structures the
// kernels (and the host index pass) use.  See dcs_common.h for the layouts.
#include <cstddef>
#include "dcs_common.h"
#include "dcs_tables.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

// Build fast[256] + trie[] for a prefix code.  payload(v) maps the list's value to the 8-bit payload.
template <size_t N, typename F>
void buildVlc(const DcsVlc (&codes)[N], uint16_t *fast, uint16_t *trie, size_t trieCap, F payload)
{
    // binary trie in a temporary pointer form
    struct Node { int child[2]; int leaf; };
    std::vector<Node> nodes;
    nodes.push_back({{-1, -1}, -1});
    for (const DcsVlc &c : codes)
    {
        int n = 0;
        for (int b = c.len - 1 ; b >= 0 ; --b)
        {
            int bit = (c.code >> b) & 1;
            if (nodes[n].child[bit] < 0)
            {
                nodes[n].child[bit] = static_cast<int>(nodes.size());
                nodes.push_back({{-1, -1}, -1});
            }
            n = nodes[n].child[bit];
        }
        nodes[n].leaf = payload(c.val) & 0xFF;
    }

    // flatten: children of an interior node are stored adjacently; the root's children sit at 0,1
    std::vector<uint16_t> flat;
    std::vector<int> flatIndexOf(nodes.size(), -1);
    struct Pending { int node; size_t slot; };
    std::vector<Pending> work;
    flat.resize(2);
    work.push_back({nodes[0].child[0], 0});
    work.push_back({nodes[0].child[1], 1});
    for (size_t w = 0 ; w < work.size() ; ++w)
    {
        const Node &nd = nodes[work[w].node];
        flatIndexOf[work[w].node] = static_cast<int>(work[w].slot);
        if (nd.leaf >= 0)
            flat[work[w].slot] = static_cast<uint16_t>(0x8000 | nd.leaf);
        else
        {
            size_t base = flat.size();
            flat.resize(base + 2);
            flat[work[w].slot] = static_cast<uint16_t>(base);
            work.push_back({nd.child[0], base});
            work.push_back({nd.child[1], base + 1});
        }
    }
    memset(trie, 0, trieCap * sizeof(uint16_t));
    for (size_t i = 0 ; i < flat.size() && i < trieCap ; ++i)
        trie[i] = flat[i];

    // first-level table on 8 bits
    for (int v = 0 ; v < 256 ; ++v)
    {
        int n = 0, used = 0;
        while (used < 8 && nodes[n].leaf < 0)
        {
            n = nodes[n].child[(v >> (7 - used)) & 1];
            ++used;
        }
        if (nodes[n].leaf >= 0)
            fast[v] = static_cast<uint16_t>(0x8000 | (used << 8) | nodes[n].leaf);
        else
        {
            // interior after 8 bits: continue from this node's child pair
            uint16_t slotVal = flat[static_cast<size_t>(flatIndexOf[n])];
            fast[v] = slotVal;      // = index of its '0' child (bit15 clear)
        }
    }
}

template <size_t N>
void expandSampleBook(const DcsVlc (&codes)[N], int maxBits, int ref, uint16_t *out)
{
    // every slot consumes at least one sample, so that a kernel loop over a band always terminates
    for (int i = 0 ; i < (1 << maxBits) ; ++i)
        out[i] = static_cast<uint16_t>(1 << 13);
    for (const DcsVlc &c : codes)
    {
        const bool twoZeros = (c.val & 0x80) != 0;
        const int sample = twoZeros ? 0 : c.val - ref;
        const uint16_t e = static_cast<uint16_t>((sample & 0xFF) | (c.len << 8) | ((twoZeros ? 2 : 1) << 13));
        int span = 1 << (maxBits - c.len);
        for (int i = 0 ; i < span ; ++i)
            out[(c.code << (maxBits - c.len)) + i] = e;
    }
}

DcsDevTables build()
{
    DcsDevTables t;
    memset(&t, 0, sizeof(t));
    static const int maxBits[7] = { 0, 2, 3, 5, 7, 8, 9 };      // DCSDecoderNative.cpp:2005
    int base = 0;
    uint16_t *cb = t.lds.cb94;
    auto put = [&](int k, auto &codes) {
        expandSampleBook(codes, maxBits[k], 1 << (k - 1), cb + base);
        t.lds.cbInfo[k] = static_cast<uint16_t>((base << 4) | maxBits[k]);
        base += 1 << maxBits[k];
    };
    put(1, kVlc94Sample1); put(2, kVlc94Sample2); put(3, kVlc94Sample3);
    put(4, kVlc94Sample4); put(5, kVlc94Sample5); put(6, kVlc94Sample6);
    // several codes per look for the device index pass: a code is taken when its bits are all there (a prefix code is known by
    // its own bits: the look-up with the missing look-ahead bits as zeros finds it)
    for (int k = 1 ; k <= 6 ; ++k)
    {
        const uint16_t *book = cb + (t.lds.cbInfo[k] >> 4);
        for (uint32_t x = 0 ; x < (1u << DCS_IDX_MULTI_BITS) ; ++x)
        {
            uint32_t at = 0, samples = 0;
            for (;;)
            {
                const uint32_t idx = ((x << at) & ((1u << DCS_IDX_MULTI_BITS) - 1)) >> (DCS_IDX_MULTI_BITS - maxBits[k]);
                const uint32_t e = book[idx], len = (e >> 8) & 0x1F, step = (e >> 13) == 2 ? 2 : 1;
                if (len == 0 || at + len > DCS_IDX_MULTI_BITS || samples + step > DCS_IDX_MULTI_SAMPLES)
                    break;
                at += len;
                samples += step;
            }
            if (at == 0 || samples == 0)
                abort();                        // (every book's longest code is shorter than the look)
            t.multi94[k - 1][x] = static_cast<uint8_t>(at | (samples << 4));
        }
    }
    for (int w = 7 ; w <= 16 ; ++w)
        t.lds.raw94[2 * (w - 7)] = t.lds.raw94[2 * (w - 7) + 1] = static_cast<uint16_t>((w << 8) | (1 << 13));

    buildVlc(kVlc94BandTypeDelta, t.fast94, t.trie94, DCS_TRIE94_MAX, [](int v) { return v + 16; });
    buildVlc(kVlc93BandType, t.lds.fast93, t.lds.trie93, DCS_TRIE93_MAX, [](int v) { return v; });

    for (int i = 0 ; i < 16 ; ++i)
    {
        t.lds.xlat94[i] = kXlatB02[i];
        t.lds.xlat94[16 + i] = kXlatB35[i];
        t.lds.xlat94[32 + i] = kXlatB6F[i];
        t.lds.preAdj94[i] = kPreAdjSub0[i];
        t.lds.preAdj94[16 + i] = kPreAdjSub3[i];
    }
    // band94: the set-up of unpack94 (dcs_kernels.hip.h) as a table; the expressions are the ones the index pass walks
    // (dcs_scan.h: dcsScan94), evaluated for every band class and band-type code
    for (int k = 0 ; k < 72 ; ++k)
    {
        const bool type1 = k < DCS_B94_TYPE0;
        const int cls = type1 ? k / 17 : 0, code0 = type1 ? k % 17 : k - DCS_B94_TYPE0;
        if (k > DCS_B94_TYPE0 + 17)
            continue;                                           // padding
        const uint32_t x = t.lds.xlat94[cls * 16 + (code0 & 15)];
        const bool fatal1 = type1 && code0 > 15;
        const int code = (type1 && !fatal1) ? static_cast<int>(x & 0xFF) : code0;
        const uint32_t adj = type1 ? x >> 8 : 0;
        const bool zero = code0 == 0;
        const bool fatal = !zero && (fatal1 || code > 16);
        const bool stop = !zero && !fatal && code == 0;
        const bool raw = code > 6;
        const uint32_t info = t.lds.cbInfo[code < 7 ? code : 7];
        const uint32_t shPeek = 32u - (raw ? static_cast<uint32_t>(code < 16 ? code : 16) : (info & 0xF));
        const uint32_t shIdx = raw ? 31u : shPeek;
        const size_t book = raw ? offsetof(DcsLdsTables, raw94) / 2 + 2 * static_cast<size_t>((code < 16 ? code : 16) - 7)
                                : offsetof(DcsLdsTables, cb94) / 2 + (info >> 4);
        // (the kernel's set-up leaves the STOP of :1985-1991 out: no band-type code but 0 translates to sample code 0)
        if (book > 0x7FF || adj > 0x7F || stop)
            abort();
        t.lds.band94[k] = static_cast<uint32_t>(book) | ((shPeek & 31u) << 11) | ((shIdx & 31u) << 16) | (raw ? DCS_B94_RAW : 0u)
                        | (zero ? DCS_B94_ZERO : 0u) | (stop ? DCS_B94_STOP : 0u) | (fatal ? DCS_B94_FATAL : 0u) | (adj << 25);
    }
    memcpy(t.lds.bandBits93a, kBandBits93a, sizeof(t.lds.bandBits93a));
    memcpy(t.lds.scaleCb93a, kScaleCb93a, sizeof(t.lds.scaleCb93a));
    for (int i = 0 ; i < 18 ; ++i)
        t.lds.inputs93a[i] = kInputsPerBand93a[i];
    memcpy(t.lds.scaleMant, kScaleMant, sizeof(t.lds.scaleMant));
    for (int c = 0 ; c < 64 ; ++c)
        t.lds.scale64[c] = static_cast<uint16_t>(kScaleMant[c & 3] >> (15 - ((c >> 2) & 15)));
    // (unpack93a in the kernel derives its own constants from the second mantissa)
    if (kScaleMant[0] != 0x8000 || kScaleMant[1] != 0x9838 || kScaleMant[2] != 0xB505 || kScaleMant[3] != 0xD745)
        abort();
    memcpy(t.pair93a, kPair93a, sizeof(t.pair93a));
    memcpy(t.fftCoef, kFftCoef, sizeof(t.fftCoef));
    // the kernel's multiplier-free butterflies (bflyIdx) rely on these exact twiddles: (-1.0, 0) and (0, -1.0)
    if (kFftCoef[0x80] != 0x8000 || kFftCoef[0] != 0x0000 || kFftCoef[0x81] != 0x0000 || kFftCoef[1] != 0x8000)
        abort();
    memcpy(t.ovlCoef, kOverlapCoef, sizeof(t.ovlCoef));
    // (the kernel carries these as constants: kTwADoubled in dcs_kernels.hip.h)
    static const int kTwAInKernel[8][2] = { { -65536, 0 }, { 0, -65536 }, { -46340, -46340 }, { 46340, -46340 },
                                            { -60548, -25080 }, { 25080, -60548 }, { -25080, -60548 }, { 60548, -25080 } };
    for (int k = 0 ; k < 8 ; ++k)
    {
        const int c = static_cast<int16_t>(kFftCoef[0x80 + k]), sn = static_cast<int16_t>(kFftCoef[k]);
        t.twA[k][0] = 2 * c; t.twA[k][1] = 2 * sn; t.twA[k][2] = -2 * sn; t.twA[k][3] = 0;
        if (kTwAInKernel[k][0] != 2 * c || kTwAInKernel[k][1] != 2 * sn)
            abort();
    }

    // per-lane transform constants (see DcsLaneConsts)
    auto rev = [](int v, int bits) { int r = 0; for (int i = 0 ; i < bits ; ++i) r |= ((v >> i) & 1) << (bits - 1 - i); return r; };
    auto tw = [&](int part) { return static_cast<uint32_t>(kFftCoef[0x80 + part]) | (static_cast<uint32_t>(kFftCoef[part]) << 16); };
    for (int lane = 0 ; lane < 64 ; ++lane)
    {
        uint32_t *k94 = t.lane94[lane], *k93 = t.lane93[lane];
        const int l8 = lane & 7, l16 = lane & 15;
        for (int j = 0 ; j < 8 ; ++j)
        {
            const int i = l8 + 8 * j;
            k94[DCS_K94_PRE + j] = static_cast<uint32_t>(kFftCoef[rev(2 + 4 * i, 9)]) | (static_cast<uint32_t>(kFftCoef[rev(4 * i, 9)]) << 16);
        }
        for (int k = 0 ; k < 2 ; ++k) k94[DCS_K94_TWB + k] = tw(2 * l8 + k);
        for (int k = 0 ; k < 4 ; ++k) k94[DCS_K94_TWB + 2 + k] = tw(4 * l8 + k);
        for (int k = 0 ; k < 8 ; ++k) k94[DCS_K94_TWB + 6 + k] = tw(8 * l8 + k);
        k93[DCS_K93_TWB] = tw(l16);
        for (int k = 0 ; k < 2 ; ++k) k93[DCS_K93_TWB + 1 + k] = tw(2 * l16 + k);
        for (int k = 0 ; k < 4 ; ++k) k93[DCS_K93_TWB + 3 + k] = tw(4 * l16 + k);
        for (int k = 0 ; k < 8 ; ++k) k93[DCS_K93_TWB + 7 + k] = tw(8 * l16 + k);
        const int m = rev(l8, 3);
        k94[DCS_K94_OVLA] = static_cast<uint32_t>(kOverlapCoef[2 * m]) | (static_cast<uint32_t>(kOverlapCoef[2 * m + 1]) << 16);
        k94[DCS_K94_OVLB] = static_cast<uint32_t>(kOverlapCoef[15 - 2 * m]) | (static_cast<uint32_t>(kOverlapCoef[14 - 2 * m]) << 16);
        const int i = rev(l16, 4);
        k93[DCS_K93_OVL] = static_cast<uint32_t>(kOverlapCoef[i]) | (static_cast<uint32_t>(kOverlapCoef[15 - i]) << 16);
    }
    return t;
}

}   // namespace

const DcsDevTables &dcsTables()
{
    static const DcsDevTables tables = build();
    return tables;
}

static_assert(sizeof(DcsLdsTables) % 16 == 0, "LDS table block must be a multiple of 16 bytes");
static_assert(DCS_LDS_DECODE_BYTES % 16 == 0 && DCS_LDS_DECODE_BYTES == offsetof(DcsLdsTables, cbInfo), "the decode kernel stages the block up to the index walk's tables");
static_assert(offsetof(DcsDevTables, lane94) % 16 == 0 && offsetof(DcsDevTables, lane93) % 16 == 0 && offsetof(DcsDevTables, twA) % 16 == 0, "lane constants and twiddles are fetched as uint4");
static_assert(sizeof(DcsSrcDesc) == 160, "DcsSrcDesc layout");
static_assert(offsetof(DcsSrcDesc, idx) == 12, "DcsSrcDesc layout");
static_assert(sizeof(DcsFrameJob) == 16, "DcsFrameJob layout");
static_assert(sizeof(DcsFrameIndex) == 148 && offsetof(DcsFrameIndex, split) == 28, "DcsFrameIndex layout");
static_assert(sizeof(DcsSlot) == 32, "DcsSlot layout");
static_assert(offsetof(DcsDevTables, pair93a) % 4 == 0, "OS93a sample pairs are read as 32-bit words");

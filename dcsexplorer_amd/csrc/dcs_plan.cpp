// dcs_plan.cpp -- cuts a job list into per-wavefront chunks.
//
// The only coupling between output frames is the 16-sample overlap tail a frame hands to its
// successor (DCSDecoderNative.cpp:538-575, :789-812).  Inside a chunk the tail travels through LDS;
// when a frame's predecessor falls in another chunk it is decoded a second time here as a HALO slot
// (tail only, no PCM written), which costs 1/fpw extra work and needs no inter-workgroup
// communication, no second kernel and no extra HBM traffic.
//
// Jobs are visited chain by chain (a chain = frames linked through `prev`, i.e. one decoder playing), not
// in job order: a batch whose job list interleaves many streams frame by frame would otherwise need a
// halo for every single frame.  Where a frame lands in the plan does not matter, its PCM goes to its job
// index.
#include "dcs_common.h"
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>

// What the planner needs to know about a source, from either form of source record: the full descriptor of the ABI, or
// the digest the pipeline works from when the index records stay on the device (DcsPlanSrc).
static inline uint64_t srcStreamOff(const DcsSrcDesc &s) { return s.streamOff; }
static inline uint32_t srcHdrLen(const DcsSrcDesc &s) { return s.hdrLen; }
static inline uint32_t srcBitOff(const DcsSrcDesc &s) { return s.idx.bitOff; }
static inline uint32_t srcNBits(const DcsSrcDesc &s) { return s.idx.nBits; }
static inline int srcNBands(const DcsSrcDesc &s) { return s.idx.nBands; }
static inline uint32_t srcFlags(const DcsSrcDesc &s) { return s.idx.flags; }
static inline uint64_t srcStreamOff(const DcsPlanSrc &s) { return s.streamOff; }
static inline uint32_t srcHdrLen(const DcsPlanSrc &s) { return s.hdrLen; }
static inline uint32_t srcBitOff(const DcsPlanSrc &s) { return s.bitOff; }
static inline uint32_t srcNBits(const DcsPlanSrc &s) { return s.nBits; }
static inline int srcNBands(const DcsPlanSrc &s) { return s.nBands; }
static inline uint32_t srcFlags(const DcsPlanSrc &s) { return s.flags; }

// the slot of one job; where its first source's bytes go in the pool is filled in by placeFrame
template <class Src>
static DcsSlot makeSlot(const DcsFrameJob &jb, uint32_t job, uint8_t prevSlot, uint8_t flags, const Src *srcs, int fpw)
{
    DcsSlot sl{ job, prevSlot, flags, jb.nSrc, static_cast<uint8_t>(jb.volShift | (jb.xform << 4)), jb.firstSrc, jb.prev, 0, 0, 0, 0, 0, 0, 0 };
    if (srcs != nullptr && jb.nSrc != 0)
    {
        const Src &sd = srcs[jb.firstSrc];
        const int sub = 64 / fpw;
        const int nb16 = srcNBands(sd) < 16 ? srcNBands(sd) : 16;
        const int bpl = (nb16 + sub - 1) / sub;
        sl.bpl = (srcFlags(sd) & DCS_IDX_SERIAL) ? 0 : static_cast<uint8_t>(bpl < 1 ? 1 : bpl);
    }
    return sl;
}

template <class Src>
static uint32_t planChunks(const DcsFrameJob *jobs, uint32_t nJobs, const Src *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                           int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t imgCap)
{
    // (diagnostic: fewer frames per chunk than the kernel variant has slots, the rest of the wavefront idles)
    const uint32_t limit = static_cast<uint32_t>(framesPerChunk >= 1 && framesPerChunk < fpw ? framesPerChunk : fpw);
    slots.clear();
    if (nJobs == 0 || fpw < 2)
        return 0;
    slots.reserve(static_cast<size_t>(nJobs) + nJobs / static_cast<uint32_t>(fpw - 1) + static_cast<size_t>(fpw));

    // slotOf[j] = position of job j inside the chunk being built (valid when stampOf[j] == chunk id)
    std::vector<uint32_t> stampOf(nJobs, 0xFFFFFFFFu);
    std::vector<uint8_t> slotOf(nJobs, 0);
    // tail hand-off between chunks: where a job's own (non-halo) slot went, and the last live slot of every closed chunk
    std::vector<uint32_t> homeChunk(nJobs, 0xFFFFFFFFu);
    std::vector<size_t> homePos(nJobs, 0);
    std::vector<size_t> lastLive;

    // LDS bit-pool budget: per unpack round r (the r-th source of every job in the chunk) the staged
    // frames must fit fpw * DCS_POOL_DW_PER_FRAME dwords
    const uint32_t poolCap = dcsPoolCapacity(fpw);
    uint32_t poolUse[DCS_MAX_CHANNELS] = { 0 };
    auto poolNeed = [&](uint32_t j, uint32_t r) -> uint32_t {
        if (srcs == nullptr || r >= jobs[j].nSrc)
            return 0;
        const Src &sd = srcs[jobs[j].firstSrc + r];
        // (rounded up to the 16-byte granule of the run staging)
        return (dcsPoolDwords(srcStreamOff(sd), srcHdrLen(sd), srcBitOff(sd), srcNBits(sd)) + 3) & ~3u;
    };
    auto poolFits = [&](uint32_t j, uint32_t halo, bool withHalo) {
        uint32_t rounds = jobs[j].nSrc;
        if (withHalo && jobs[halo].nSrc > rounds)
            rounds = jobs[halo].nSrc;
        for (uint32_t r = 0 ; r < rounds && r < DCS_MAX_CHANNELS ; ++r)
            if (poolUse[r] + poolNeed(j, r) + (withHalo ? poolNeed(halo, r) : 0) > poolCap)
                return false;
        return true;
    };
    auto poolAdd = [&](uint32_t j) {
        for (uint32_t r = 0 ; r < jobs[j].nSrc && r < DCS_MAX_CHANNELS ; ++r)
            poolUse[r] += poolNeed(j, r);
    };

    uint32_t chunk = 0;
    uint32_t used = 0;                  // slots filled in the current chunk
    // unpack round 0 is staged as runs of blob dwords (DcsSlot): a frame that starts inside or right behind the
    // chunk's last run extends it, anything else opens a new run at the next 16-byte boundary of the pool
    struct Run { uint32_t start, n, poolOff; };
    std::vector<Run> runs;
    uint32_t runUse = 0;                // pool dwords the runs take (<= poolUse[0], which counts every frame in full)
    auto placeFrame = [&](DcsSlot &sl, const DcsFrameJob &jb) {
        if (srcs == nullptr || jb.nSrc == 0)
            return;
        const Src &sd = srcs[jb.firstSrc];
        const uint64_t bitPos = (srcStreamOff(sd) + 2 + srcHdrLen(sd)) * 8 + srcBitOff(sd);
        const uint32_t st = static_cast<uint32_t>(bitPos >> 5);
        const uint32_t n = dcsPoolDwords(srcStreamOff(sd), srcHdrLen(sd), srcBitOff(sd), srcNBits(sd));
        if (!runs.empty() && st >= runs.back().start && st <= runs.back().start + runs.back().n)
        {
            Run &r = runs.back();
            if (st + n > r.start + r.n)
                r.n = st + n - r.start;
            runUse = r.poolOff + ((r.n + 3) & ~3u);
        }
        else
        {
            runs.push_back(Run{ st, n, runUse });
            runUse += (n + 3) & ~3u;
        }
        sl.poolOff = static_cast<uint16_t>(runs.back().poolOff + (st - runs.back().start));
    };
    // pool dwords the chunk's runs would take with job jb's first source added (what placeFrame will do)
    auto runUseWith = [&](const DcsFrameJob &jb, uint32_t from) -> uint32_t {
        if (srcs == nullptr || jb.nSrc == 0)
            return from;
        const Src &sd = srcs[jb.firstSrc];
        const uint64_t bitPos = (srcStreamOff(sd) + 2 + srcHdrLen(sd)) * 8 + srcBitOff(sd);
        const uint32_t st = static_cast<uint32_t>(bitPos >> 5);
        const uint32_t n = dcsPoolDwords(srcStreamOff(sd), srcHdrLen(sd), srcBitOff(sd), srcNBits(sd));
        if (!runs.empty() && from == runUse && st >= runs.back().start && st <= runs.back().start + runs.back().n)
        {
            const Run &r = runs.back();
            const uint32_t len = st + n > r.start + r.n ? st + n - r.start : r.n;
            return r.poolOff + ((len + 3) & ~3u);
        }
        return from + ((n + 3) & ~3u);
    };
    const DcsSlot empty{ 0xFFFFFFFFu, DCS_NO_PREV_SLOT, DCS_SLOT_EMPTY, 0, 0, 0, DCS_PREV_NONE, 0, 0, 0, 0, 0, 0, 0 };
    auto closeChunk = [&]() {
        lastLive.push_back(slots.size() - 1);
        while (used < static_cast<uint32_t>(fpw)) { slots.push_back(empty); ++used; }
        // run k rides in slot k of the chunk (there are never more runs than frames)
        DcsSlot *cs = &slots[slots.size() - static_cast<size_t>(fpw)];
        for (size_t k = 0 ; k < runs.size() && k < static_cast<size_t>(fpw) ; ++k)
        {
            cs[k].runStartDw = runs[k].start;
            cs[k].runNDw = static_cast<uint16_t>(runs[k].n);
            cs[k].runPoolOff = static_cast<uint16_t>(runs[k].poolOff);
        }
        runs.clear();
        runUse = 0;
        ++chunk;
        used = 0;
        for (uint32_t &u : poolUse) u = 0;
    };
    auto inChunk = [&](uint32_t j) { return j < nJobs && stampOf[j] == chunk; };

    // chain order: a job is followed by its first successor; everything else starts a new chain
    auto linked = [&](uint32_t j) {
        const uint32_t prev = jobs[j].prev;
        return prev != DCS_PREV_NONE && (prev & DCS_PREV_EXT) == 0 && prev < nJobs && prev != j;
    };
    std::vector<uint32_t> succ(nJobs, 0xFFFFFFFFu);
    std::vector<uint8_t> follows(nJobs, 0);
    for (uint32_t j = 0 ; j < nJobs ; ++j)
        if (linked(j) && succ[jobs[j].prev] == 0xFFFFFFFFu)
        {
            succ[jobs[j].prev] = j;
            follows[j] = 1;
        }
    std::vector<uint32_t> order;
    order.reserve(nJobs);
    for (uint32_t head = 0 ; head < nJobs ; ++head)
    {
        if (follows[head])
            continue;
        for (uint32_t j = head ; j != 0xFFFFFFFFu ; j = succ[j])
            order.push_back(j);
    }
    // (a cycle of prev links has no head: append whatever was not reached, in job order)
    if (order.size() != nJobs)
    {
        std::vector<uint8_t> seen(nJobs, 0);
        for (uint32_t j : order) seen[j] = 1;
        for (uint32_t j = 0 ; j < nJobs ; ++j)
            if (!seen[j]) order.push_back(j);
    }

    for (uint32_t j : order)
    {
        const uint32_t prev = jobs[j].prev;
        const bool ext = prev != DCS_PREV_NONE && (prev & DCS_PREV_EXT) != 0;
        const bool link = prev != DCS_PREV_NONE && !ext && prev < nJobs && prev != j;

        // The predecessor's tail reaches this frame (a) through LDS when both are in one chunk, (b) through the
        // hand-off buffer when the predecessor is the LAST frame of an EARLIER chunk (the usual case along a chain:
        // that wavefront publishes the tail, this one picks it up after its own transform; chunks are dispatched in
        // index order, so the producer is never behind the consumer), (c) otherwise by decoding the predecessor a
        // second time in this chunk as a halo slot.
        auto canImport = [&](uint32_t p) {
            return handoff && homeChunk[p] != 0xFFFFFFFFu && homeChunk[p] < chunk && lastLive[homeChunk[p]] == homePos[p];
        };
        uint32_t need = (link && !inChunk(prev) && !canImport(prev)) ? 2u : 1u;
        // imgCap (resident batches, dcsPlanChunksCapped): a chunk whose runs would outgrow the batch's pool image is closed early
        // as well -- a few per cent of the chunks then hold a frame less, and every package is that much shorter
        const bool imgFull = imgCap != 0 && used != 0
                             && runUseWith(jobs[j], need == 2 ? runUseWith(jobs[prev], runUse) : runUse) > imgCap;
        if (used + need > (need == 2 && limit < 2 ? 2u : limit) || (used != 0 && !poolFits(j, prev, need == 2)) || imgFull)
        {
            closeChunk();
            need = (link && !canImport(prev)) ? 2u : 1u;      // nothing of the new chunk exists yet
        }

        uint8_t prevSlot = DCS_NO_PREV_SLOT;
        uint8_t flags = static_cast<uint8_t>(ext ? DCS_SLOT_EXT_TAIL : 0);
        // whose tail goes to tailsOut: the last frame of a chain (nothing in the batch follows it), or every frame on request
        if (keepAllTails || succ[j] == 0xFFFFFFFFu)
            flags |= DCS_SLOT_KEEP_TAIL;
        uint32_t importFrom = 0;
        if (link)
        {
            if (inChunk(prev))
                prevSlot = slotOf[prev];
            else if (canImport(prev))
            {
                slots[homePos[prev]].flags |= DCS_SLOT_EXPORT;
                slots[homePos[prev]].nextJob = j;           // (whoever arrives second at the rendezvous finishes job j's first samples)
                flags |= DCS_SLOT_IMPORT;
                importFrom = homeChunk[prev];
            }
            else
            {
                slots.push_back(makeSlot(jobs[prev], prev, DCS_NO_PREV_SLOT, DCS_SLOT_HALO, srcs, fpw));
                placeFrame(slots.back(), jobs[prev]);
                poolAdd(prev);
                stampOf[prev] = chunk;
                slotOf[prev] = static_cast<uint8_t>(used++);
                prevSlot = slotOf[prev];
            }
        }
        slots.push_back(makeSlot(jobs[j], j, prevSlot, flags, srcs, fpw));
        if (flags & DCS_SLOT_IMPORT)
            slots.back().prevJob = importFrom;          // the chunk whose last frame publishes the tail
        placeFrame(slots.back(), jobs[j]);
        poolAdd(j);
        stampOf[j] = chunk;
        slotOf[j] = static_cast<uint8_t>(used++);
        homeChunk[j] = chunk;
        homePos[j] = slots.size() - 1;
        if (used >= limit)
            closeChunk();
    }
    if (used != 0)
        closeChunk();

    // The chunks stay in chain order.  (Rounds 2-5 reordered them by depth in the hand-off graph, so that a consumer did not reach
    // its wait before the tail was there; consumers no longer wait -- the rendezvous, dcs_kernels.hip.h -- and the order measures
    // the same either way: 33.15 us for 65 536 frames, tools/ab_order.sh.)
    (void)depthOrder;
    return chunk;
}

// test hook: the chunks of a plan in a seeded random order (a chunk that takes a tail names its predecessor's chunk: renumbered)
void dcsShuffleChunks(std::vector<DcsSlot> &slots, uint32_t nChunks, int fpw, uint32_t seed)
{
    if (seed == 0 || nChunks < 2)
        return;
    const size_t F = static_cast<size_t>(fpw);
    std::vector<uint32_t> newIndex(nChunks);
    for (uint32_t c = 0 ; c < nChunks ; ++c)
        newIndex[c] = c;
    uint64_t x = seed;
    for (uint32_t c = nChunks - 1 ; c > 0 ; --c)
    {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        std::swap(newIndex[c], newIndex[static_cast<uint32_t>((x >> 33) % (c + 1))]);
    }
    std::vector<DcsSlot> moved(slots.size());
    for (uint32_t c = 0 ; c < nChunks ; ++c)
        for (size_t k = 0 ; k < F ; ++k)
        {
            DcsSlot sl = slots[c * F + k];
            if (!(sl.flags & DCS_SLOT_EMPTY) && (sl.flags & DCS_SLOT_IMPORT))
                sl.prevJob = newIndex[sl.prevJob];
            moved[static_cast<size_t>(newIndex[c]) * F + k] = sl;
        }
    slots.swap(moved);
}

uint32_t dcsPlanChunks(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                       int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t shuffleSeed)
{
    const uint32_t n = planChunks(jobs, nJobs, srcs, fpw, slots, handoff, framesPerChunk, depthOrder, keepAllTails, 0);
    dcsShuffleChunks(slots, n, fpw, shuffleSeed);
    return n;
}

// A resident batch is planned for the shortest packages that cost it next to nothing: the plan above, then -- when its fullest
// chunks are outliers -- once more with the pool image capped at what 97 % of the chunks need (rounded up to 32 dwords).  The
// chunks that would have been fuller close a frame early.  *imgDwOut = dcsImageDwords of the plan returned.
template <class Src>
static uint32_t planChunksCapped(const DcsFrameJob *jobs, uint32_t nJobs, const Src *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                                 int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t *imgDwOut, uint32_t places)
{
    uint32_t nChunks = planChunks(jobs, nJobs, srcs, fpw, slots, handoff, framesPerChunk, depthOrder, keepAllTails, 0);
    uint32_t imgDw = dcsImageDwords(slots.data(), nChunks, fpw);
    if (nChunks >= 64 && srcs != nullptr)
    {
        std::vector<uint32_t> hist(dcsPoolCapacity(fpw) / 32 + 2, 0);
        for (uint32_t c = 0 ; c < nChunks ; ++c)
        {
            uint32_t use = 0;
            for (int k = 0 ; k < fpw ; ++k)
            {
                const DcsSlot &sl = slots[static_cast<size_t>(c) * static_cast<size_t>(fpw) + static_cast<size_t>(k)];
                if (sl.runNDw != 0 && static_cast<uint32_t>(sl.runPoolOff) + sl.runNDw > use)
                    use = static_cast<uint32_t>(sl.runPoolOff) + sl.runNDw;
            }
            ++hist[(use + 31) / 32 < hist.size() ? (use + 31) / 32 : hist.size() - 1];
        }
        uint32_t seen = 0, cap = imgDw;
        for (size_t b = 0 ; b < hist.size() ; ++b)
        {
            seen += hist[b];
            if (static_cast<uint64_t>(seen) * 100 >= static_cast<uint64_t>(nChunks) * 97)
            {
                cap = static_cast<uint32_t>(b) * 32;
                break;
            }
        }
        // (the cap is a target, not a limit: an empty chunk takes its first frame -- and a halo with its successor -- whatever
        // their size, and the image is sized by what the plan really holds, dcsImageDwords)
        if (cap >= 32 && cap + 32 <= imgDw)
        {
            std::vector<DcsSlot> again;
            const uint32_t n2 = planChunks(jobs, nJobs, srcs, fpw, again, handoff, framesPerChunk, depthOrder, keepAllTails, cap);
            const uint32_t img2 = dcsImageDwords(again.data(), n2, fpw);
            // Taken when the packages really get shorter in total -- and the launch no longer: `places` wavefronts run at a time
            // (0: unknown), a launch lasts as many generations of them as it has chunks, and the chunks closed early must not
            // open another one (measured, round 5: 65 536 frames = 8 192 chunks = exactly two generations, 33.4 us; the same
            // frames in 8 216 chunks 36.2)
            const bool sameGenerations = places == 0 || (n2 + places - 1) / places == (nChunks + places - 1) / places;
            if (sameGenerations && static_cast<uint64_t>(n2) * dcsPkgStride(fpw, img2) < static_cast<uint64_t>(nChunks) * dcsPkgStride(fpw, imgDw))
            {
                slots.swap(again);
                nChunks = n2;
                imgDw = img2;
            }
        }
    }
    if (imgDwOut != nullptr)
        *imgDwOut = imgDw;
    return nChunks;
}
uint32_t dcsPlanChunksCapped(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                             int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t *imgDwOut, uint32_t places)
{
    return planChunksCapped(jobs, nJobs, srcs, fpw, slots, handoff, framesPerChunk, depthOrder, keepAllTails, imgDwOut, places);
}
uint32_t dcsPlanChunksCappedLite(const DcsFrameJob *jobs, uint32_t nJobs, const DcsPlanSrc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                                 int framesPerChunk, bool depthOrder, bool keepAllTails, uint32_t *imgDwOut, uint32_t places)
{
    return planChunksCapped(jobs, nJobs, srcs, fpw, slots, handoff, framesPerChunk, depthOrder, keepAllTails, imgDwOut, places);
}

uint32_t dcsPlanChunksLite(const DcsFrameJob *jobs, uint32_t nJobs, const DcsPlanSrc *srcs, int fpw, std::vector<DcsSlot> &slots, bool handoff,
                           int framesPerChunk, bool depthOrder, bool keepAllTails)
{
    return planChunks(jobs, nJobs, srcs, fpw, slots, handoff, framesPerChunk, depthOrder, keepAllTails, 0);
}

// every source the jobs draw on is a 1994+ frame (the packages then carry 4-byte split records, dcs_common.h)
bool dcsAllSources94(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs)
{
    if (srcs == nullptr)
        return false;
    for (uint32_t j = 0 ; j < nJobs ; ++j)
        for (uint32_t r = 0 ; r < jobs[j].nSrc ; ++r)
            if (srcs[jobs[j].firstSrc + r].format < DCS_FMT_94_T0)
                return false;
    return true;
}

uint32_t dcsImageDwords(const DcsSlot *slots, uint32_t nChunks, int fpw)
{
    uint32_t use = 0;
    for (size_t i = 0, n = static_cast<size_t>(nChunks) * static_cast<size_t>(fpw) ; i < n ; ++i)
        if (slots[i].runNDw != 0)
        {
            const uint32_t end = static_cast<uint32_t>(slots[i].runPoolOff) + slots[i].runNDw;
            if (end > use)
                use = end;
        }
    const uint32_t cap = dcsPoolCapacity(fpw);
    use = (use + 31u) & ~31u;
    if (use < 32u) use = 32u;           // (the kernel's image loads clamp to imgDw - 4)
    return use < cap ? use : cap;
}

extern "C" DcsStatus dcs_plan_chunks2(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw, int handoff,
                                      uint64_t *slotsOut, size_t cap, uint32_t *nChunksOut)
{
    if (jobs == nullptr || nChunksOut == nullptr || fpw < 4 || fpw > 64)
        return DCS_ERR_INVALID_ARG;
    std::vector<DcsSlot> slots;
    // (the plan of a RESIDENT batch, dcs_batch_create: planned for the shortest packages)
    *nChunksOut = dcsPlanChunksCapped(jobs, nJobs, srcs, fpw, slots, handoff != 0, 0, true, false, nullptr, DCS_MI355X_WAVE_PLACES);
    if (slotsOut != nullptr)
    {
        if (cap < slots.size())
            return DCS_ERR_CAPACITY;
        for (size_t i = 0 ; i < slots.size() ; ++i)
            slotsOut[i] = static_cast<uint64_t>(slots[i].job) | (static_cast<uint64_t>(slots[i].prevSlot) << 32)
                        | (static_cast<uint64_t>(slots[i].flags) << 40);
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_plan_chunks(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, int fpw,
                                     uint64_t *slotsOut, size_t cap, uint32_t *nChunksOut)
{
    return dcs_plan_chunks2(jobs, nJobs, srcs, fpw, 1, slotsOut, cap, nChunksOut);
}

// ---------------------------------------------------------------------------------------------------------
// Chunk packages (dcs_common.h): everything unpack round 0 of a chunk needs, gathered next to each other on the
// host while the batch is being prepared -- slots, descriptor heads, stream headers (aligned), the split record of
// every lane, and the compressed bytes of the chunk's runs as an image of the LDS bit pool, dwords in bit order
// (big-endian).  A layout change only; nothing is decoded.  The device then reads nothing else in round 0.
// ---------------------------------------------------------------------------------------------------------
static void packChunks(const DcsSlot *slots, uint32_t c0, uint32_t c1, int fpw, const DcsSrcDesc *srcs,
                       const uint8_t *blob, size_t blobLen, uint8_t *out, uint32_t layout)
{
    const uint32_t pkgBytes = dcsPkgStride(fpw, layout);
    const uint32_t poolCap = dcsPkgImgDw(layout);
    const bool split4 = (layout & DCS_PKG_SPLIT4) != 0;
    const int sub = 64 / fpw;
    for (uint32_t c = c0 ; c < c1 ; ++c)
    {
        uint8_t *pkg = out + static_cast<size_t>(c) * pkgBytes;
        const DcsSlot *cs = slots + static_cast<size_t>(c) * static_cast<size_t>(fpw);
        memset(pkg, 0, pkgBytes);
        for (int s = 0 ; s < fpw ; ++s)
        {
            const DcsSlot &sl = cs[s];
            uint8_t *ps = pkg + static_cast<size_t>(s) * DCS_PKG_SLOT_BYTES;
            memcpy(ps, &sl, 16);                            // job, prevSlot | flags | nSrc | shiftXform, firstSrc, prevJob
            memcpy(ps + 56, &sl.poolOff, 2);
            ps[58] = sl.bpl;
            memcpy(ps + 60, &sl.nextJob, 4);
            if ((sl.flags & DCS_SLOT_EMPTY) || sl.nSrc == 0 || srcs == nullptr)
                continue;
            const DcsSrcDesc &sd = srcs[sl.firstSrc];
            memcpy(ps + 16, &sd, 40);
            uint8_t *hd = ps + 64;
            const size_t hOff = static_cast<size_t>(sd.streamOff) + 2;
            const size_t hLen = sd.hdrLen == 1 ? 1 : 16;
            for (size_t i = 0 ; i < hLen ; ++i)
                hd[i] = hOff + i < blobLen ? blob[hOff + i] : 0;
            // Which header bands the frame's q-th unpack lane (lane = s + q * fpw) takes (dcsLaneFirstBand): its first band
            // travels in bits 12..15 of the state word of the lane's split record (the record of that band's start,
            // split[band - 1]); a lane without bands has bit 15 of bitDelta set.
            const int bpl = sl.bpl;
            if (bpl == 0)
                continue;                                   // one lane unpacks the whole frame
            const int nb16 = sd.idx.nBands < 16 ? sd.idx.nBands : 16;
            const int nbEnd = dcsDealEnd(sd.format, sd.idx.nBands);
            int base[17];
            for (int q = 0 ; q <= sub ; ++q)
                base[q] = dcsLaneFirstBand(sd.format, q, bpl, nbEnd);
            for (int q = 1 ; q < sub ; ++q)
            {
                DcsSplit rec;
                memset(&rec, 0, sizeof(rec));
                if (q == sub - 1 && dcsMid15(sd.format, bpl, nb16, sd.idx.split[14].prv))
                {
                    // the second half of band 15 (1994+, one band per lane)
                    rec.bitDelta = sd.idx.split[14].prv;
                    rec.state = static_cast<uint16_t>((sd.idx.split[14].prvDelta & 0x3FFu) | DCS_SPLIT_MID15 | (15u << 12));
                }
                else if (base[q] >= nbEnd)
                    rec.bitDelta = 0x8000u;                 // no bands for this lane
                else if (base[q] >= 16)
                {
                    // OS93a Type 1, bands 16 and 17: their records travel in the frame record's bandType bytes
                    memcpy(&rec, sd.idx.bandType + (base[q] - 16) * 8, 8);
                    rec.state = static_cast<uint16_t>((rec.state & 0x0DFFu) | DCS_SPLIT_BASE16 | (static_cast<unsigned>(base[q] - 16) << 12));
                }
                else
                {
                    rec = sd.idx.split[base[q] - 1];
                    rec.state = static_cast<uint16_t>((rec.state & 0x0FFFu) | (static_cast<unsigned>(base[q]) << 12));
                }
                if (split4)
                {
                    const uint32_t w = static_cast<uint32_t>(rec.bitDelta) | (static_cast<uint32_t>(rec.state) << 16);
                    memcpy(pkg + dcsPkgOffSplit(fpw) + static_cast<size_t>(s + q * fpw) * 4, &w, 4);
                }
                else
                    memcpy(pkg + dcsPkgOffSplit(fpw) + static_cast<size_t>(s + q * fpw) * 8, &rec, 8);
            }
        }
        uint8_t *img = pkg + dcsPkgOffPool(fpw, layout);
        for (int k = 0 ; k < fpw ; ++k)
        {
            const uint32_t n = cs[k].runNDw, st = cs[k].runStartDw, o = cs[k].runPoolOff;
            if (n == 0)
                break;
            if (o + n > poolCap)
                continue;                                   // cannot happen: imgDw covers every run of the plan (dcsImageDwords)
            // dword w of the blob in bit order = its four bytes as they come; bytes past the blob read as zero
            const size_t b0 = static_cast<size_t>(st) * 4, bytes = static_cast<size_t>(n) * 4;
            uint8_t *dst = img + static_cast<size_t>(o) * 4;
            const size_t avail = b0 < blobLen ? (blobLen - b0 < bytes ? blobLen - b0 : bytes) : 0;
            // the image is an array of uint32 on a little-endian machine: byte j of the stream goes to byte (j ^ 3)
            for (size_t j = 0 ; j + 4 <= avail ; j += 4)
            {
                uint32_t w;
                memcpy(&w, blob + b0 + j, 4);
                w = __builtin_bswap32(w);
                memcpy(dst + j, &w, 4);
            }
            for (size_t j = avail & ~static_cast<size_t>(3) ; j < avail ; ++j)
                dst[j ^ 3] = blob[b0 + j];
        }
    }
}

void dcsBuildPackages(const DcsSlot *slots, uint32_t nChunks, int fpw, const DcsSrcDesc *srcs,
                      const uint8_t *blob, size_t blobLen, uint8_t *out, uint32_t layout)
{
    // (measured on the 2 x EPYC host of an MI355X box, 65 536 frames: 1 thread 2.5 ms, 4 threads 1.2 ms, 8 and 16
    // threads no faster -- the work is memory traffic -- and they slow the single-threaded planner of the next batch down)
    unsigned nt = nChunks >= 2048 ? static_cast<unsigned>(dcs_host_threads()) : 1;
    if (nt > 4) nt = 4;
    if (nt <= 1)
    {
        packChunks(slots, 0, nChunks, fpw, srcs, blob, blobLen, out, layout);
        return;
    }
    std::vector<std::thread> th;
    const uint32_t per = (nChunks + nt - 1) / nt;
    for (unsigned t = 0 ; t < nt ; ++t)
    {
        const uint32_t c0 = t * per, c1 = c0 + per < nChunks ? c0 + per : nChunks;
        if (c0 < c1)
            th.emplace_back(packChunks, slots, c0, c1, fpw, srcs, blob, blobLen, out, layout);
    }
    for (std::thread &t : th)
        t.join();
}

// Diagnostic / test entry: plan + pack on the host, exactly what dcs_batch_create uploads.
extern "C" DcsStatus dcs_pack_chunks(const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs,
                                     const uint8_t *blob, size_t blobLen, int fpw,
                                     uint8_t *out, size_t cap, uint32_t *nChunksOut, uint32_t *packageBytesOut)
{
    if (jobs == nullptr || srcs == nullptr || nChunksOut == nullptr || !(fpw == 4 || fpw == 8 || fpw == 16))
        return DCS_ERR_INVALID_ARG;
    std::vector<DcsSlot> slots;
    uint32_t imgDw = 0;
    const uint32_t nChunks = dcsPlanChunksCapped(jobs, nJobs, srcs, fpw, slots, true, 0, true, false, &imgDw, DCS_MI355X_WAVE_PLACES);
    *nChunksOut = nChunks;
    const uint32_t layout = imgDw | (dcsAllSources94(jobs, nJobs, srcs) ? DCS_PKG_SPLIT4 : 0u);
    if (packageBytesOut != nullptr)
        *packageBytesOut = dcsPkgStride(fpw, layout);
    if (out == nullptr)
        return DCS_OK;
    if (cap < static_cast<size_t>(nChunks) * dcsPkgStride(fpw, layout))
        return DCS_ERR_CAPACITY;
    dcsBuildPackages(slots.data(), nChunks, fpw, srcs, blob, blobLen, out, layout);
    return DCS_OK;
}

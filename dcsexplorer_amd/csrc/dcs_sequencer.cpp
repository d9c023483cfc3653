// dcs_sequencer.cpp -- the track-program sequencer in front of the frame decode: everything
// DCSDecoderNative::MainLoop does in a 7.68 ms tick EXCEPT decompress and transform, run on the host for
// any number of ticks ahead, producing the batch (which frame of which stream on which channel, at which
// mixing multiplier, with which shared scale) that one kernel launch then decodes.
//
// Mirrors, with the same audible result and the same bytes sent to the host:
//   MainLoop: forced-stop sweep, command queue, ExecTrack per channel, shared scale   DCSDecoderNative.cpp:89-306
//   LoadTrack / ExecTrack (opcodes 00-12), loop stack, MixingLevelOp                  :826-1371
//   LoadAudioStream / InitChannelStream / DecodeStream (looping, end of stream)       :1387-1463, :1546-1589
//   UpdateMixingLevels (fades, multiplier, track counters, host event timers)         :3042-3135
//   IRQ2Handler (data-port protocol: track commands, volume, version query)           :3297-3437
//   the sample pump's ResetException retry                                            DCSDecoder.cpp:1579-1690
// Opcodes 10-12 only feed state no audio depends on (DCSDecoderNative.h:645); they are parsed and ignored.
// Defined where the reference is not: a channel operand >= 8 or a track whose type byte is > 3 takes the
// reference's ResetException path; a zero-frame stream is not played.
#include "dcs_rom.h"
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <chrono>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <memory>
#include <vector>

namespace {

struct ResetException { };

// the blob's bytes: a vector whose resize() leaves new bytes as they are (room for a stream that is still being walked is taken
// without touching it: a megabyte of fresh pages cost 150-180 us to zero on the thread that loads the stream)
template <class T>
struct LeaveAsIs : std::allocator<T>
{
    template <class U> struct rebind { using other = LeaveAsIs<U>; };
    LeaveAsIs() = default;
    template <class U> LeaveAsIs(const LeaveAsIs<U> &) { }
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&... a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
using Bytes = std::vector<uint8_t, LeaveAsIs<uint8_t>>;

// A stream as the sequencer plays it: the index pass's record of every frame and where its bytes lie in the blob.  A long stream is
// walked by the sequencer's BACKGROUND WALKER while its first frames are already being planned, decoded and handed out (round 6):
// then `index` has its full size from the start, its first `ready` entries are valid (published by the walker, release / acquire),
// and nBytes / nValidFrames / payloadBits of `info` are only final once `done` is set.
struct StreamEntry
{
    std::vector<DcsFrameIndex> index;       // (an unusable stream -- no frames, no header -- has none)
    DcsStreamInfo info;
    uint64_t blobOff = 0;
    std::atomic<uint32_t> ready{ 0 };
    std::atomic<bool> done{ true };
    std::atomic<size_t> stableBytes{ 0 };   // bytes of the stream in the blob that no longer change (while it is being walked)
    // frames that can be planned now
    uint32_t frames() const { return done.load(std::memory_order_acquire) ? static_cast<uint32_t>(info.nValidFrames) : ready.load(std::memory_order_acquire); }
    bool usable() const { return !index.empty(); }
};

struct Mixer
{
    int cur = 0, target = 0, delta = 0, steps = 0;
    void reset() { cur = 0; target = 0; steps = 0; }           // (the delta survives, :489)
};

struct LoopPos { uint32_t counter; DcsRomCursor pos; };

struct Chan
{
    bool stop = false;
    DcsRomCursor track;                     // null = no program
    uint32_t trackCounter = 0;              // uint16 in the reference
    struct { uint8_t data = 0; uint16_t interval = 0, counter = 0; } timer;
    std::vector<LoopPos> loops;
    uint8_t nextTrackType = 0;
    uint16_t nextTrackLink = 0;
    Mixer mixer[DCS_MAX_CHANNELS];
    int sourceChannel = -1;
    uint16_t channelVolume = 0xFF;
    bool maxOverride = false;
    uint16_t mixMul = 0x7FFF;               // Channel::mixingMultiplier (DCSDecoderNative.h:514)
    // audio stream
    const StreamEntry *st = nullptr;        // null = nothing playing
    uint32_t frameCounter = 0, loopCounter = 0, pos = 0;
};

// what a snapshot holds: the decoder state proper.  Copyable: stream entries are referenced by pointer and live
// as long as the sequencer.
struct VmState
{
    uint16_t volumeMultiplier = 0x0391;     // DCSDecoderNative.h:161
    Chan ch[DCS_MAX_CHANNELS];
    std::deque<uint16_t> commandQueue;
    std::deque<uint8_t> dataPortQueue;
    uint8_t lastDataPortByte = 0;
    // host-protocol receiver (portByte): the message being assembled and how long the port has been silent
    uint8_t portMsg[4] = { 0, 0, 0, 0 };
    int portLen = 0;
    int portIdleTicks = 0;
    uint8_t variables[256] = { 0 };
    bool fatal = false;
    uint64_t fatalTick = 0;                 // the first tick that produced silence because of it
    uint64_t tick = 0;
    uint32_t idleRun = 0;                   // ticks in a row that began and ended with nothing playing, nothing queued, no program, no timer
};

}   // namespace

// Threads for the sequencers' background walks, kept for the life of the process (a decoder object per file is how the reference's
// callers work, DCSEncoder.cpp:522-540, and making a thread costs 100-200 us where the HIP runtime is loaded): a sequencer borrows
// one for the length of a walk.  A thread is told, with every job, to run anywhere BUT on the core of the thread that hands the job
// over: a thread woken through a futex is put on its waker's core, and there it would wait for the very thread that polls for its
// records (measured in the build container: the whole walk went by before the first record was seen).  Never destroyed, like the
// index pool's workers: nothing joins at process exit.
namespace {
class WalkerPool
{
public:
    struct Worker
    {
        pthread_t handle{};
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool has = false;
        void main()
        {
            for (;;)
            {
                std::function<void()> fn;
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return has; });
                    fn.swap(job);
                    has = false;
                }
                fn();
            }
        }
        void post(std::function<void()> fn)
        {
            {
                std::lock_guard<std::mutex> lk(m);
                // (anywhere but here; a process confined to one CPU keeps it, and the walk is then simply not concurrent)
                cpu_set_t allowed;
                const int here = sched_getcpu();
                if (here >= 0 && sched_getaffinity(0, sizeof(allowed), &allowed) == 0 && CPU_COUNT(&allowed) >= 2 && CPU_ISSET(here, &allowed))
                {
                    // (not on this core's other hardware thread either, if there is room elsewhere: the caller polls and pumps)
                    cpu_set_t elsewhere = allowed;
                    char path[96], text[128] = { 0 };
                    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", here);
                    if (FILE *f = fopen(path, "r"))
                    {
                        if (fgets(text, sizeof(text), f) != nullptr)
                            for (const char *q = text ; *q != 0 ; )
                            {
                                char *end = nullptr;
                                long a = strtol(q, &end, 10), b = a;
                                if (end == q)
                                    break;
                                if (*end == '-')
                                    b = strtol(end + 1, &end, 10);
                                for (long c = a ; c <= b && c < CPU_SETSIZE ; ++c)
                                    CPU_CLR(static_cast<int>(c), &elsewhere);
                                q = *end == ',' ? end + 1 : end;
                                if (*end != ',')
                                    break;
                            }
                        fclose(f);
                    }
                    CPU_CLR(here, &allowed);
                    CPU_CLR(here, &elsewhere);
                    (void)pthread_setaffinity_np(handle, sizeof(allowed), CPU_COUNT(&elsewhere) >= 1 ? &elsewhere : &allowed);
                }
                job = std::move(fn);
                has = true;
            }
            cv.notify_one();
        }
    };
    static WalkerPool &get()
    {
        static WalkerPool *pool = new WalkerPool;
        return *pool;
    }
    Worker *take()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            if (!idle.empty())
            {
                Worker *w = idle.back();
                idle.pop_back();
                return w;
            }
        }
        Worker *w = new (std::nothrow) Worker;
        if (w == nullptr)
            return nullptr;
        try
        {
            std::thread th([w] { w->main(); });
            w->handle = th.native_handle();
            th.detach();                    // (the handle stays good: the thread never ends)
        }
        catch (...)
        {
            delete w;                       // (no thread to be had: the caller walks its stream itself)
            return nullptr;
        }
        return w;
    }
    void give(Worker *w)
    {
        std::lock_guard<std::mutex> lk(m);
        idle.push_back(w);
    }
private:
    std::mutex m;
    std::vector<Worker *> idle;
};
}   // namespace

struct DcsSequencer : VmState
{
    const DcsRomSet *rs = nullptr;
    DcsRomSet emptyRoms;                    // stand-alone mode: no ROMs, streams come from caller memory
    int os = DCS_OS94;
    bool totan = false;                     // GameID::TOTAN quirk of the data-port handler (:3345-3351)
    uint16_t reportedVersion = 0x0106;

    // the batch planned so far, and what is needed to go back inside it
    std::map<std::pair<const void *, size_t>, std::unique_ptr<StreamEntry>> streams;
    std::vector<std::unique_ptr<StreamEntry>> uncached;                   // streams loaded from caller memory
    Bytes blob;
    std::vector<DcsSrcDesc> srcs;
    std::vector<DcsFrameJob> jobs;
    std::vector<DcsHostByte> hostBytes;
    // Going back inside the current batch (dcs_seq_rewind).  A snapshot of the machine costs a microsecond (two queues, eight loop
    // stacks), a tick a tenth of that, and the machine is deterministic: so a snapshot is kept every kSnapStride ticks only, and
    // going back to tick k means restoring the last snapshot at or before k and running the ticks in between again, with the bytes
    // for the host (which were already handed out) suppressed.  (Round 5 kept one per tick: 2 KB and a microsecond a frame.)
    struct Snapshot { uint32_t ticks; size_t nSrcs; VmState st; };
    static const uint32_t kSnapStride = 64;
    std::vector<Snapshot> history;          // ascending by ticks; history[0].ticks == 0: the state the batch started from
    std::vector<uint8_t> playingAfter;      // per tick of the current batch: which channels have a stream loaded after it (IsStreamPlaying)
    std::vector<uint8_t> tracksAfter;       // ... and which have a track program (ClearTracks changes nothing where both are zero)
    std::vector<int16_t> batchTails;        // after a decode: the 16-sample tail each tick of the batch left
    int16_t batchTail0[16] = { 0 };         // ... and the tail the batch started from
    bool batchDecoded = false;
    bool keepHistory = false;               // dcs_seq_set_rewindable
    bool replaying = false;                 // ticks run again by dcs_seq_rewind: nothing goes to the host twice
    int16_t tail[16] = { 0 };
    uint64_t blobId = 0;                    // names the blob for the context's live decoder: append-only under one name (dcs_decode_batch_live)
    std::string lastError;

    void toHost(uint8_t b)
    {
        if (!replaying)
            hostBytes.push_back(DcsHostByte{ static_cast<uint32_t>(tick), static_cast<uint32_t>(b) });
    }
    uint32_t batchTicks() const { return static_cast<uint32_t>(playingAfter.size()); }
    bool quiescent() const
    {
        if (!commandQueue.empty() || !dataPortQueue.empty())
            return false;
        for (const Chan &c : ch)
            if (c.st != nullptr || !c.track.isNull() || c.timer.interval != 0 || c.stop)
                return false;
        return true;
    }
    uint8_t trackMask() const
    {
        uint8_t m = 0;
        for (int c = 0 ; c < DCS_MAX_CHANNELS ; ++c)
            if (!ch[c].track.isNull())
                m |= static_cast<uint8_t>(1u << c);
        return m;
    }
    uint8_t playingMask() const
    {
        uint8_t m = 0;
        for (int c = 0 ; c < DCS_MAX_CHANNELS ; ++c)
            if (ch[c].st != nullptr)
                m |= static_cast<uint8_t>(1u << c);
        return m;
    }
    void startBatch()
    {
        batchDecoded = false;
        history.clear();
        history.push_back(Snapshot{ 0, srcs.size(), static_cast<const VmState &>(*this) });
        playingAfter.clear();
        tracksAfter.clear();
        batchTails.clear();
    }
    // A command from outside (data port, track command, volume, a stream loaded, tracks cleared) changes the machine where it
    // stands, behind the batch's last tick: going back to any LATER tick must run from a state that has the command in it, so the
    // state right behind the command becomes the snapshot of that tick.
    void noteCommand()
    {
        if (history.empty())
            return;
        if (history.back().ticks == batchTicks())
        {
            history.back().st = static_cast<const VmState &>(*this);
            history.back().nSrcs = srcs.size();
        }
        else if (keepHistory)
            history.push_back(Snapshot{ batchTicks(), srcs.size(), static_cast<const VmState &>(*this) });
    }
    void tickDone(bool wasQuiet)            // bookkeeping behind every tick of a batch (wasQuiet: quiescent when the tick began)
    {
        idleRun = wasQuiet && quiescent() ? idleRun + 1 : 0;
        playingAfter.push_back(playingMask());
        tracksAfter.push_back(trackMask());
        if (keepHistory && batchTicks() % kSnapStride == 0 && history.back().ticks != batchTicks())
            history.push_back(Snapshot{ batchTicks(), srcs.size(), static_cast<const VmState &>(*this) });
    }
    void advance(uint32_t nTicks, uint32_t stopWhenIdleFor, uint32_t *ran);
    Chan &chan(uint32_t c)
    {
        if (c >= DCS_MAX_CHANNELS)
            throw ResetException();
        return ch[c];
    }
    void resetMixingLevels(int c)           // :3233-3239: channel c's contribution to every channel's mix
    {
        for (Chan &x : ch)
            x.mixer[c].reset();
    }

    // ---- the background walker (a thread borrowed from WalkerPool for the length of a walk): one stream at a time, always the last
    // one in the blob (its bytes are copied in as the walk reaches them); anything else that wants to append to the blob, a caller's
    // command after which the stream's memory may go away (ClearTracks, another LoadAudioStream) and the destructor wait for it
    static const uint32_t kWalkInlineFrames = 384;      // shorter streams are walked where they are loaded (a hand-over costs ~30 us)
    struct Walk { StreamEntry *e = nullptr; const uint8_t *src = nullptr; size_t avail = 0; size_t reserved = 0; };
    WalkerPool::Worker *walkWorker = nullptr;
    StreamEntry *walking = nullptr;         // (sequencer thread only) the entry whose walk has not been taken in yet
    size_t walkReserved = 0;
    void runWalk(const Walk &w);
    void finishWalk();                      // wait for the walker, shrink the blob to what the stream really takes
    size_t blobStableLen() const { return walking != nullptr ? static_cast<size_t>(walking->blobOff) + walking->stableBytes.load(std::memory_order_acquire) : blob.size(); }
    uint32_t plannableTicks() const;        // ticks the channels' streams have records for right now (UINT32_MAX: no limit)
    ~DcsSequencer();

    const StreamEntry *streamAt(DcsRomCursor p);
    const StreamEntry *addStream(const uint8_t *data, size_t avail, std::pair<const void *, size_t> key, bool cache);
    void compact();
    static uint64_t newBlobId();
    void loadAudioStream(uint32_t streamChannel, int sourceChannel, uint32_t loopCounter, DcsRomCursor p);
    void loadStreamEntry(uint32_t streamChannel, int sourceChannel, uint32_t loopCounter, const StreamEntry *e);
    void loadTrack(uint32_t c, DcsRomCursor p);
    void execTrack(int c);
    void mixingLevelOp(int cur, DcsRomCursor &p, int mode, bool fade);
    void portByte(uint8_t data);
    void mainLoop();
};

// the stream at a ROM position: indexed once, its bytes copied once into the batch's blob
const StreamEntry *DcsSequencer::streamAt(DcsRomCursor p)
{
    const auto key = std::make_pair(static_cast<const void *>(p.rom), p.pos);
    auto it = streams.find(key);
    if (it != streams.end())
        return it->second.get();
    return addStream(p.rom->data() + p.pos, p.rom->size() - p.pos, key, true);
}

const StreamEntry *DcsSequencer::addStream(const uint8_t *data, size_t avail, std::pair<const void *, size_t> key, bool cache)
{
    finishWalk();                           // (the blob grows at its end: only one stream at a time can be filling in there)
    std::unique_ptr<StreamEntry> e(new StreamEntry);
    const uint32_t nFrames = avail >= 2 ? (static_cast<uint32_t>(data[0]) << 8) | data[1] : 0;
    static const bool noWalker = getenv("DCS_SEQ_NO_WALKER") != nullptr && atoi(getenv("DCS_SEQ_NO_WALKER")) != 0;
    // (a long stream goes to a walker thread -- if one is to be had: where no thread can be made the stream is walked here)
    WalkerPool::Worker *worker = (nFrames > kWalkInlineFrames && avail >= 3 && !noWalker) ? WalkerPool::get().take() : nullptr;
    const bool background = worker != nullptr;
    if (nFrames != 0 && avail >= 3 && !background)
    {
        e->index.resize(nFrames);
        if (dcs_index_stream(static_cast<DcsOsVersion>(os), data, avail, e->index.data(), nFrames, &e->info) != DCS_OK)
            e->index.clear();
        else
            e->index.resize(static_cast<size_t>(e->info.nValidFrames));
        if (!e->index.empty())
        {
            while (blob.size() & 3)
                blob.push_back(0);
            e->blobOff = blob.size();
            const size_t used = static_cast<size_t>(e->info.nBytes) < avail ? static_cast<size_t>(e->info.nBytes) : avail;
            blob.insert(blob.end(), data, data + used);
            blob.insert(blob.end(), static_cast<size_t>(e->info.nBytes) - used + 16, 0);
        }
    }
    else if (background && dcsStreamContainer(static_cast<DcsOsVersion>(os), data, avail, &e->info) == DCS_OK)
    {
        // room for the largest the stream can be (a frame is at most DCS_MAX_FRAME_BITS long), left as it is: the walker copies the
        // stream's bytes in as it reaches them (what lies behind them is never decoded, only looked ahead at) and finishWalk() gives
        // back what was not needed
        e->index.resize(nFrames);
        while (blob.size() & 3)
            blob.push_back(0);
        e->blobOff = blob.size();
        size_t bound = 2 + static_cast<size_t>(e->info.hdrLen) + static_cast<size_t>(nFrames) * (DCS_MAX_FRAME_BITS / 8) + 64;
        if (bound > avail + 64)
            bound = avail + 64;
        blob.resize(blob.size() + bound);
        e->done.store(false, std::memory_order_relaxed);
        walking = e.get();
        walkReserved = bound;
        const Walk w{ e.get(), data, avail, bound };
        walkWorker = worker;
        walkWorker->post([this, w] { runWalk(w); });
    }
    else if (worker != nullptr)
        WalkerPool::get().give(worker);         // (a stream without a usable container: nothing to walk)
    StreamEntry *r = e.get();
    if (cache)
        streams[key] = std::move(e);
    else
        uncached.push_back(std::move(e));
    return r;
}

// the walk of one stream, on the walker's thread: every frame's record goes to its place in the entry, the stream's bytes into the
// blob up to where the walk has read, and a frame is PUBLISHED once everything a decode of it will read lies there for good (its
// own bytes and the sixteen the bit pool's look-ahead may take in behind them)
void DcsSequencer::runWalk(const Walk &w)
{
    StreamEntry *e = w.e;
    uint8_t *dst = blob.data() + e->blobOff;        // (stable: nothing else appends while a walk is on)
    size_t copied = 0;
    uint32_t published = 0;
    const size_t hdr = 2 + static_cast<size_t>(e->info.hdrLen);
    auto frameEnd = [&](uint32_t f) { return hdr + (static_cast<size_t>(e->index[f].bitOff) + e->index[f].nBits + 7) / 8; };
    auto copyTo = [&](size_t upTo) {
        upTo = std::min(std::min(upTo, w.avail), w.reserved);
        if (upTo > copied)
        {
            memcpy(dst + copied, w.src + copied, upTo - copied);
            copied = upTo;
        }
    };
    DcsStreamInfo info;
    const std::function<void(uint32_t, const DcsFrameIndex &)> onFrame = [&](uint32_t f, const DcsFrameIndex &fi) {
        e->index[f] = fi;
        copyTo(frameEnd(f) + 4);                    // (as far as the walk's own reader has looked: bytes the caller's buffer is sure to have)
        while (published < f && frameEnd(published) + 16 <= copied)
            ++published;
        e->stableBytes.store(copied >= 16 ? copied - 16 : 0, std::memory_order_relaxed);
        e->ready.store(published, std::memory_order_release);
    };
    const auto tw0 = std::chrono::steady_clock::now();
    dcsIndexStreamProgressive(static_cast<DcsOsVersion>(os), w.src, w.avail, &info, onFrame);
    if (getenv("DCS_SEQ_WALK_TRACE"))
        fprintf(stderr, "walk of %d frames: %.1f us, published %u before the end\n", info.nFrames,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw0).count(), published);
    copyTo(static_cast<size_t>(info.nBytes));
    if (copied + 16 <= w.reserved)
        memset(dst + copied, 0, 16);                // (what a stream walked at once has behind it)
    // (the summary's container fields are what dcsStreamContainer gave and the planner reads them meanwhile; the rest is final now)
    e->info.nBytes = info.nBytes;
    e->info.nValidFrames = info.nValidFrames;
    e->info.payloadBits = info.payloadBits;
    e->stableBytes.store(std::min(static_cast<size_t>(info.nBytes) + 16, w.reserved), std::memory_order_relaxed);
    e->done.store(true, std::memory_order_release);
}

void DcsSequencer::finishWalk()
{
    if (walking == nullptr)
        return;
    StreamEntry *e = walking;
    // (its last records come within microseconds of each other: look again at once)
    for (unsigned spins = 0 ; !e->done.load(std::memory_order_acquire) ; ++spins)
    {
        if (spins < 20000)
            __builtin_ia32_pause();
        else
            std::this_thread::yield();
    }
    walking = nullptr;
    WalkerPool::get().give(walkWorker);
    walkWorker = nullptr;
    // the blob ends with this stream: keep what it takes (as a stream walked at once would have got)
    const size_t keep = std::min(static_cast<size_t>(e->info.nBytes) + 16, walkReserved);
    blob.resize(static_cast<size_t>(e->blobOff) + keep);
    if (e->info.nValidFrames <= 0)
        e->index.clear();
}

DcsSequencer::~DcsSequencer()
{
    finishWalk();                           // (a walk in progress runs to its end first: at most a few milliseconds)
}

// ticks that can be planned without running ahead of the walker: the fewest records a channel's stream has beyond its position
uint32_t DcsSequencer::plannableTicks() const
{
    if (walking == nullptr || walking->done.load(std::memory_order_acquire))
        return 0xFFFFFFFFu;
    uint32_t n = 0xFFFFFFFFu;
    const uint32_t have = walking->ready.load(std::memory_order_acquire);
    for (const Chan &c : ch)
        if (c.st == walking)
            n = std::min(n, have > c.pos ? have - c.pos : 0u);
    return n;
}

uint64_t DcsSequencer::newBlobId()
{
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}

// Between batches: when the blob has grown large (a long-running decoder keeps loading streams), keep only the
// streams a channel is still playing.
void DcsSequencer::compact()
{
    if (blob.size() < (32u << 20))
        return;
    finishWalk();
    Bytes fresh;
    auto keep = [&](StreamEntry *e) {
        const size_t n = static_cast<size_t>(e->info.nBytes) + 16;
        while (fresh.size() & 3)
            fresh.push_back(0);
        const uint64_t off = fresh.size();
        fresh.insert(fresh.end(), blob.begin() + static_cast<long>(e->blobOff), blob.begin() + static_cast<long>(e->blobOff + n));
        e->blobOff = off;
    };
    auto playing = [&](const StreamEntry *e) {
        for (const Chan &c : ch)
            if (c.st == e)
                return true;
        return false;
    };
    for (auto it = streams.begin() ; it != streams.end() ; )
        if (playing(it->second.get())) { keep(it->second.get()); ++it; }
        else it = streams.erase(it);
    for (auto it = uncached.begin() ; it != uncached.end() ; )
        if (playing(it->get())) { keep(it->get()); ++it; }
        else it = uncached.erase(it);
    blob.swap(fresh);
    blobId = newBlobId();                   // (what the live decoder keeps resident under the old name is no longer this blob)
}

// LoadAudioStream (:1408-1431) + InitChannelStream (:1433-1463)
void DcsSequencer::loadAudioStream(uint32_t streamChannel, int source, uint32_t loopCounter, DcsRomCursor p)
{
    chan(streamChannel);                    // validate before touching the stream cache
    loadStreamEntry(streamChannel, source, loopCounter, streamAt(p));
}

void DcsSequencer::loadStreamEntry(uint32_t streamChannel, int source, uint32_t loopCounter, const StreamEntry *e)
{
    Chan &c = chan(streamChannel);
    if (!e->usable())
    {
        c.st = nullptr;                     // zero frames (or unusable): nothing to play
        return;
    }
    c.st = e;
    c.frameCounter = static_cast<uint32_t>(e->info.nFrames);
    c.pos = 0;
    c.loopCounter = loopCounter;
    if (c.sourceChannel >= 0 && c.sourceChannel != source)
        c.mixer[c.sourceChannel].reset();
    c.sourceChannel = source;
}

void DcsSequencer::loadTrack(uint32_t cn, DcsRomCursor p)       // :826-836
{
    Chan &c = chan(cn);
    c.track = p;
    c.st = nullptr;
    c.trackCounter = 0;
    c.timer.interval = c.timer.counter = 0;
    c.loops.clear();
    resetMixingLevels(static_cast<int>(cn));
}

void DcsSequencer::mixingLevelOp(int cur, DcsRomCursor &p, int mode, bool fade)     // :1316-1371
{
    const uint32_t target = p.u8();
    const int param = static_cast<int>(static_cast<int8_t>(p.u8())) * 64;
    const int steps = fade ? static_cast<int>(p.u16()) : 0;
    Mixer &m = chan(target).mixer[cur];
    m.steps = steps;
    const int oldLevel = m.cur;
    int newLevel = mode == 0 ? param : mode == 1 ? oldLevel + param : oldLevel - param;
    const int delta = newLevel - oldLevel;
    newLevel = newLevel > 8191 ? 8191 : newLevel < -8191 ? -8191 : newLevel;
    m.target = newLevel;
    if (steps != 0)
        m.delta = delta / steps;
    else
        m.cur = newLevel;
}

void DcsSequencer::execTrack(int cur)       // :838-1290
{
    Chan &me = ch[cur];
    DcsRomCursor p = me.track;
    if (p.isNull())
        return;
    for (;;)
    {
        const uint32_t countPrefix = p.u16();
        if (countPrefix == 0xFFFF || (me.trackCounter & 0xFFFF) != countPrefix)
        {
            p.skip(-2);
            me.track = p;
            return;
        }
        me.trackCounter = 0;
        const uint32_t opcode = p.u8();
        switch (opcode)
        {
        case 0x00:
            me.track.clear();
            me.st = nullptr;
            me.loops.clear();
            me.timer.interval = me.timer.counter = 0;
            resetMixingLevels(cur);
            return;

        case 0x01:
            {
                const uint32_t streamChannel = p.u8();
                if (streamChannel == 5)
                    ch[5].maxOverride = false;
                const DcsRomCursor stream = rs->at(p.u24());
                const uint32_t loopCounter = p.u8();
                loadAudioStream(streamChannel, cur, loopCounter, stream);
            }
            break;

        case 0x02:
            {
                Chan &t = chan(p.u8());
                if (t.st != nullptr)
                {
                    t.st = nullptr;
                    resetMixingLevels(static_cast<int>(&t - ch));
                }
                t.track.clear();
                t.timer.interval = t.timer.counter = 0;
                if (me.track.isNull())
                    return;
            }
            break;

        case 0x03:
            commandQueue.push_back(static_cast<uint16_t>(p.u16()));
            break;

        case 0x04:
            if (os == DCS_OS93A)
            {
                const uint8_t cmdByte = static_cast<uint8_t>(p.u8());
                const uint16_t counter = static_cast<uint16_t>(p.u16());
                if (cmdByte == 0)
                    me.timer.interval = me.timer.counter = 0;
                else
                {
                    toHost(cmdByte);
                    me.timer.data = cmdByte;
                    me.timer.interval = me.timer.counter = counter;         // 0 = cleared
                }
            }
            else
            {
                const uint8_t b = static_cast<uint8_t>(p.u8());
                toHost(b);
                if (rs->nominalVersion == 0x0105)
                {
                    if (b == 0x69) ch[5].maxOverride = true;
                    else if (b == 0x6A) ch[5].maxOverride = false;
                }
            }
            break;

        case 0x05:
            {
                Chan &t = chan(p.u8());
                const uint8_t type = t.nextTrackType;
                if (type == 0)
                    break;
                t.nextTrackType = 0;
                if (type == 2)
                    commandQueue.push_back(t.nextTrackLink);
                else if (type == 3)
                {
                    const uint32_t lo = t.nextTrackLink & 0xFF, hi = (t.nextTrackLink >> 8) & 0xFF;
                    DcsRomCursor table = rs->at(rs->u2U24(rs->indirectIndex + lo * 3));
                    table.skip(static_cast<long>(variables[hi]) * 2);
                    commandQueue.push_back(static_cast<uint16_t>(table.u16()));
                }
            }
            break;

        case 0x06:
            if (os != DCS_OS93A && os != DCS_OS93B)     // (the 1993 software reads no operands here, :1092-1100)
            {
                const uint32_t var = p.u8();
                variables[var] = static_cast<uint8_t>(p.u8());
            }
            break;

        case 0x07: case 0x08: case 0x09:
            mixingLevelOp(cur, p, static_cast<int>(opcode) - 0x07, false);
            break;
        case 0x0A: case 0x0B: case 0x0C:
            mixingLevelOp(cur, p, static_cast<int>(opcode) - 0x0A, true);
            break;

        case 0x0D:
            break;

        case 0x0E:
            {
                const uint32_t counter = p.u8();
                me.loops.push_back(LoopPos{ counter, p });
            }
            break;

        case 0x0F:
            if (!me.loops.empty())
            {
                LoopPos &top = me.loops.back();
                if (top.counter == 0)
                    p = top.pos;
                else if (top.counter == 1)
                    me.loops.pop_back();
                else
                {
                    --top.counter;
                    p = top.pos;
                }
            }
            break;

        case 0x10:
            p.skip(2);
            break;
        case 0x11: case 0x12:
            p.skip(4);
            break;

        default:
            throw ResetException();
        }
    }
}

// ---- the host's side of the data port ----------------------------------------------------------------------
// What the WPC board sends is a stream of MESSAGES: a 16-bit command word, high byte first, followed -- for the
// commands that carry a parameter -- by the parameter byte and its complement.  The table names every range of command
// words the firmware tells apart; everything it does not list (and that has bit 15 clear) is a track number.  Same
// behaviour as DCSDecoderNative::IRQ2Handler (DCSDecoderNative.cpp:3297-3437), which walks the same protocol as a
// four-state byte counter.
namespace {
enum class PortAction : uint8_t { MasterVolume, ChannelVolume, Nothing, VersionHigh, VersionLow };
struct PortCommand
{
    uint16_t first, last;       // command words [first, last]
    uint8_t length;             // bytes of the whole message: 2, or 4 with parameter and complement
    PortAction action;
};
const PortCommand kPortCommands[] = {
    { 0x55AA, 0x55AA, 4, PortAction::MasterVolume },
    { 0x55AB, 0x55B2, 4, PortAction::ChannelVolume },       // channels 0..7
    { 0x55B3, 0x55B9, 2, PortAction::Nothing },
    { 0x55BA, 0x55C1, 4, PortAction::Nothing },             // a per-channel parameter nothing audible depends on (:3400-3420)
    { 0x55C2, 0x55C2, 2, PortAction::VersionHigh },
    { 0x55C3, 0x55C3, 2, PortAction::VersionLow },
    { 0x8000, 0xFFFF, 2, PortAction::Nothing },
};
const int kPortSilenceTicks = 13;   // a message whose next byte takes this long is forgotten (:3301-3306)
}   // namespace

void DcsSequencer::portByte(uint8_t data)
{
    if (portIdleTicks >= kPortSilenceTicks)
        portLen = 0;                                        // whatever was being assembled has gone stale
    portIdleTicks = 0;
    portMsg[portLen++] = data;
    if (portLen < 2)
        return;
    const uint16_t word = static_cast<uint16_t>((portMsg[0] << 8) | portMsg[1]);
    const PortCommand *cmd = nullptr;
    for (const PortCommand &c : kPortCommands)
        if (word >= c.first && word <= c.last)
        {
            cmd = &c;
            break;
        }
    if (cmd == nullptr)
    {
        // a track number -- except one word "Tales of the Arabian Nights" answers itself (:3345-3351)
        portLen = 0;
        if (word == 0x03E7 && totan)
            toHost(0x11);
        else
            commandQueue.push_back(word);
        return;
    }
    if (portLen < cmd->length)
        return;                                             // parameter and complement still to come
    portLen = 0;
    const uint8_t value = portMsg[2];
    if (cmd->length == 4 && portMsg[3] != static_cast<uint8_t>(~value))
        return;                                             // complement does not match: the message is dropped
    switch (cmd->action)
    {
    case PortAction::MasterVolume:
        volumeMultiplier = dcs_volume_multiplier(value);
        break;
    case PortAction::ChannelVolume:
        ch[word - 0x55AB].channelVolume = value;
        break;
    case PortAction::VersionHigh:
        toHost(static_cast<uint8_t>(reportedVersion >> 8));
        break;
    case PortAction::VersionLow:
        toHost(static_cast<uint8_t>(reportedVersion & 0xFF));
        break;
    case PortAction::Nothing:
        break;
    }
}

// one MainLoop pass minus decompress/transform: appends one job (and its sources) to the plan
void DcsSequencer::mainLoop()
{
    // forced-stop sweep (:95-116)
    for (int c = 0 ; c < DCS_MAX_CHANNELS ; ++c)
        if (ch[c].stop)
        {
            ch[c].stop = false;
            if (ch[c].st != nullptr)
            {
                ch[c].st = nullptr;
                resetMixingLevels(c);
            }
            ch[c].timer.interval = ch[c].timer.counter = 0;
            ch[c].track.clear();
        }

    // pending commands = track numbers (:129-171)
    while (!commandQueue.empty())
    {
        const uint16_t cmd = commandQueue.front();
        commandQueue.pop_front();
        if (cmd >= rs->nTracks)
            continue;
        const uint32_t trackOfs = rs->u2U24(rs->trackIndex + static_cast<size_t>(cmd) * 3);
        if ((trackOfs & 0xFF0000u) == 0xFF0000u)
            continue;
        DcsRomCursor p = rs->at(trackOfs);
        const uint32_t type = p.u8(), c = p.u8();
        if (type == 1)
            loadTrack(c, p);
        else if (type <= 3)
        {
            Chan &t = chan(c);
            t.nextTrackType = static_cast<uint8_t>(type);
            t.nextTrackLink = static_cast<uint16_t>(p.u16());
        }
        else
            throw ResetException();
    }

    // the track programs, channel by channel (:184-197)
    for (int c = 0 ; c < DCS_MAX_CHANNELS ; ++c)
        execTrack(c);

    // shared scale (:227-269)
    uint16_t vol[DCS_MAX_CHANNELS], mm[DCS_MAX_CHANNELS];
    uint8_t counted[DCS_MAX_CHANNELS];
    for (int c = 0 ; c < DCS_MAX_CHANNELS ; ++c)
    {
        vol[c] = ch[c].maxOverride ? 0x7FFE : volumeMultiplier;
        counted[c] = (ch[c].maxOverride || ch[c].st != nullptr) ? 1 : 0;
        mm[c] = ch[c].mixMul;
    }
    const int volShift = dcsFrameScaleV(vol, mm, counted, DCS_MAX_CHANNELS);

    // DecodeStream per channel (:1546-1589): which frame of which stream, then the stream's own bookkeeping
    DcsFrameJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.firstSrc = static_cast<uint32_t>(srcs.size());
    jb.volShift = static_cast<uint8_t>(volShift);
    jb.xform = (os == DCS_OS93A || os == DCS_OS93B) ? DCS_XFORM_93 : DCS_XFORM_94;
    jb.prev = jobs.empty() ? (DCS_PREV_EXT | 0u) : static_cast<uint32_t>(jobs.size() - 1);
    for (int c = 0 ; c < DCS_MAX_CHANNELS ; ++c)
    {
        Chan &x = ch[c];
        if (x.st == nullptr)
            continue;
        if (x.pos < x.st->frames())
        {
            DcsSrcDesc sd;
            memset(&sd, 0, sizeof(sd));
            sd.streamOff = x.st->blobOff;
            sd.mixMul = mm[c];
            sd.format = static_cast<uint8_t>(x.st->info.format);
            sd.hdrLen = static_cast<uint8_t>(x.st->info.hdrLen);
            sd.idx = x.st->index[x.pos];
            srcs.push_back(sd);
            ++jb.nSrc;
            if ((sd.idx.flags >> 4) != 0)
                x.stop = true;              // the frame decoder raises the channel's stop flag (:1989, :2216)
        }
        ++x.pos;
        if (--x.frameCounter != 0)
            continue;
        x.frameCounter = static_cast<uint32_t>(x.st->info.nFrames);
        x.pos = 0;
        if (x.loopCounter == 0)
            continue;
        if (--x.loopCounter != 0)
            continue;
        x.st = nullptr;
        x.sourceChannel = -1;
    }
    jobs.push_back(jb);

    // UpdateMixingLevels (:3042-3135): fades, next tick's multipliers, track counters, host event timers
    for (Chan &x : ch)
        for (Mixer &m : x.mixer)
        {
            if (m.steps == 1)
            {
                m.steps = 0;
                m.cur = m.target;
            }
            else if (m.steps > 1)
            {
                --m.steps;
                m.cur += m.delta;
                m.cur = m.cur > 8191 ? 8191 : m.cur < -8191 ? -8191 : m.cur;
            }
        }
    for (Chan &x : ch)
    {
        int sum = 0;
        for (const Mixer &m : x.mixer)
            sum += m.cur;
        x.mixMul = x.maxOverride ? dcs_mixing_multiplier(DCS_OS95, sum, 0xFF)
                                 : dcs_mixing_multiplier(static_cast<DcsOsVersion>(os), sum, x.channelVolume);
    }
    for (Chan &x : ch)
    {
        ++x.trackCounter;
        if (x.timer.interval != 0 && --x.timer.counter == 0)
        {
            x.timer.counter = x.timer.interval;
            toHost(x.timer.data);
        }
    }
    if (portIdleTicks < kPortSilenceTicks)
        ++portIdleTicks;
}

// ---------------------------------------------------------------------------------------------------------
extern "C" DcsSequencer *dcs_seq_create_standalone(DcsOsVersion os)
{
    if (os < DCS_OS93A || os > DCS_OS95)
        return nullptr;
    DcsSequencer *s = new (std::nothrow) DcsSequencer;
    if (s == nullptr)
        return nullptr;
    s->emptyRoms.missing.assign(0x2000, 0xFF);
    s->emptyRoms.os = os;
    s->rs = &s->emptyRoms;
    s->os = os;
    s->blobId = DcsSequencer::newBlobId();
    s->startBatch();
    return s;
}

extern "C" DcsSequencer *dcs_seq_create(const DcsRomSet *rs)
{
    if (rs == nullptr || rs->os < 0 || !rs->rom[0].present)
        return nullptr;
    DcsSequencer *s = new (std::nothrow) DcsSequencer;
    if (s == nullptr)
        return nullptr;
    s->rs = rs;
    s->os = rs->os;
    // the U2 signature names the game; one title has a data-port quirk (DCSDecoder.cpp:147, DCSDecoderNative.cpp:3345)
    const std::vector<uint8_t> &u2 = rs->rom[0].data;
    static const char key[] = "arabian nights";
    for (size_t i = 4 ; i + sizeof(key) - 1 <= 128 && i + sizeof(key) - 1 <= u2.size() && u2[i] != 0 ; ++i)
    {
        size_t k = 0;
        while (k < sizeof(key) - 1 && (u2[i + k] | 0x20) == key[k]) ++k;
        if (k == sizeof(key) - 1) { s->totan = true; break; }
    }
    s->blobId = DcsSequencer::newBlobId();
    s->startBatch();
    return s;
}

extern "C" void dcs_seq_destroy(DcsSequencer *s) { delete s; }
extern "C" const char *dcs_seq_last_error(const DcsSequencer *s) { return s != nullptr ? s->lastError.c_str() : ""; }

extern "C" DcsStatus dcs_seq_set_master_volume(DcsSequencer *s, int vol)
{
    if (s == nullptr) return DCS_ERR_INVALID_ARG;
    s->volumeMultiplier = dcs_volume_multiplier(vol);
    s->noteCommand();
    return DCS_OK;
}

extern "C" DcsStatus dcs_seq_set_reported_version(DcsSequencer *s, uint16_t v)
{
    if (s == nullptr) return DCS_ERR_INVALID_ARG;
    s->reportedVersion = v;
    return DCS_OK;
}

extern "C" DcsStatus dcs_seq_write_data_port(DcsSequencer *s, uint8_t byte)
{
    if (s == nullptr) return DCS_ERR_INVALID_ARG;
    s->dataPortQueue.push_back(byte);
    s->noteCommand();
    return DCS_OK;
}

extern "C" DcsStatus dcs_seq_add_track_command(DcsSequencer *s, uint16_t track)
{
    if (s == nullptr) return DCS_ERR_INVALID_ARG;
    s->commandQueue.push_back(track);
    s->noteCommand();
    return DCS_OK;
}

extern "C" DcsStatus dcs_seq_clear_tracks(DcsSequencer *s)       // :1466-1473
{
    if (s == nullptr) return DCS_ERR_INVALID_ARG;
    s->finishWalk();                        // (behind this call the caller may take a loaded stream's memory away)
    for (Chan &c : s->ch)
    {
        c.track.clear();
        c.st = nullptr;
    }
    s->noteCommand();
    return DCS_OK;
}

extern "C" DcsStatus dcs_seq_load_audio_stream(DcsSequencer *s, int channel, uint32_t linearAddress, int mixingLevel)     // :1387-1406
{
    if (s == nullptr || channel < 0 || channel >= DCS_MAX_CHANNELS) return DCS_ERR_INVALID_ARG;
    Chan &c = s->ch[channel];
    c.track.clear();
    s->loadAudioStream(static_cast<uint32_t>(channel), channel, 1, s->rs->at(linearAddress));
    Mixer &m = c.mixer[channel];
    m.reset();
    m.cur = m.target = mixingLevel * 64;
    s->noteCommand();
    return DCS_OK;
}

// LoadAudioStream for a stream that lives in caller memory rather than in a ROM image (the ROM-less recipe of
// DCSEncoder.cpp:522-571); the bytes the stream uses are copied now, the buffer is not referenced later.  Bytes
// past `len` read as zero.
extern "C" DcsStatus dcs_seq_load_audio_stream_mem(DcsSequencer *s, int channel, const uint8_t *data, size_t len, int mixingLevel)
{
    if (s == nullptr || data == nullptr || len < 3 || channel < 0 || channel >= DCS_MAX_CHANNELS) return DCS_ERR_INVALID_ARG;
    Chan &c = s->ch[channel];
    c.track.clear();
    s->loadStreamEntry(static_cast<uint32_t>(channel), channel, 1, s->addStream(data, len, std::make_pair(static_cast<const void *>(data), size_t(0)), false));
    Mixer &m = c.mixer[channel];
    m.reset();
    m.cur = m.target = mixingLevel * 64;
    s->noteCommand();
    return DCS_OK;
}

// Go back inside the current batch (the ticks planned since the last decode, or -- right after a decode -- the
// ticks just decoded): the decoder state, the overlap tail and the host bytes become what they were after the
// batch's first `keepTicks` ticks; later ticks are dropped.  This is how a caller that decodes ahead of time
// stays exact when a command arrives: rewind to the last frame it handed out, apply the command, plan again.
extern "C" DcsStatus dcs_seq_set_rewindable(DcsSequencer *s, int on)
{
    if (s == nullptr || !s->jobs.empty())
        return DCS_ERR_INVALID_ARG;             // only between batches
    s->keepHistory = on != 0;
    s->startBatch();
    return DCS_OK;
}

// `nTicks` ticks further (or fewer: once the machine has been quiescent for `stopWhenIdleFor` ticks in a row, 0 = never stop);
// every tick appends one frame job to the pending plan.  Bytes written to the data port since the last tick are handled first, as
// the sample pump does (DCSDecoder.cpp:1617).
void DcsSequencer::advance(uint32_t nTicks, uint32_t stopWhenIdleFor, uint32_t *ran)
{
    DcsSequencer *s = this;
    uint32_t done = 0;
    for (uint32_t t = 0 ; t < nTicks ; ++t)
    {
        if (stopWhenIdleFor != 0 && done != 0 && s->idleRun >= stopWhenIdleFor)
            break;
        // a stream still being walked is planned only as far as its records go: what there is now is decoded and handed out while
        // the walker goes on (a call that has nothing yet waits for the next record; ticks run again by a rewind had theirs)
        if (!s->replaying && s->plannableTicks() == 0)
        {
            if (done != 0)
                break;
            // (a record comes every half microsecond: look again at once; the core is given up only when the walker seems not to
            // be running at all -- it may be waiting for this very core)
            for (unsigned spins = 0 ; s->plannableTicks() == 0 ; ++spins)
            {
                if (spins < 20000)
                    __builtin_ia32_pause();
                else
                    std::this_thread::yield();
            }
        }
        if (s->fatal)
        {
            // DecoderFatalError: silence from here on (DCSDecoder.cpp:1672-1675)
            DcsFrameJob jb;
            memset(&jb, 0, sizeof(jb));
            jb.volShift = 8;
            jb.xform = (s->os == DCS_OS93A || s->os == DCS_OS93B) ? DCS_XFORM_93 : DCS_XFORM_94;
            jb.prev = DCS_PREV_NONE;
            jb.flags = 0;
            s->jobs.push_back(jb);
            ++s->tick;
            ++done;
            s->tickDone(true);
            continue;
        }
        const bool wasQuiet = s->quiescent();
        while (!s->dataPortQueue.empty())
        {
            s->lastDataPortByte = s->dataPortQueue.front();
            s->dataPortQueue.pop_front();
            s->portByte(s->lastDataPortByte);
        }
        const size_t jobsBefore = s->jobs.size(), srcsBefore = s->srcs.size();
        for (int retries = 0 ; ; )
        {
            try
            {
                s->mainLoop();
                break;
            }
            catch (const ResetException &)
            {
                // MainLoop is simply entered again, state as the failed pass left it (DCSDecoder.cpp:1624-1648)
                s->jobs.resize(jobsBefore);
                s->srcs.resize(srcsBefore);
                if (++retries > 3)
                {
                    s->fatal = true;
                    s->fatalTick = s->tick;
                    s->lastError = "the decoder reset itself after repeated fatal errors in the track data";
                    break;
                }
            }
        }
        if (s->fatal)
        {
            --t;                            // this tick is produced by the fatal branch above
            continue;
        }
        ++s->tick;
        ++done;
        s->tickDone(wasQuiet);
    }
    if (ran != nullptr)
        *ran = done;
}

extern "C" DcsStatus dcs_seq_rewind(DcsSequencer *s, uint32_t keepTicks)
{
    if (s == nullptr || !s->keepHistory || keepTicks > s->batchTicks())
        return DCS_ERR_INVALID_ARG;
    if (keepTicks == s->batchTicks() && !s->history.empty())
        return DCS_OK;                          // nothing behind it to drop
    const uint64_t firstTick = s->history[0].st.tick;
    // the last snapshot at or before the tick asked for, then the ticks in between once more
    size_t k = s->history.size() - 1;
    while (s->history[k].ticks > keepTicks)
        --k;
    s->history.resize(k + 1);
    const DcsSequencer::Snapshot &snap = s->history[k];
    static_cast<VmState &>(*s) = snap.st;
    const uint32_t from = snap.ticks;
    s->playingAfter.resize(from);
    s->tracksAfter.resize(from);
    while (!s->hostBytes.empty() && s->hostBytes.back().tick >= firstTick + keepTicks)
        s->hostBytes.pop_back();
    if (s->batchDecoded)
    {
        memcpy(s->tail, keepTicks == 0 ? s->batchTail0 : &s->batchTails[(keepTicks - 1) * 16], sizeof(s->tail));
        s->batchTails.resize(static_cast<size_t>(keepTicks) * 16);
    }
    else
    {
        s->jobs.resize(from);
        s->srcs.resize(snap.nSrcs);
    }
    if (keepTicks > from)
    {
        s->replaying = true;
        s->advance(keepTicks - from, 0, nullptr);
        s->replaying = false;
        if (s->batchDecoded)
        {
            s->jobs.clear();                    // (these frames have been decoded already)
            s->srcs.clear();
        }
    }
    return DCS_OK;
}

// Run the sequencer `nTicks` ticks further; every tick appends one frame job to the pending plan.
extern "C" DcsStatus dcs_seq_plan(DcsSequencer *s, uint32_t nTicks)
{
    if (s == nullptr) return DCS_ERR_INVALID_ARG;
    if (s->batchDecoded)
    {
        // a new batch starts: the previous one can no longer be rewound into
        s->startBatch();
        s->compact();
    }
    s->advance(nTicks, 0, nullptr);
    return DCS_OK;
}

// The same for a caller that decodes AHEAD of what it has been asked for (DCSDecoderHIP's sample pump): at most maxTicks, at least
// one, and no further than idleTicks ticks into silence -- once nothing plays, no program runs, no timer is set and nothing is
// queued, every further frame is digital silence until the next command, and a command takes the caller back (dcs_seq_rewind)
// anyway.  idleTicks = 2 covers the frame that carries the last overlap tail out and one of silence (DCSExplorer.cpp:1674).
extern "C" DcsStatus dcs_seq_plan_ahead(DcsSequencer *s, uint32_t maxTicks, uint32_t idleTicks, uint32_t *plannedOut)
{
    if (s == nullptr || maxTicks == 0) return DCS_ERR_INVALID_ARG;
    if (s->batchDecoded)
    {
        s->startBatch();
        s->compact();
    }
    uint32_t ran = 0;
    s->advance(maxTicks, idleTicks, &ran);
    if (plannedOut != nullptr)
        *plannedOut = ran;
    return DCS_OK;
}

extern "C" uint32_t dcs_seq_pending_ticks(const DcsSequencer *s) { return s != nullptr ? static_cast<uint32_t>(s->jobs.size()) : 0; }
extern "C" int dcs_seq_is_fatal(const DcsSequencer *s) { return s != nullptr && s->fatal ? 1 : 0; }
extern "C" uint64_t dcs_seq_tick(const DcsSequencer *s) { return s != nullptr ? s->tick : 0; }
extern "C" uint64_t dcs_seq_fatal_tick(const DcsSequencer *s) { return s != nullptr && s->fatal ? s->fatalTick : ~uint64_t(0); }
extern "C" int dcs_seq_stream_playing(const DcsSequencer *s, int channel)
{
    return s != nullptr && channel >= 0 && channel < DCS_MAX_CHANNELS && s->ch[channel].st != nullptr ? 1 : 0;
}

// ... as it was after the first `ticks` ticks of the current batch (planned or just decoded), without going back there
extern "C" int dcs_seq_stream_playing_at(const DcsSequencer *s, uint32_t ticks, int channel)
{
    if (s == nullptr || channel < 0 || channel >= DCS_MAX_CHANNELS || ticks > s->batchTicks() || s->history.empty())
        return 0;
    if (ticks == s->batchTicks())
        return s->ch[channel].st != nullptr ? 1 : 0;        // the machine as it stands, commands behind the last tick included
    if (ticks == 0)
        return s->history[0].st.ch[channel].st != nullptr ? 1 : 0;
    return (s->playingAfter[ticks - 1] >> channel) & 1;
}

// whether ClearTracks (:1466-1473: every channel's program and stream dropped) would have changed anything after the first `ticks`
// ticks of the current batch: 0 = no channel had a program or a stream then
extern "C" int dcs_seq_tracks_active_at(const DcsSequencer *s, uint32_t ticks)
{
    if (s == nullptr || ticks > s->batchTicks() || s->history.empty())
        return 1;
    if (ticks == s->batchTicks())
    {
        for (const Chan &c : s->ch)                          // the machine as it stands, commands behind the last tick included
            if (c.st != nullptr || !c.track.isNull())
                return 1;
        return 0;
    }
    if (ticks == 0)
    {
        for (const Chan &c : s->history[0].st.ch)
            if (c.st != nullptr || !c.track.isNull())
                return 1;
        return 0;
    }
    return (s->playingAfter[ticks - 1] | s->tracksAfter[ticks - 1]) != 0 ? 1 : 0;
}

extern "C" uint32_t dcs_seq_host_bytes(DcsSequencer *s, DcsHostByte *out, uint32_t cap)
{
    if (s == nullptr) return 0;
    const uint32_t n = static_cast<uint32_t>(s->hostBytes.size());
    if (out != nullptr && cap >= n)
    {
        if (n != 0)
            memcpy(out, s->hostBytes.data(), sizeof(DcsHostByte) * n);
        s->hostBytes.clear();
    }
    return n;
}

// Decode everything planned since the last call in ONE launch; the overlap tail carries over to the next call.  The PCM (and
// error words) stay where the context's live decoder put them -- pinned memory, valid until the context decodes again.
extern "C" DcsStatus dcs_seq_decode_view(DcsCtx *ctx, DcsSequencer *s, const int16_t **pcmOut, uint32_t *nFramesOut, const uint32_t **errOut)
{
    if (ctx == nullptr || s == nullptr || pcmOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    const size_t n = s->jobs.size();
    *pcmOut = nullptr;
    if (nFramesOut != nullptr)
        *nFramesOut = static_cast<uint32_t>(n);
    if (n == 0)
        return DCS_OK;
    if (s->blob.empty())
        s->blob.assign(16, 0);
    const int16_t *tails = nullptr;
    const DcsStatus st = dcs_decode_batch_live(ctx, s->blob.data(), s->blobStableLen(), s->blobId, s->srcs.empty() ? nullptr : s->srcs.data(),
                                               static_cast<uint32_t>(s->srcs.size()), s->jobs.data(), static_cast<uint32_t>(n),
                                               s->tail, 1, pcmOut, errOut, &tails);
    if (st != DCS_OK)
    {
        s->lastError = dcs_last_error(ctx);
        return st;
    }
    memcpy(s->batchTail0, s->tail, sizeof(s->tail));
    memcpy(s->tail, tails + (n - 1) * 16, sizeof(s->tail));
    s->batchTails.assign(tails, tails + n * 16);
    s->batchDecoded = true;
    s->jobs.clear();
    s->srcs.clear();
    // the blob and the stream cache stay: streams already copied are reused by later plans
    return DCS_OK;
}

extern "C" DcsStatus dcs_seq_decode(DcsCtx *ctx, DcsSequencer *s, int16_t *pcmOut, size_t pcmCapFrames, uint32_t *errOut)
{
    if (ctx == nullptr || s == nullptr || pcmOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    const size_t n = s->jobs.size();
    if (n == 0)
        return DCS_OK;
    if (n > pcmCapFrames)
        return DCS_ERR_CAPACITY;
    if (n > (1u << 17))
    {
        // more than the live decoder takes in one call (a whole ROM's tracks planned ahead, dcs_extract_tracks): the one-shot way
        if (s->blob.empty())
            s->blob.assign(16, 0);
        std::vector<int16_t> tails(n * 16);
        s->finishWalk();
        const DcsStatus st = dcs_decode_batch(ctx, s->blob.data(), s->blob.size(), s->srcs.empty() ? nullptr : s->srcs.data(),
                                              static_cast<uint32_t>(s->srcs.size()), s->jobs.data(), static_cast<uint32_t>(n),
                                              s->tail, 1, pcmOut, errOut, tails.data());
        if (st != DCS_OK)
        {
            s->lastError = dcs_last_error(ctx);
            return st;
        }
        memcpy(s->batchTail0, s->tail, sizeof(s->tail));
        memcpy(s->tail, &tails[(n - 1) * 16], sizeof(s->tail));
        s->batchTails.swap(tails);
        s->batchDecoded = true;
        s->jobs.clear();
        s->srcs.clear();
        return DCS_OK;
    }
    const int16_t *pcm = nullptr;
    const uint32_t *err = nullptr;
    const DcsStatus st = dcs_seq_decode_view(ctx, s, &pcm, nullptr, &err);
    if (st != DCS_OK)
        return st;
    memcpy(pcmOut, pcm, n * DCS_FRAME_SAMPLES * sizeof(int16_t));
    if (errOut != nullptr)
        memcpy(errOut, err, n * sizeof(uint32_t));
    return DCS_OK;
}

// The track loop of `DCSExplorer --extract-tracks` (DCSExplorer.cpp:1628-1721, :1905-1925) on one sequencer: all ticks of
// all tracks planned ahead, one launch.
extern "C" DcsStatus dcs_extract_tracks(DcsCtx *ctx, const DcsRomSet *rs, const DcsExtractTrack *items, uint32_t n,
                                        int16_t *pcmOut, size_t pcmCapFrames, uint32_t *frameOffsets, uint32_t *errOut)
{
    if (ctx == nullptr || rs == nullptr || (items == nullptr && n != 0) || pcmOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    uint64_t total = 0;
    for (uint32_t i = 0 ; i < n ; ++i)
        total += items[i].nFrames;
    if (total > pcmCapFrames || total > 0xFFFFFFFFull)
        return DCS_ERR_CAPACITY;
    DcsSequencer *seq = dcs_seq_create(rs);                 // (the state SoftBoot leaves)
    if (seq == nullptr)
        return DCS_ERR_BAD_STREAM;
    DcsStatus st = dcs_seq_set_master_volume(seq, 255);
    uint32_t at = 0;
    for (uint32_t i = 0 ; i < n && st == DCS_OK ; ++i)
    {
        if (frameOffsets != nullptr)
            frameOffsets[i] = at;
        st = dcs_seq_clear_tracks(seq);
        if (st == DCS_OK) st = dcs_seq_add_track_command(seq, static_cast<uint16_t>(items[i].track));
        const uint32_t nFrames = items[i].nFrames;
        // every frame but the last two in one go; ClearTracks() behind each of those (ExtractToWAV :1701-1712)
        const uint32_t plain = nFrames > 2 ? nFrames - 2 : 0;
        if (st == DCS_OK && plain != 0) st = dcs_seq_plan(seq, plain);
        for (uint32_t f = plain ; f < nFrames && st == DCS_OK ; ++f)
        {
            st = dcs_seq_plan(seq, 1);
            if (st == DCS_OK) st = dcs_seq_clear_tracks(seq);
        }
        at += nFrames;
    }
    if (frameOffsets != nullptr)
        frameOffsets[n] = at;
    if (st == DCS_OK && total != 0)
        st = dcs_seq_decode(ctx, seq, pcmOut, pcmCapFrames, errOut);
    dcs_seq_destroy(seq);
    return st;
}

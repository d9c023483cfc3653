// dcs_index.cpp -- host entry points of the index pass (the walker itself is dcs_scan.h, shared with
// the device index kernel).
#include "dcs_scan.h"
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

struct HostFetch
{
    const uint8_t *data;
    size_t len;
    uint32_t operator()(size_t i) const { return i < len ? data[i] : 0u; }
};

// The reader the host walk uses: a 64-bit window refilled four bytes at a time.  The reference reader's byte pointer
// (what StreamInfo.nBytes reports, DcsBits in dcs_scan.h reproduces it literally) is not kept but computed, as the
// device reader does: a Peek(n) at bit position B leaves that pointer at floor((B + n) / 8) + 1 bytes into the payload,
// so after the walk it stands at the maximum of that over every look.  Same records and the same nBytes as DcsBits
// (tests/test_host.py holds both against the oracle and the compiled reference), about twice as fast.
struct WinBits
{
    static constexpr bool kAnalytic = true;
    const uint8_t *data;
    size_t len;                 // bytes past it read as zero
    size_t payOff = 0;
    size_t next = 0;            // next byte to pull into the window
    uint64_t win = 0;           // unread bits, MSB first
    int have = 0;
    uint32_t pos = 0;           // payload bits consumed
    uint32_t hi = 0;            // max over looks of (pos + n)
    bool any = false;

    uint32_t byteAt(size_t i) const { return i < len ? data[i] : 0u; }
    void setPayload(size_t off) { payOff = next = off; win = 0; have = 0; pos = 0; hi = 0; any = false; }
    uint32_t load32(size_t i) const
    {
        if (i + 4 <= len)
        {
            uint32_t w;
            memcpy(&w, data + i, 4);
            return __builtin_bswap32(w);
        }
        return (byteAt(i) << 24) | (byteAt(i + 1) << 16) | (byteAt(i + 2) << 8) | byteAt(i + 3);
    }
    void refill()
    {
        win |= static_cast<uint64_t>(load32(next)) << (32 - have);
        next += 4;
        have += 32;
    }
    uint32_t peek(int n)
    {
        any = true;
        const uint32_t reach = pos + static_cast<uint32_t>(n);
        hi = reach > hi ? reach : hi;
        if (have < n)
            refill();
        return n == 0 ? 0u : static_cast<uint32_t>(win >> 32) >> (32 - n);
    }
    // the next n bits without counting as a look of the reference's reader (nBytes is unaffected): for the multi-code table
    uint32_t look(int n)
    {
        if (have < n)
            refill();
        return static_cast<uint32_t>(win >> 32) >> (32 - n);
    }
    void consume(int n)
    {
        // (after look(), which may leave fewer than n <= 32 valid bits only at the very end of the data, where zeros follow)
        win <<= n;
        have -= n;
        pos += static_cast<uint32_t>(n);
    }
    // n bits looked at with look() and consumed, counted like n looks of one bit each: the last reaches pos + n
    void took(int n)
    {
        consume(n);
        any = true;
        hi = pos > hi ? pos : hi;
    }
    uint32_t get(int n)
    {
        const uint32_t r = peek(n);
        consume(n);
        return r;
    }
    void skipRun(int count, int width)
    {
        if (count <= 0 || width <= 0)
            return;
        any = true;
        const uint32_t total = static_cast<uint32_t>(count) * static_cast<uint32_t>(width);
        pos += total;
        hi = pos > hi ? pos : hi;               // the last field's look reached exactly its own end
        if (total <= static_cast<uint32_t>(have))
        {
            win = total >= 64 ? 0 : win << total;
            have -= static_cast<int>(total);
            return;
        }
        const size_t bit = payOff * 8 + pos;
        next = bit >> 3;
        win = 0;
        have = 0;
        refill();
        const int skip = static_cast<int>(bit & 7);
        win <<= skip;
        have -= skip;
    }
    uint32_t bitPos() const { return pos; }
    size_t bytesFetched() const { return any ? payOff + (hi >> 3) + 1 : payOff; }
};

struct ArraySink
{
    DcsFrameIndex *out;
    uint32_t cap;
    uint32_t written = 0;
    bool overflow = false;
    void operator()(uint32_t f, const DcsFrameIndex &fi)
    {
        if (f < cap) { out[f] = fi; written = f + 1; }
        else overflow = true;
    }
};

}   // namespace

// the multi-code table of DcsScanTables, built once from the sample codebooks
static const uint16_t *multi94Table()
{
    static const std::vector<uint16_t> table = [] {
        const DcsLdsTables &T = dcsTables().lds;
        std::vector<uint16_t> t(static_cast<size_t>(6) << DCS_MULTI_BITS, 0);
        for (int code = 1 ; code <= 6 ; ++code)
        {
            const int maxBits = T.cbInfo[code] & 0xF;
            const uint16_t *book = T.cb94 + (T.cbInfo[code] >> 4);
            for (uint32_t v = 0 ; v < (1u << DCS_MULTI_BITS) ; ++v)
            {
                int at = 0, steps = 0;
                // a code is taken only if its whole look-ahead lies inside the DCS_MULTI_BITS bits
                while (at + maxBits <= DCS_MULTI_BITS && steps < 200)
                {
                    const uint32_t idx = (v >> (DCS_MULTI_BITS - at - maxBits)) & ((1u << maxBits) - 1);
                    const uint32_t e = book[idx];
                    at += static_cast<int>((e >> 8) & 0x1F);
                    steps += static_cast<int>(e >> 13) == 2 ? 2 : 1;
                }
                t[(static_cast<size_t>(code - 1) << DCS_MULTI_BITS) + v] = static_cast<uint16_t>(at | (steps << 8));
            }
        }
        return t;
    }();
    return table.data();
}

extern "C" DcsStatus dcs_index_stream(DcsOsVersion os, const uint8_t *stream, size_t len,
                                      DcsFrameIndex *out, uint32_t cap, DcsStreamInfo *info)
{
    if (stream == nullptr || len < 3 || os < DCS_OS93A || os > DCS_OS95 || (out == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;
    WinBits reader{ stream, len };
    const DcsScanTables tabs{ &dcsTables().lds, dcsTables().trie94, multi94Table(), dcsTables().fast94 };
    ArraySink sink{ out, cap };
    DcsScanMem mem;
    const DcsStreamInfo si = dcsScanStream(static_cast<int>(os), reader, tabs, &mem, sink);
    if (info != nullptr)
        *info = si;
    if (si.nFrames == 0)
        return DCS_ERR_BAD_STREAM;
    if (out == nullptr && cap == 0)
        return DCS_OK;
    return sink.overflow ? DCS_ERR_CAPACITY : DCS_OK;
}

// The same walk for a caller that uses the records as they come (the sequencer's background walker, dcs_sequencer.cpp): onFrame(f,
// record) is called for every frame, in order, on the calling thread.  Same records, same summary as dcs_index_stream.
namespace {
struct CallbackSink
{
    const std::function<void(uint32_t, const DcsFrameIndex &)> &fn;
    void operator()(uint32_t f, const DcsFrameIndex &fi) { fn(f, fi); }
};
}   // namespace
DcsStatus dcsIndexStreamProgressive(DcsOsVersion os, const uint8_t *stream, size_t len, DcsStreamInfo *info,
                                    const std::function<void(uint32_t, const DcsFrameIndex &)> &onFrame)
{
    if (stream == nullptr || len < 3 || os < DCS_OS93A || os > DCS_OS95)
        return DCS_ERR_INVALID_ARG;
    WinBits reader{ stream, len };
    const DcsScanTables tabs{ &dcsTables().lds, dcsTables().trie94, multi94Table(), dcsTables().fast94 };
    CallbackSink sink{ onFrame };
    DcsScanMem mem;
    const DcsStreamInfo si = dcsScanStream(static_cast<int>(os), reader, tabs, &mem, sink);
    if (info != nullptr)
        *info = si;
    return si.nFrames == 0 ? DCS_ERR_BAD_STREAM : DCS_OK;
}

// What a stream says about itself before any frame is walked: frame count, header, layout (the container part of the walk,
// InitChannelStream / InitStreamPlayback, DCSDecoderNative.cpp:1433-1463, :1595-1641).  nBytes, nValidFrames and payloadBits are 0.
DcsStatus dcsStreamContainer(DcsOsVersion os, const uint8_t *stream, size_t len, DcsStreamInfo *info)
{
    if (stream == nullptr || len < 3 || os < DCS_OS93A || os > DCS_OS95 || info == nullptr)
        return DCS_ERR_INVALID_ARG;
    auto byteAt = [&](size_t i) -> uint32_t { return i < len ? stream[i] : 0u; };
    DcsStreamInfo si = {};
    si.nFrames = static_cast<int32_t>((byteAt(0) << 8) | byteAt(1));
    const bool typeBit = (byteAt(2) & 0x80) != 0;
    si.hdrLen = (os == DCS_OS93A && typeBit) ? 1 : 16;
    for (int i = 0 ; i < 16 ; ++i)
        si.header[i] = i < si.hdrLen ? static_cast<uint8_t>(byteAt(2 + static_cast<size_t>(i))) : static_cast<uint8_t>(0);
    si.formatType = typeBit ? 1 : 0;
    if (os == DCS_OS94 || os == DCS_OS95)
        si.formatSubType = ((si.header[1] & 0x80) >> 6) | ((si.header[1] & 0x80) >> 7);
    if (os == DCS_OS93A)
        si.format = typeBit ? DCS_FMT_93A_T1 : DCS_FMT_93_T0;
    else if (os == DCS_OS93B)
        si.format = typeBit ? DCS_FMT_93B_T1 : DCS_FMT_93_T0;
    else if (!typeBit)
        si.format = DCS_FMT_94_T0;
    else
        si.format = (((si.header[1] | si.header[2]) & 0x80) == 0) ? DCS_FMT_94_T1_S0 : DCS_FMT_94_T1_S3;
    *info = si;
    return si.nFrames == 0 ? DCS_ERR_BAD_STREAM : DCS_OK;
}

// Diagnostic: the same walk with the reader that keeps the reference's byte pointer LITERALLY (DcsBits, dcs_scan.h:
// Peek pulls whole bytes while nBits <= n) and without the multi-code table.  The tests hold dcs_index_stream against it:
// same records, same nBytes.
extern "C" DcsStatus dcs_index_stream_literal(DcsOsVersion os, const uint8_t *stream, size_t len,
                                              DcsFrameIndex *out, uint32_t cap, DcsStreamInfo *info)
{
    if (stream == nullptr || len < 3 || os < DCS_OS93A || os > DCS_OS95 || (out == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;
    DcsBits<HostFetch> reader{ HostFetch{ stream, len } };
    const DcsScanTables tabs{ &dcsTables().lds, dcsTables().trie94, nullptr };
    ArraySink sink{ out, cap };
    DcsScanMem mem;
    const DcsStreamInfo si = dcsScanStream(static_cast<int>(os), reader, tabs, &mem, sink);
    if (info != nullptr)
        *info = si;
    if (si.nFrames == 0)
        return DCS_ERR_BAD_STREAM;
    return sink.overflow ? DCS_ERR_CAPACITY : DCS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Many streams at once.  The worker threads are created on first use and kept (spawning a thread costs
// about as much as indexing a short stream); the pool object is deliberately never destroyed, so nothing
// joins or tears down at process exit.  Streams are handed out through an atomic counter.
// ---------------------------------------------------------------------------------------------------------
namespace {

class IndexPool
{
public:
    static IndexPool &get()
    {
        static IndexPool *pool = new IndexPool;
        return *pool;
    }

    // run fn(k) for k in [0, n) on up to `threads` threads (the caller is one of them)
    void run(uint32_t n, int threads, const std::function<void(uint32_t)> &fn)
    {
        struct Count { std::atomic<int> &c; explicit Count(std::atomic<int> &c) : c(c) { ++c; } ~Count() { --c; } } count(regions);
        std::lock_guard<std::mutex> serial(runMutex);           // one parallel region at a time
        const int helpers = threads - 1;
        grow(helpers);
        {
            std::lock_guard<std::mutex> lk(m);
            job = &fn;
            total = n;
            next.store(0);
            wanted = helpers;
            running = helpers;
            ++generation;
        }
        cv.notify_all();
        drain(fn, n);
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return running == 0; });
        job = nullptr;
    }

    // parallel regions running or waiting for their turn right now (a caller that would only queue up behind them can do without)
    int busy() const { return regions.load(); }

private:
    std::atomic<int> regions{ 0 };
    void drain(const std::function<void(uint32_t)> &fn, uint32_t n)
    {
        for (uint32_t k = next.fetch_add(1) ; k < n ; k = next.fetch_add(1))
            fn(k);
    }
    void grow(int helpers)
    {
        std::lock_guard<std::mutex> lk(m);
        while (static_cast<int>(nWorkers) < helpers)
        {
            const int id = static_cast<int>(nWorkers++);
            std::thread([this, id] { worker(id); }).detach();
        }
    }
    void worker(int id)
    {
        uint64_t seen = 0;
        for (;;)
        {
            const std::function<void(uint32_t)> *fn;
            uint32_t n;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (id >= wanted)
                    continue;               // this region uses fewer threads
                fn = job;
                n = total;
            }
            drain(*fn, n);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0)
                    done.notify_one();
            }
        }
    }

    std::mutex runMutex, m;
    std::condition_variable cv, done;
    const std::function<void(uint32_t)> *job = nullptr;
    std::atomic<uint32_t> next{ 0 };
    uint32_t total = 0;
    uint64_t generation = 0;
    int wanted = 0, running = 0;
    size_t nWorkers = 0;
};

}   // namespace

// whether the process-wide index pool is serving (or queueing) a parallel region: several contexts decoding large lists at once
// (dcs_node, dcs_decode_streams_sharded: a thread per device) would otherwise walk their lists one after the other on it
bool dcsIndexPoolBusy() { return IndexPool::get().busy() > 0; }

// Host threads this process may actually run at once: the CPUs of its affinity mask, further limited by a cgroup
// CPU quota when there is one (a container given 16 CPUs of a 256-thread host reports 256 hardware threads).
static long long readNumber(const char *path, bool *isMax = nullptr)
{
    FILE *f = fopen(path, "r");
    if (f == nullptr)
        return -1;
    char word[32] = { 0 };
    const int got = fscanf(f, "%31s", word);
    fclose(f);
    if (got != 1)
        return -1;
    if (isMax != nullptr)
        *isMax = word[0] == 'm';                                // cgroup v2 writes "max" for "no quota"
    return word[0] == 'm' ? -1 : atoll(word);
}

extern "C" int dcs_host_threads(void)
{
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        n = CPU_COUNT(&set);
    if (n <= 0)
        n = static_cast<int>(std::thread::hardware_concurrency());
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r"))        // cgroup v2: "<quota|max> <period>"
    {
        char word[32] = { 0 };
        if (fscanf(f, "%31s %lld", word, &period) == 2 && word[0] != 'm')
            quota = atoll(word);
        fclose(f);
    }
    else                                                        // cgroup v1
    {
        quota = readNumber("/sys/fs/cgroup/cpu/cpu.cfs_quota_us");
        period = readNumber("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    }
    if (quota > 0 && period > 0)
    {
        const long long q = (quota + period - 1) / period;
        if (q >= 1 && q < n)
            n = static_cast<int>(q);
    }
    return n < 1 ? 1 : n;
}

// Stream k's records go to out + firstRecord[k]; it writes at most nFrames(k) of them (the U16 prefix
// of the stream).  nThreads 0 = dcs_host_threads(), at most 64 and at most one per stream.
// `done(k)`, when given, is called (on whichever pool thread indexed it) as soon as stream k's records are complete
DcsStatus dcsIndexStreamsNotify(const DcsStreamRef *streams, uint32_t nStreams, int nThreads,
                                DcsFrameIndex *out, const uint64_t *firstRecord, DcsStreamInfo *infos,
                                const std::function<void(uint32_t)> *done)
{
    if (streams == nullptr || out == nullptr || firstRecord == nullptr || infos == nullptr)
        return DCS_ERR_INVALID_ARG;
    if (nStreams == 0)
        return DCS_OK;
    if (nThreads <= 0)
    {
        nThreads = dcs_host_threads();
        if (nThreads > 64) nThreads = 64;
    }
    if (nThreads < 1) nThreads = 1;
    if (nThreads > 256) nThreads = 256;
    if (static_cast<uint32_t>(nThreads) > nStreams) nThreads = static_cast<int>(nStreams);
    std::atomic<int> firstError{ DCS_OK };
    const std::function<void(uint32_t)> one = [&](uint32_t k) {
        const DcsStreamRef &sr = streams[k];
        DcsStatus st = DCS_ERR_INVALID_ARG;
        if (sr.data != nullptr && sr.len >= 3)
        {
            const uint32_t nf = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
            st = dcs_index_stream(static_cast<DcsOsVersion>(sr.os), sr.data, sr.len, out + firstRecord[k], nf, &infos[k]);
            if (st == DCS_ERR_BAD_STREAM)
                st = DCS_OK;                // reported through infos[k].nFrames == 0
        }
        if (st != DCS_OK)
        {
            int expected = DCS_OK;
            firstError.compare_exchange_strong(expected, st);
        }
        if (done != nullptr)
            (*done)(k);
    };
    if (nThreads == 1)
        for (uint32_t k = 0 ; k < nStreams ; ++k)
            one(k);
    else
        IndexPool::get().run(nStreams, nThreads, one);
    return static_cast<DcsStatus>(firstError.load());
}

extern "C" DcsStatus dcs_index_streams(const DcsStreamRef *streams, uint32_t nStreams, int nThreads,
                                       DcsFrameIndex *out, const uint64_t *firstRecord, DcsStreamInfo *infos)
{
    return dcsIndexStreamsNotify(streams, nStreams, nThreads, out, firstRecord, infos, nullptr);
}

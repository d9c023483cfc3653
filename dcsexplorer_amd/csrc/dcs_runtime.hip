// dcs_runtime.hip -- context, resident batches and kernel launches of libdcs_hip.so (C ABI of
// include/dcs_hip.h).  The product path has no CPU decode: without a usable gfx950 device
// dcs_ctx_create fails and nothing below can run.
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <new>
#include <chrono>
#include <algorithm>
#include <mutex>
#include <atomic>
#include <unordered_set>
#include <unordered_map>
#include "dcs_common.h"
#include "dcs_kernels.hip.h"
#include "dcs_scan.h"
#include "dcs_index_wave.hip.h"

struct DcsCtx
{
    int device = 0;
    hipStream_t stream = nullptr;
    DcsDevTables *dTables = nullptr;
    int fpwOverride = 0;
    bool handoff = true;                // tails cross chunk boundaries through the hand-off buffer (else: halo re-decode)
    uint32_t shuffleSeed = 0;           // test hook: host-planned batches get their chunks in a seeded random order (dcs_ctx_set_test_hooks)
    bool noXcdRanges = false;           // test hook: no batch of this context is launched in XCD ranges
    int framesPerChunk = 0;             // diagnostic: frames a wavefront decodes (0 = as many as the kernel variant has slots)
    bool xcdRanges = false;             // batches of this context: chain order, launched in XCD ranges (dcs_ctx_set_concurrent_batches)
    bool keepAllTails = false;          // resident batches store EVERY frame's tail (dcs_ctx_set_batch_tails); default: the last frame of every chain
    bool largeListOnDevice = true;      // dcs_decode_streams on a large list: index walk, planner and packer on the device (dcs_ctx_set_large_list_path)
    bool largeListShared = true;        // ... with the host pool walking the first parts of the list next to the device (mode 2, the default)
    uint32_t sharedProbe = 0;           // calls made with a host share of nothing (every sixteenth tries one part again)
    int sharedHostParts = 7;            // how many of the eight parts the host walks (where 16 pool threads settle); follows the measured finish times from call to call
    std::mutex cacheMutex;              // the buffer cache is shared by the pipeline's worker threads
    struct DcsPipeline *internalPipe = nullptr;     // dcs_decode_streams takes large lists through it in parts (dcs_pipeline.hip.h)
    struct DcsLive *live = nullptr;     // the context's persistent small-batch decoder (dcs_decode_batch_live)
    int numCUs = 256;
    std::string lastError;              // written through setError only (the pipeline's threads fail concurrently)
    std::mutex errMutex;
    // inputs of the last dcs_index_streams_gpu call, resident for dcs_index_streams_gpu_time
    uint32_t *dIdxBlob = nullptr;
    DcsStreamLoc *dIdxLocs = nullptr;
    DcsFrameIndex *dIdxOut = nullptr;
    DcsStreamInfo *dIdxInfos = nullptr;
    size_t idxBlobLen = 0, idxBlobDw = 0;
    uint32_t idxStreams = 0;
    uint64_t idxCap = 0;
    // device and pinned-host buffers of destroyed batches, kept for the next batch (hipMalloc / hipFree cost
    // about as much as decoding a few thousand frames)
    struct Cached { void *p; size_t cap; };
    std::vector<Cached> devCache, pinCache;
    size_t cachedBytes = 0, cachedPinBytes = 0;
    // what the cache may keep (set at dcs_ctx_create from the card's free memory and the host's RAM; dcs_ctx_set_cache_limits)
    size_t devCacheLimit = size_t(4) << 30, pinCacheLimit = size_t(1) << 30;
    // real size of every buffer handed out by cacheAlloc (a reused buffer may be up to twice what was asked for; the
    // callers only remember what they asked for, and the cache must account for what it really holds)
    std::unordered_map<void *, size_t> liveCap;
};

// Bytes kept per context.  Device: a list in flight holds ~80 MB and a pipeline keeps up to 64 of them, and giving a
// buffer back with hipFree waits for the WHOLE device (milliseconds while an index round runs): the card has 288 GB.
// The ceilings are at most these, and less on a small card or host: an eighth of the card's free memory and a sixteenth
// of the host's RAM at dcs_ctx_create (eight ranks of a node then pin at most half of it between them).
static const size_t kDevCacheCeiling = size_t(32) << 30, kPinCacheCeiling = size_t(8) << 30;

// give everything the cache holds of one kind back to the runtime (largest first, until `want` bytes could be had or all
// of it when want == 0); returns the bytes released.  Caller does NOT hold the mutex.
static size_t cacheTrim(DcsCtx *ctx, bool pinned, size_t want)
{
    std::vector<DcsCtx::Cached> victims;
    {
        std::lock_guard<std::mutex> lock(ctx->cacheMutex);
        std::vector<DcsCtx::Cached> &c = pinned ? ctx->pinCache : ctx->devCache;
        // victims by size, picked through an index copy: the vector itself stays in the order the buffers came back
        // (cacheFree drops from its FRONT, the oldest, when a buffer returns to a full cache -- ADVICE r4)
        std::vector<size_t> bySize(c.size());
        for (size_t i = 0 ; i < c.size() ; ++i)
            bySize[i] = i;
        std::sort(bySize.begin(), bySize.end(), [&c](size_t a, size_t b) { return c[a].cap > c[b].cap; });
        size_t got = 0, n = 0;
        std::vector<char> gone(c.size(), 0);
        while (n < bySize.size() && (want == 0 || got < want))
        {
            got += c[bySize[n]].cap;
            victims.push_back(c[bySize[n]]);
            gone[bySize[n++]] = 1;
        }
        size_t keep = 0;
        for (size_t i = 0 ; i < c.size() ; ++i)
            if (!gone[i])
                c[keep++] = c[i];
        c.resize(keep);
        (pinned ? ctx->cachedPinBytes : ctx->cachedBytes) -= got;
    }
    size_t released = 0;
    for (const DcsCtx::Cached &v : victims)
    {
        if (pinned) (void)hipHostFree(v.p); else (void)hipFree(v.p);
        released += v.cap;
    }
    return released;
}

static hipError_t cacheAlloc(DcsCtx *ctx, bool pinned, void **out, size_t bytes)
{
    std::unique_lock<std::mutex> lock(ctx->cacheMutex);
    std::vector<DcsCtx::Cached> &c = pinned ? ctx->pinCache : ctx->devCache;
    size_t best = c.size();
    for (size_t i = 0 ; i < c.size() ; ++i)
        if (c[i].cap >= bytes && c[i].cap <= bytes * 2 + 4096 && (best == c.size() || c[i].cap < c[best].cap))
            best = i;
    if (best != c.size())
    {
        *out = c[best].p;
        (pinned ? ctx->cachedPinBytes : ctx->cachedBytes) -= c[best].cap;
        ctx->liveCap[c[best].p] = c[best].cap;
        c.erase(c.begin() + static_cast<long>(best));
        return hipSuccess;
    }
    lock.unlock();
    hipError_t e = pinned ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
    if (e != hipSuccess)
    {
        // out of memory while the cache sits on buffers of other sizes (reuse needs cap <= 2 x bytes): give those back,
        // largest first, and ask again; only when nothing is left to give does the error go to the caller
        (void)hipGetLastError();
        while (e != hipSuccess && cacheTrim(ctx, pinned, bytes) != 0)
        {
            e = pinned ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
            if (e != hipSuccess)
                (void)hipGetLastError();
        }
    }
    if (e == hipSuccess)
    {
        lock.lock();
        ctx->liveCap[*out] = bytes;
    }
    return e;
}

static void cacheFree(DcsCtx *ctx, bool pinned, void *p, size_t cap)
{
    if (p == nullptr)
        return;
    // A buffer that comes back is the one most likely to be asked for again (the next list of the same size): when the cache
    // is full it is the OLDEST cached buffers that go, not this one.  (Round 4: a context that had held one batch of 17 M frames
    // kept its 7 GB buffers and gave every 291 MB buffer of the lists that followed back to the runtime -- a hipHostFree and a
    // hipHostMalloc per list, 0.34 s for a job that takes 0.22.)
    std::vector<DcsCtx::Cached> victims;
    bool keep = true;
    {
        std::lock_guard<std::mutex> lock(ctx->cacheMutex);
        const auto live = ctx->liveCap.find(p);
        if (live != ctx->liveCap.end())
        {
            cap = live->second;             // what the buffer really holds, not what its last user asked for
            ctx->liveCap.erase(live);
        }
        size_t &held = pinned ? ctx->cachedPinBytes : ctx->cachedBytes;
        const size_t limit = pinned ? ctx->pinCacheLimit : ctx->devCacheLimit;
        std::vector<DcsCtx::Cached> &c = pinned ? ctx->pinCache : ctx->devCache;
        if (cap > limit)
            keep = false;
        else
        {
            size_t n = 0;                   // (buffers are appended as they come back: the front is the oldest)
            while (held + cap > limit && n < c.size())
                held -= c[n++].cap;
            victims.assign(c.begin(), c.begin() + static_cast<long>(n));
            c.erase(c.begin(), c.begin() + static_cast<long>(n));
            c.push_back(DcsCtx::Cached{ p, cap });
            held += cap;
        }
    }
    if (!keep)
        victims.push_back(DcsCtx::Cached{ p, cap });
    for (const DcsCtx::Cached &v : victims)
        if (pinned) (void)hipHostFree(v.p); else (void)hipFree(v.p);
}

static double hipchkNow() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Host waits of THIS thread leave the core to others: set by the pipeline's threads, of which dozens wait at any time while
// a few prepare lists.  The runtime's own waits do not do that on this stack -- hipStreamSynchronize and hipEventSynchronize
// burn the waiting thread's core for the whole wait, with an event created hipEventBlockingSync as well (tools/wait_cost.hip:
// 8.00 ms of thread CPU time for a wait of 8.01 ms) -- so such a thread records an event and asks for it between sleeps: a
// short look first (the kernels of a small list are done in microseconds), then naps of 50 us, which the kernel's timer slack
// makes ~110.  DCS_WAIT_RUNTIME=1 restores hipEventSynchronize.  Everybody else polls inside the runtime (shortest latency).
static thread_local bool tlsBlockingWaits = false;
// Batches made by THIS thread keep their chunks in chain order and are launched in XCD ranges (DCS_BATCH_XCD_RANGES): set by the
// pipelines' workers, whose decode kernels run next to each other's.
static thread_local bool tlsXcdRanges = false;
static thread_local bool tlsResidentBatch = false;   // dcs_batch_create: a batch that is run many times is planned twice for the shortest packages (dcsPlanChunksCapped)
static thread_local bool tlsKeepAllTails = false;   // dcs_decode_batch with a tailsOut array: every frame's tail (the sequencer resumes from any tick)

// DCS_WAIT_STATS=1: what the host's waits cost, summed over all threads, printed when a pipeline is destroyed
static std::atomic<uint64_t> g_waitCalls{0}, g_waitPolls{0}, g_waitCpuNs{0}, g_waitWallNs{0};
static const bool g_waitStats = getenv("DCS_WAIT_STATS") != nullptr && atoi(getenv("DCS_WAIT_STATS")) != 0;
static uint64_t threadCpuNs()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return static_cast<uint64_t>(ts.tv_sec) * 1000000000ull + static_cast<uint64_t>(ts.tv_nsec);
}
static void printWaitStats(const char *who)
{
    if (g_waitStats)
        fprintf(stderr, "wait stats (%s): %llu waits, %llu polls, %.1f ms of thread CPU, %.1f ms of wall time in them\n", who,
                static_cast<unsigned long long>(g_waitCalls.load()), static_cast<unsigned long long>(g_waitPolls.load()),
                g_waitCpuNs.load() * 1e-6, g_waitWallNs.load() * 1e-6);
}

// wait for everything enqueued on `stream`
static hipError_t streamWait(DcsCtx *ctx, hipStream_t stream)
{
    if (!tlsBlockingWaits)
        return hipStreamSynchronize(stream);
    // one event per thread and device, destroyed when the thread ends (pipeline workers and indexers come and go with
    // their pipeline)
    struct ThreadEvent
    {
        hipEvent_t ev = nullptr;
        int device = -1;
        ~ThreadEvent() { if (ev != nullptr) (void)hipEventDestroy(ev); }
    };
    thread_local ThreadEvent te;
    if (te.ev == nullptr || te.device != ctx->device)
    {
        if (te.ev != nullptr) { (void)hipEventDestroy(te.ev); te.ev = nullptr; }
        const hipError_t e = hipEventCreateWithFlags(&te.ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess)
            return e;
        te.device = ctx->device;
    }
    hipError_t e = hipEventRecord(te.ev, stream);
    if (e != hipSuccess)
        return e;
    static const bool runtimeWait = getenv("DCS_WAIT_RUNTIME") != nullptr && atoi(getenv("DCS_WAIT_RUNTIME")) != 0;
    if (runtimeWait)
        return hipEventSynchronize(te.ev);
    const double t0 = hipchkNow();
    const uint64_t c0 = g_waitStats ? threadCpuNs() : 0;
    uint64_t polls = 0;
    for (;;)
    {
        e = hipEventQuery(te.ev);
        ++polls;
        if (e != hipErrorNotReady)
        {
            if (g_waitStats)
            {
                g_waitCalls += 1; g_waitPolls += polls; g_waitCpuNs += threadCpuNs() - c0;
                g_waitWallNs += static_cast<uint64_t>((hipchkNow() - t0) * 1000.0);
            }
            return e;
        }
        if (hipchkNow() - t0 > 30.0)
        {
            static const long napNs = getenv("DCS_WAIT_NAP_US") != nullptr ? std::max(1, atoi(getenv("DCS_WAIT_NAP_US"))) * 1000L : 50000L;
            timespec nap{ 0, napNs };
            nanosleep(&nap, nullptr);
        }
    }
}

// A copy between pinned host memory and device memory done by a kernel on `stream` (both sides are device-visible
// addresses; 4-byte granularity).  The pipeline's small uploads go this way: hipMemcpyAsync from pinned memory now and then
// holds the CALLING thread for ~7 ms -- several threads at once, released together -- which a launch has not been seen to
// do (DCS_HIP_SLOW; profiles/NOTES.md item 17).  The PCM's way down stays with the copy engines.
namespace {
__global__ __launch_bounds__(256) void dcsCopyKernel(uint32_t *dst, const uint32_t *src, size_t nDw)
{
    const size_t n4 = nDw / 4, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0)
    {
        for (size_t i = t ; i < n4 ; i += stride)
            reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (size_t i = n4 * 4 + t ; i < nDw ; i += stride)
            dst[i] = src[i];
    }
    else
        for (size_t i = t ; i < nDw ; i += stride)
            dst[i] = src[i];
}
}   // namespace
static hipError_t copyByKernel(hipStream_t stream, void *dst, const void *src, size_t bytes, unsigned blockCap = 1024)
{
    if (bytes == 0)
        return hipSuccess;
    if ((bytes & 3u) != 0 || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3u) != 0)
        return hipErrorInvalidValue;
    const size_t nDw = bytes / 4;
    const unsigned blocks = static_cast<unsigned>(std::min<size_t>((nDw / 4 + 255) / 256 + 1, blockCap));
    hipLaunchKernelGGL(dcsCopyKernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint32_t *>(dst), static_cast<const uint32_t *>(src), nDw);
    return hipGetLastError();
}

struct DcsBatch
{
    DcsCtx *ctx = nullptr;
    uint32_t nJobs = 0, nSrcs = 0, nTailsIn = 0;
    size_t blobLen = 0;
    size_t blobOnDevice = 0;                // bytes of the blob that were uploaded (0 when no frame has a second source)
    int fpw = 0;
    uint32_t nChunks = 0;
    uint64_t algoBytes = 0;                 // payload + stream headers + 56 B per source + 480 B per frame
    uint64_t abiBytes = 0;                  // the same with the records as the ABI has them (160 B per source, 16 per job)
    hipStream_t stream = nullptr;           // uploads, default launches and downloads (the context's, or the batch's own)
    // device buffers
    uint8_t *dBlob = nullptr;
    DcsSrcDesc *dSrcs = nullptr;
    int16_t *dTailsIn = nullptr;
    int16_t *dPcm = nullptr;
    uint32_t *dErr = nullptr;
    int16_t *dTailsOut = nullptr;
    unsigned long long *dDebug = nullptr;   // DCS_STAMPS builds only
    unsigned long long *dHandoff = nullptr; // nChunks x 16 words (DcsKernelArgs.handoff)
    uint8_t *dPackages = nullptr;           // nChunks x dcsPkgBytes(fpw) (DcsKernelArgs.packages)
    uint32_t epoch = 0;                     // launches of this batch so far
    uint32_t flags = 0;                     // DCS_BATCH_*
    uint32_t imgDw = 0;                     // the packages' LAYOUT word (dcs_common.h): dwords of pool image | DCS_PKG_SPLIT4; packages at dcsPkgStride(fpw, imgDw)
    size_t cap[10] = { 0 };                 // allocated bytes of the buffers above, in that order
    // packages assembled on the device: the plan and the source digests as uploaded for the pack kernel
    void *dPlanSlots = nullptr, *dPlanSrcs = nullptr, *hStage = nullptr, *dTable = nullptr;      // (dTable: the stream table of the device planner; cap[2])
    size_t planSlotsCap = 0, planSrcsCap = 0, hStageCap = 0;
    // pinned host mirror of (pcm, err), filled by dcs_batch_download_view
    int16_t *hPcm = nullptr;
    uint32_t *hErr = nullptr;
    size_t hCap[2] = { 0 };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // recorded behind the last launch on whatever stream the caller launched on (a caller's stream is not ordered
    // against the context's non-blocking one): sync, download, download_view and destroy wait for it
    hipEvent_t evDone = nullptr;
    bool launched = false;
    bool downByKernel = false;  // the PCM's way down by dcsCopyKernel instead of the runtime's copy (the pipelines)
    unsigned downBlocks = 1024; // ... with at most this many workgroups
    bool settled = false;       // a wait has covered everything enqueued for this batch and nothing was enqueued since
    uint32_t planFpc = 0;       // frames per chunk of a plan made on the device (dcsPlanKernel)
    bool errJoined = false;     // dErr lies behind dPcm in ONE allocation (and hErr behind hPcm): the two come down in one copy
};

// the device work of the batch's last launch has finished (host-side wait)
static hipError_t waitLaunched(DcsBatch *b)
{
    return b->launched ? hipEventSynchronize(b->evDone) : hipSuccess;
}

static std::string g_createError;
static std::mutex g_createErrorMutex;
static void liveDestroy(DcsCtx *ctx);

static void setError(DcsCtx *ctx, const std::string &text)
{
    std::lock_guard<std::mutex> lock(ctx->errMutex);
    ctx->lastError = text;
}
static void setCreateError(const std::string &text)
{
    std::lock_guard<std::mutex> lock(g_createErrorMutex);
    g_createError = text;
}

// (DCS_HIP_SLOW=<microseconds>: report every runtime call that takes longer, a diagnostic for the pipeline's threads)
static const double g_hipSlowUs = getenv("DCS_HIP_SLOW") ? atof(getenv("DCS_HIP_SLOW")) : 0.0;
#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        const double t_ = g_hipSlowUs > 0 ? hipchkNow() : 0.0;                                   \
        hipError_t e_ = (call);                                                                  \
        if (g_hipSlowUs > 0 && hipchkNow() - t_ > g_hipSlowUs)                                   \
            fprintf(stderr, "slow hip call: %.0f us %s [from %.3f ms]\n", hipchkNow() - t_, #call, t_ * 1e-3);  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[256];                                                                      \
            snprintf(buf_, sizeof(buf_), "%s failed: %s", #call, hipGetErrorString(e_));         \
            setError((ctx), buf_);                                                               \
            return DCS_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and packets of
// different streams that share a queue run one behind the other.  A pipeline has a dozen streams with chains of short kernels
// and half-millisecond copies; with eight queues a list's chain stops waiting behind another list's copy (0.75 -> 0.70 ms per
// list sustained, round 4; sixteen queues showed stalls of seconds).  The variable is read when the runtime initialises.
// Loading this library changes NOTHING in the host's environment (round 5; a constructor used to): dcs_runtime_defaults() is an
// explicit call, and the library's own first calls into HIP (dcs_device_count, dcs_ctx_create) make it once unless
// DCS_NO_RUNTIME_DEFAULTS is set -- effective only if HIP has not been initialised by the host before, harmless otherwise, and a
// variable the user has set is never overwritten.  INTEGRATION.md, "Runtime settings".
extern "C" int dcs_runtime_defaults(void)
{
    if (getenv("GPU_MAX_HW_QUEUES") != nullptr)
        return 0;
    return setenv("GPU_MAX_HW_QUEUES", "8", 0) == 0 ? 1 : 0;
}

static void runtimeDefaultsOnce()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("DCS_NO_RUNTIME_DEFAULTS") == nullptr)
            (void)dcs_runtime_defaults();
    });
}

extern "C" uint32_t dcs_abi_version(void) { return DCS_ABI_VERSION; }
#ifndef DCS_BUILD_ID
#define DCS_BUILD_ID "unknown"
#endif
extern "C" const char *dcs_build_id(void) { return DCS_BUILD_ID; }

extern "C" int dcs_device_count(void)
{
    runtimeDefaultsOnce();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" const char *dcs_last_error(const DcsCtx *ctx)
{
    // a copy owned by the calling thread: the string itself may be rewritten by a pipeline thread at any time
    thread_local std::string copy;
    if (ctx != nullptr)
    {
        std::lock_guard<std::mutex> lock(const_cast<DcsCtx *>(ctx)->errMutex);
        copy = ctx->lastError;
    }
    else
    {
        std::lock_guard<std::mutex> lock(g_createErrorMutex);
        copy = g_createError;
    }
    return copy.c_str();
}

extern "C" DcsStatus dcs_ctx_create(int deviceId, DcsCtx **out)
{
    if (out == nullptr)
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;
    runtimeDefaultsOnce();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
    {
        setCreateError("no HIP device available (this library has no CPU fallback)");
        return DCS_ERR_NO_DEVICE;
    }
    if (deviceId < 0 || deviceId >= n)
    {
        setCreateError("device id out of range");
        return DCS_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, deviceId) != hipSuccess)
    {
        setCreateError("hipGetDeviceProperties failed");
        return DCS_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        setCreateError(std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        return DCS_ERR_NO_DEVICE;
    }

    DcsCtx *ctx = new (std::nothrow) DcsCtx;
    if (ctx == nullptr)
        return DCS_ERR_NO_MEMORY;
    ctx->device = deviceId;
    ctx->numCUs = prop.multiProcessorCount;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(deviceId));
        size_t freeB = 0, totalB = 0;
        HIPCHK(ctx, hipMemGetInfo(&freeB, &totalB));
        ctx->devCacheLimit = std::min(kDevCacheCeiling, freeB / 8);
        const long pages = sysconf(_SC_PHYS_PAGES), pageSize = sysconf(_SC_PAGE_SIZE);
        if (pages > 0 && pageSize > 0)
            ctx->pinCacheLimit = std::min(kPinCacheCeiling, static_cast<size_t>(pages) * static_cast<size_t>(pageSize) / 16);
        if (const char *mb = getenv("DCS_CACHE_DEV_MB")) ctx->devCacheLimit = static_cast<size_t>(atol(mb)) << 20;
        if (const char *mb = getenv("DCS_CACHE_PIN_MB")) ctx->pinCacheLimit = static_cast<size_t>(atol(mb)) << 20;
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dTables), sizeof(DcsDevTables)));
        HIPCHK(ctx, hipMemcpy(ctx->dTables, &dcsTables(), sizeof(DcsDevTables), hipMemcpyHostToDevice));
        // opt in to the LDS the largest configuration needs
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&dcsk::dcsDecodeKernel<16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, dcsk::ldsBytes(16)));
        return DCS_OK;
    }();
    if (st != DCS_OK)
    {
        setCreateError(dcs_last_error(ctx));
        dcs_ctx_destroy(ctx);
        return st;
    }
    *out = ctx;
    return DCS_OK;
}

extern "C" void dcs_ctx_destroy(DcsCtx *ctx)
{
    if (ctx == nullptr)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->internalPipe) dcs_pipeline_destroy(ctx->internalPipe);
    liveDestroy(ctx);
    if (ctx->dTables) (void)hipFree(ctx->dTables);
    if (ctx->dIdxBlob) (void)hipFree(ctx->dIdxBlob);
    if (ctx->dIdxLocs) (void)hipFree(ctx->dIdxLocs);
    if (ctx->dIdxOut) (void)hipFree(ctx->dIdxOut);
    if (ctx->dIdxInfos) (void)hipFree(ctx->dIdxInfos);
    for (const DcsCtx::Cached &c : ctx->devCache) (void)hipFree(c.p);
    for (const DcsCtx::Cached &c : ctx->pinCache) (void)hipHostFree(c.p);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" DcsStatus dcs_ctx_set_cache_limits(DcsCtx *ctx, uint64_t deviceBytes, uint64_t pinnedBytes)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> lock(ctx->cacheMutex);
        ctx->devCacheLimit = static_cast<size_t>(deviceBytes);
        ctx->pinCacheLimit = static_cast<size_t>(pinnedBytes);
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_trim_cache(DcsCtx *ctx, uint64_t *deviceBytesReleased, uint64_t *pinnedBytesReleased)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const size_t d = cacheTrim(ctx, false, 0), p = cacheTrim(ctx, true, 0);
    if (deviceBytesReleased) *deviceBytesReleased = d;
    if (pinnedBytesReleased) *pinnedBytesReleased = p;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_cache_bytes(DcsCtx *ctx, uint64_t *deviceBytes, uint64_t *pinnedBytes, uint64_t *deviceLimit, uint64_t *pinnedLimit)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->cacheMutex);
    if (deviceBytes) *deviceBytes = ctx->cachedBytes;
    if (pinnedBytes) *pinnedBytes = ctx->cachedPinBytes;
    if (deviceLimit) *deviceLimit = ctx->devCacheLimit;
    if (pinnedLimit) *pinnedLimit = ctx->pinCacheLimit;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_set_concurrent_batches(DcsCtx *ctx, int enable)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    ctx->xcdRanges = enable != 0;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_set_frames_per_wave(DcsCtx *ctx, int fpw)
{
    if (ctx == nullptr || !(fpw == 0 || fpw == 4 || fpw == 8 || fpw == 16))
        return DCS_ERR_INVALID_ARG;
    ctx->fpwOverride = fpw;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_set_frames_per_chunk(DcsCtx *ctx, int frames)
{
    if (ctx == nullptr || frames < 0 || frames > 16)
        return DCS_ERR_INVALID_ARG;
    ctx->framesPerChunk = frames;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_set_tail_handoff(DcsCtx *ctx, int enable)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    ctx->handoff = enable != 0;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_set_large_list_path(DcsCtx *ctx, int mode)
{
    if (ctx == nullptr || mode < 0 || mode > 2)
        return DCS_ERR_INVALID_ARG;
    const bool onDevice = mode != 0;
    if (ctx->largeListOnDevice != onDevice && ctx->internalPipe != nullptr)
    {
        dcs_pipeline_destroy(ctx->internalPipe);        // (made for the other path; the next large list makes its own)
        ctx->internalPipe = nullptr;
    }
    ctx->largeListOnDevice = onDevice;
    ctx->largeListShared = mode == 2;
    return DCS_OK;
}

extern "C" DcsStatus dcs_ctx_set_batch_tails(DcsCtx *ctx, int allFrames)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    ctx->keepAllTails = allFrames != 0;
    return DCS_OK;
}

extern "C" uint32_t dcs_batch_package_bytes(const DcsBatch *b)
{
    return b ? dcsPkgStride(b->fpw, dcsPkgImgDw(b->imgDw) != 0 ? b->imgDw : (b->imgDw | dcsPoolCapacity(b->fpw))) : 0;
}

extern "C" DcsStatus dcs_ctx_set_test_hooks(DcsCtx *ctx, uint32_t chunkOrderSeed, int noXcdRanges)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    ctx->shuffleSeed = chunkOrderSeed;
    ctx->noXcdRanges = noXcdRanges != 0;
    return DCS_OK;
}

// Frames per wavefront, from measurement (bench.py --fpw over batch sizes, rounds 2 and 3): 4 (16 lanes unpack a frame: the
// shortest serial path per wavefront) while the batch is small, 8 (four wavefronts per SIMD, full rounds) beyond.  The
// crossover depends on what is decoded: a batch of 1994+ frames only -- the longer symbol loops -- gains from 8 lanes per
// frame from about 10 frames per SIMD, 1993 and mixed batches from about 32.  The 16-frames variant (fewest instructions
// per frame, but only three wavefronts per SIMD for its LDS) no longer wins at any size and is only taken on request.
static int chooseFpw(const DcsCtx *ctx, uint32_t nJobs, bool all94)
{
    if (ctx->fpwOverride != 0)
        return ctx->fpwOverride;
    const uint64_t simds = static_cast<uint64_t>(ctx->numCUs) * 4;
    return nJobs > simds * (all94 ? 10 : 32) ? 8 : 4;
}

extern "C" void dcs_batch_destroy(DcsBatch *b)
{
    if (b == nullptr)
        return;
    (void)hipSetDevice(b->ctx->device);
    // nothing of this batch is in flight when its buffers are recycled, on the caller's launch stream or on the batch's own.
    // (A batch whose PCM has been waited for is known to be through: its stream may be busy with other lists by now --
    // the pipeline's workers keep theirs -- and waiting for those costs the collecting thread 0.7 ms a list.)
    if (!b->settled)
    {
        if (b->evDone) (void)waitLaunched(b);
        (void)streamWait(b->ctx, b->stream);
    }
    void *ptrs[] = { b->dBlob, b->dSrcs, b->dTable, b->dTailsIn, b->dPcm, b->errJoined ? nullptr : b->dErr, b->dTailsOut, b->dDebug, b->dHandoff, b->dPackages };
    for (int i = 0 ; i < 10 ; ++i)
        cacheFree(b->ctx, false, ptrs[i], b->cap[i]);
    cacheFree(b->ctx, false, b->dPlanSlots, b->planSlotsCap);
    cacheFree(b->ctx, false, b->dPlanSrcs, b->planSrcsCap);
    cacheFree(b->ctx, true, b->hStage, b->hStageCap);
    cacheFree(b->ctx, true, b->hPcm, b->hCap[0]);
    if (!b->errJoined || b->hCap[1] != 0)
        cacheFree(b->ctx, true, b->hErr, b->hCap[1]);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->evDone) (void)hipEventDestroy(b->evDone);
    delete b;
}

static DcsKernelArgs kernelArgs(const DcsBatch *b)
{
    DcsKernelArgs args;
    args.blob = b->dBlob;
    args.blobLen = b->blobOnDevice;
    args.srcs = b->dSrcs;
    args.packages = b->dPackages;
    args.nChunks = b->nChunks;
    args.nJobs = b->nJobs;
    args.pcm = b->dPcm;
    args.err = b->dErr;
    args.tailsIn = b->dTailsIn;
    args.tailsOut = b->dTailsOut;
    args.tables = b->ctx->dTables;
    args.debug = b->dDebug;
    args.handoff = b->dHandoff;
    args.epoch = b->epoch;
    args.flags = b->flags | ((dcsPkgImgDw(b->imgDw) != 0 ? b->imgDw : (b->imgDw | dcsPoolCapacity(b->fpw))) << DCS_BATCH_IMG_SHIFT);
    return args;
}

// The description of a batch is checked on the host: the kernel trusts indices and formats.  *payloadBits = the bits the batch's
// sources occupy, *batchFlags = DCS_BATCH_HAS_93A_T1 where it applies.
static DcsStatus validateBatch(DcsCtx *ctx, size_t blobLen, const DcsSrcDesc *srcs, uint32_t nSrcs, const DcsFrameJob *jobs, uint32_t nJobs,
                               const int16_t *tailsIn, uint32_t nTailsIn, uint64_t *payloadBitsOut, uint32_t *batchFlagsOut)
{
    uint64_t payloadBits = 0;
    uint32_t batchFlags = 0;
    for (uint32_t j = 0 ; j < nJobs ; ++j)
    {
        const DcsFrameJob &jb = jobs[j];
        if (jb.nSrc > DCS_MAX_CHANNELS || jb.volShift > 8 || jb.xform > DCS_XFORM_94
            || (jb.nSrc != 0 && (jb.firstSrc >= nSrcs || jb.firstSrc + jb.nSrc > nSrcs)))
        {
            setError(ctx, "job " + std::to_string(j) + ": bad source range / volShift / xform");
            return DCS_ERR_INVALID_ARG;
        }
        if (jb.prev != DCS_PREV_NONE)
        {
            const bool ext = (jb.prev & DCS_PREV_EXT) != 0;
            if (ext ? ((jb.prev & 0x7FFFFFFFu) >= nTailsIn || tailsIn == nullptr) : (jb.prev >= nJobs || jb.prev == j))
            {
                setError(ctx, "job " + std::to_string(j) + ": bad overlap predecessor");
                return DCS_ERR_INVALID_ARG;
            }
            // a decoder object has ONE transform (DCSDecoderNative.cpp:3147-3160), so a frame and the frame whose tail
            // it overlaps with share it; the two transforms also publish tails of different word counts
            if (!ext && jobs[jb.prev].xform != jb.xform)
            {
                setError(ctx, "job " + std::to_string(j) + ": overlap predecessor uses the other transform");
                return DCS_ERR_INVALID_ARG;
            }
        }
    }
    for (uint32_t s = 0 ; s < nSrcs ; ++s)
    {
        const DcsSrcDesc &sd = srcs[s];
        if (sd.format > DCS_FMT_94_T1_S3 || (sd.hdrLen != 16 && sd.hdrLen != 1) || sd.streamOff + 2 + sd.hdrLen > blobLen)
        {
            setError(ctx, "source " + std::to_string(s) + ": bad format / header length / stream offset");
            return DCS_ERR_INVALID_ARG;
        }
        // the index record steers where lanes start reading and writing inside LDS: it must be self-consistent
        const DcsFrameIndex &ix = sd.idx;
        bool ok = ix.hdrBits <= ix.nBits && ix.nBands <= (sd.format == DCS_FMT_93A_T1 ? 31 : 16);
        for (int k = 0 ; k < 15 && ok ; ++k)
            ok = ix.split[k].bitDelta <= ix.nBits && (ix.split[k].state & 0x1FFu) <= 256u;
        // the two places a lane can start from besides the band starts: the middle of band 15 of a 1994+ frame
        // (split[14].prv / .prvDelta) and bands 16 and 17 of an OS93a Type-1 frame (records in the bandType bytes)
        if (ok && sd.format >= DCS_FMT_94_T0)
            ok = ix.split[14].prv <= ix.nBits && (ix.split[14].prvDelta & 0x1FFu) <= 256u;
        if (ok && sd.format == DCS_FMT_93A_T1)
            for (int k = 0 ; k < 2 && ok ; ++k)
            {
                DcsSplit t;
                memcpy(&t, ix.bandType + 8 * k, sizeof(t));
                ok = t.bitDelta <= ix.nBits && (t.state & 0x1FFu) <= 256u;
            }
        if (!ok)
        {
            setError(ctx, "source " + std::to_string(s) + ": inconsistent frame index record");
            return DCS_ERR_INVALID_ARG;
        }
        payloadBits += sd.idx.nBits;
        if (sd.format == DCS_FMT_93A_T1)
            batchFlags |= DCS_BATCH_HAS_93A_T1;
    }

    *payloadBitsOut = payloadBits;
    *batchFlagsOut = batchFlags;
    return DCS_OK;
}

// stream: where the batch's uploads, default launches and downloads run (nullptr: the context's); handoff: how tails
// cross chunk boundaries (DcsCtx::handoff, or forced off for the second attempt after a lost tail)
static DcsStatus createBatch(DcsCtx *ctx,
                             const uint8_t *blob, size_t blobLen,
                             const DcsSrcDesc *srcs, uint32_t nSrcs,
                             const DcsFrameJob *jobs, uint32_t nJobs,
                             const int16_t *tailsIn, uint32_t nTailsIn,
                             hipStream_t stream, bool handoff, DcsBatch **out)
{
    if (ctx == nullptr || out == nullptr || jobs == nullptr || nJobs == 0 || (nSrcs != 0 && (srcs == nullptr || blob == nullptr)))
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;

    uint64_t payloadBits = 0;
    uint32_t batchFlags = 0;
    {
        const DcsStatus vst = validateBatch(ctx, blobLen, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, &payloadBits, &batchFlags);
        if (vst != DCS_OK)
            return vst;
    }

    DcsBatch *b = new (std::nothrow) DcsBatch;
    if (b == nullptr)
        return DCS_ERR_NO_MEMORY;
    b->ctx = ctx;
    b->stream = stream ? stream : ctx->stream;
    b->nJobs = nJobs; b->nSrcs = nSrcs; b->nTailsIn = nTailsIn; b->blobLen = blobLen;
    {
        bool all94 = nJobs != 0;
        for (uint32_t j = 0 ; j < nJobs && all94 ; ++j)
            all94 = jobs[j].xform == DCS_XFORM_94;
        b->fpw = chooseFpw(ctx, nJobs, all94);
    }
    b->flags = batchFlags;

    thread_local std::vector<DcsSlot> slots;    // (kept from batch to batch: see the pipeline's scratch)
    uint8_t *hPackages = nullptr;           // pinned staging for the chunk packages
    size_t pkgBytes = 0;
    static const bool forceRanges = getenv("DCS_BATCH_XCD_RANGES") != nullptr && atoi(getenv("DCS_BATCH_XCD_RANGES")) != 0;     // (experiment switch)
    const bool ranges = (tlsXcdRanges || ctx->xcdRanges || forceRanges) && !ctx->noXcdRanges;
    static const bool chainOrder = getenv("DCS_EXP_CHAIN_ORDER") != nullptr && atoi(getenv("DCS_EXP_CHAIN_ORDER")) != 0;      // (experiment switch)
    // (experiment switch, for A/B runs on one binary: round 4's packages -- the full pool image, every frame's tail stored)
    static const bool fullImage = getenv("DCS_EXP_FULL_IMAGE") != nullptr && atoi(getenv("DCS_EXP_FULL_IMAGE")) != 0;
    const bool allTails = ctx->keepAllTails || tlsKeepAllTails || fullImage;
    if (tlsResidentBatch && !fullImage)
        b->nChunks = dcsPlanChunksCapped(jobs, nJobs, srcs, b->fpw, slots, handoff, ctx->framesPerChunk, !ranges && !chainOrder, allTails, &b->imgDw,
                                          static_cast<uint32_t>(ctx->numCUs) * 16u);
    else
    {
        b->nChunks = dcsPlanChunks(jobs, nJobs, srcs, b->fpw, slots, handoff, ctx->framesPerChunk, !ranges && !chainOrder, allTails);
        b->imgDw = fullImage ? dcsPoolCapacity(b->fpw) : dcsImageDwords(slots.data(), b->nChunks, b->fpw);
    }
    dcsShuffleChunks(slots, b->nChunks, b->fpw, ctx->shuffleSeed);          // (test hook)
    if (!fullImage && dcsAllSources94(jobs, nJobs, srcs))
        b->imgDw |= DCS_PKG_SPLIT4;             // (the layout word: every source a 1994+ frame -> 4-byte split records)
    if (ranges && handoff)
        b->flags |= DCS_BATCH_XCD_RANGES;

    // algorithmic bytes per launch: the exact compressed payload, the header (and U16 frame count) of every
    // stream the batch draws on, a 56-byte frame descriptor per source, 480 bytes of PCM per output frame.  abiBytes
    // counts the descriptors as this ABI has them instead: 160-byte DcsSrcDesc, 16-byte DcsFrameJob.
    {
        uint64_t hdrBytes = 0;
        uint64_t lastOff = ~0ull;
        std::unordered_set<uint64_t> seen;
        for (uint32_t k = 0 ; k < nSrcs ; ++k)
            if (srcs[k].streamOff != lastOff)               // (the frames of a stream are normally consecutive sources)
            {
                lastOff = srcs[k].streamOff;
                if (seen.insert(lastOff).second)
                    hdrBytes += 2u + srcs[k].hdrLen;
            }
        const uint64_t payload = (payloadBits + 7) / 8, pcm = static_cast<uint64_t>(nJobs) * DCS_FRAME_SAMPLES * 2;
        b->algoBytes = payload + hdrBytes + static_cast<uint64_t>(nSrcs) * 56u + pcm;
        b->abiBytes = payload + static_cast<uint64_t>(nSrcs) * sizeof(DcsSrcDesc) + static_cast<uint64_t>(nJobs) * sizeof(DcsFrameJob) + pcm;
    }

    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        // Round 0 of every frame reads the chunk packages only; the streams and the full descriptors are needed on the
        // device just for the further sources of multi-channel frames.
        bool multi = false;
        for (uint32_t j = 0 ; j < nJobs && !multi ; ++j)
            multi = jobs[j].nSrc > 1;
        b->blobOnDevice = multi ? blobLen : 0;
        if (multi)
        {
            const size_t blobAlloc = ((blobLen + 3) & ~size_t(3)) + 64;     // zero tail: the bit reader prefetches past the end
            b->cap[0] = blobAlloc; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dBlob), b->cap[0]));
            HIPCHK(ctx, hipMemsetAsync(b->dBlob, 0, blobAlloc, b->stream));
            if (blobLen)
                HIPCHK(ctx, hipMemcpyAsync(b->dBlob, blob, blobLen, hipMemcpyHostToDevice, b->stream));
        }
        if (multi && nSrcs)
        {
            b->cap[1] = sizeof(DcsSrcDesc) * nSrcs; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dSrcs), b->cap[1]));
            HIPCHK(ctx, hipMemcpyAsync(b->dSrcs, srcs, sizeof(DcsSrcDesc) * nSrcs, hipMemcpyHostToDevice, b->stream));
        }
        // built in pinned host memory (recycled by the context like the device buffers): the upload runs at link speed
        pkgBytes = static_cast<size_t>(b->nChunks) * dcsPkgStride(b->fpw, b->imgDw);
        HIPCHK(ctx, cacheAlloc(ctx, true, reinterpret_cast<void **>(&hPackages), pkgBytes));
        dcsBuildPackages(slots.data(), b->nChunks, b->fpw, srcs, blob, blobLen, hPackages, b->imgDw);
        b->cap[9] = pkgBytes; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dPackages), b->cap[9]));
        HIPCHK(ctx, hipMemcpyAsync(b->dPackages, hPackages, pkgBytes, hipMemcpyHostToDevice, b->stream));
        if (nTailsIn)
        {
            b->cap[3] = sizeof(int16_t) * 16 * nTailsIn; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dTailsIn), b->cap[3]));
            HIPCHK(ctx, hipMemcpyAsync(b->dTailsIn, tailsIn, sizeof(int16_t) * 16 * nTailsIn, hipMemcpyHostToDevice, b->stream));
        }
        b->cap[4] = sizeof(int16_t) * DCS_FRAME_SAMPLES * nJobs; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dPcm), b->cap[4]));
        b->cap[5] = sizeof(uint32_t) * nJobs; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dErr), b->cap[5]));
        b->cap[6] = sizeof(int16_t) * 16 * nJobs; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dTailsOut), b->cap[6]));
        HIPCHK(ctx, hipMemsetAsync(b->dTailsOut, 0, b->cap[6], b->stream));   // (rows of frames whose tail is not kept read as zero)
        HIPCHK(ctx, hipMemsetAsync(b->dErr, 0, sizeof(uint32_t) * nJobs, b->stream));
        b->cap[8] = sizeof(unsigned long long) * 16 * (b->nChunks + 1); HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dHandoff), b->cap[8]));
        HIPCHK(ctx, hipMemsetAsync(b->dHandoff, 0, b->cap[8], b->stream));     // epoch 0 = never written
#ifdef DCS_STAMPS
        b->cap[7] = sizeof(unsigned long long) * 16 * (b->nChunks + 4); HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dDebug), b->cap[7]));
        HIPCHK(ctx, hipMemsetAsync(b->dDebug, 0, sizeof(unsigned long long) * 16 * (b->nChunks + 4), b->stream));
#endif
        HIPCHK(ctx, hipEventCreate(&b->ev0));
        HIPCHK(ctx, hipEventCreate(&b->ev1));
        HIPCHK(ctx, hipEventCreateWithFlags(&b->evDone, hipEventDisableTiming | (tlsBlockingWaits ? hipEventBlockingSync : 0u)));
        HIPCHK(ctx, streamWait(b->ctx, b->stream));
        return DCS_OK;
    }();
    if (st != DCS_OK)
        (void)streamWait(b->ctx, b->stream);    // (on success the lambda has already waited for the uploads)
    cacheFree(ctx, true, hPackages, pkgBytes);
    if (st != DCS_OK)
    {
        dcs_batch_destroy(b);
        return st;
    }
    *out = b;
    return DCS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The packer on the device: the same packages dcsBuildPackages (dcs_plan.cpp) lays out on the host, assembled by one
// wavefront per chunk from what is already resident -- the index records the device index pass left there, the streams
// as uploaded for that pass -- and the plan the host made from an 8-byte-per-frame digest.  Byte for byte the same
// packages (tests/test_gpu_corpus.py compares them).  The packages buffer is zeroed beforehand.
// ---------------------------------------------------------------------------------------------------------
namespace {
template <int FPW>
__global__ __launch_bounds__(256) void dcsPackKernel(const DcsSlot *slots, uint32_t nChunks, const DcsPlanSrc *srcs,
                                                      const DcsFrameIndex *records, const uint8_t *blob, uint64_t blobLen,
                                                      uint8_t *packages, uint32_t layout)
{
    const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = static_cast<int>(threadIdx.x & 63);
    if (chunk >= nChunks)
        return;
    uint8_t *pkg = packages + static_cast<size_t>(chunk) * dcsPkgStride(FPW, layout);
    const DcsSlot *cs = slots + static_cast<size_t>(chunk) * FPW;
    const uint32_t imgDw = dcsPkgImgDw(layout);
    const bool split4 = (layout & DCS_PKG_SPLIT4) != 0;
    // (order: the image first -- its loads need nothing but the plan's slots, and a wavefront issues its loads in program order: behind the
    // slot -> source -> record chain of the descriptors they waited for three round trips before they were even asked for)
    // the image of the bit pool: the chunk's runs of stream dwords, in bit order; zero between them (the runs lie one behind the
    // other, each on a 16-byte boundary: gaps of at most three dwords) and behind the last one
    uint32_t *img = reinterpret_cast<uint32_t *>(pkg + dcsPkgOffPool(FPW, layout));
    uint32_t end = 0;
    for (int k = 0 ; k < FPW ; ++k)
    {
        const uint32_t n = cs[k].runNDw, st = cs[k].runStartDw, o = cs[k].runPoolOff;
        if (n == 0)
            break;
        if (o + n > imgDw || o < end)
            continue;                           // (cannot happen: the image covers every run of the plan, in order)
        for (uint32_t i = end + static_cast<uint32_t>(lane) ; i < o ; i += 64)
            img[i] = 0;
        // (four dwords a lane where the run and the blob have them: the run starts on a 16-byte boundary of the image, the stream
        // bytes on a dword boundary only)
        const uint32_t n4 = (static_cast<uint64_t>(st) + n) * 4 <= blobLen ? n & ~3u : 0u;
        for (uint32_t i = static_cast<uint32_t>(lane) * 4 ; i < n4 ; i += 256)
        {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(blob + (static_cast<uint64_t>(st) + i) * 4);
            const uint32_t w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
            *reinterpret_cast<uint4 *>(img + o + i) = make_uint4(__builtin_bswap32(w0), __builtin_bswap32(w1), __builtin_bswap32(w2), __builtin_bswap32(w3));
        }
        for (uint32_t i = n4 + static_cast<uint32_t>(lane) ; i < n ; i += 64)
        {
            const uint64_t b0 = (static_cast<uint64_t>(st) + i) * 4;
            uint32_t w = 0;
            if (b0 + 4 <= blobLen)
                w = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(blob + b0));
            else
                for (int j = 0 ; j < 4 ; ++j)
                    if (b0 + j < blobLen)
                        w |= static_cast<uint32_t>(blob[b0 + j]) << (24 - 8 * j);
            img[o + i] = w;
        }
        end = o + n;
    }
    {
        // (behind the last run: dwords up to the next 16-byte boundary, then sixteen bytes a lane)
        const uint32_t end4 = (end + 3u) & ~3u;
        for (uint32_t i = end + static_cast<uint32_t>(lane) ; i < end4 && i < imgDw ; i += 64)
            img[i] = 0;
        for (uint32_t i = end4 + static_cast<uint32_t>(lane) * 4 ; i + 4 <= imgDw ; i += 256)
            *reinterpret_cast<uint4 *>(img + i) = make_uint4(0, 0, 0, 0);
        for (uint32_t i = (imgDw & ~3u) + static_cast<uint32_t>(lane) ; i < imgDw ; i += 64)
            if (i >= end4)
                img[i] = 0;
    }
    // EVERY byte of the package is written here, once (round 5: the buffer used to be cleared first, 700 MB of writes in front of
    // the packer for 2 M frames): what does not apply is written as zero, as the host packer's memset leaves it.
    // slot `lane`: its first 16 bytes, the descriptor head (the first 40 bytes of what DcsSrcDesc would be) with poolOff and bpl
    // behind it, the stream header (dcs_common.h: five 16-byte pieces per slot)
    if (lane < FPW)
    {
        const DcsSlot sl = cs[lane];
        uint4 *ps = reinterpret_cast<uint4 *>(pkg + static_cast<size_t>(lane) * DCS_PKG_SLOT_BYTES);
        uint32_t d[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, h[4] = { 0, 0, 0, 0 };
        if (!(sl.flags & DCS_SLOT_EMPTY) && sl.nSrc != 0)
        {
            const DcsPlanSrc sd = srcs[sl.firstSrc];
            const uint32_t *rec = reinterpret_cast<const uint32_t *>(&records[sd.record]);
            d[0] = static_cast<uint32_t>(sd.streamOff);
            d[1] = static_cast<uint32_t>(sd.streamOff >> 32);
            d[2] = static_cast<uint32_t>(sd.mixMul) | (static_cast<uint32_t>(sd.format) << 16) | (static_cast<uint32_t>(sd.hdrLen) << 24);
#pragma unroll
            for (int i = 0 ; i < 7 ; ++i)
                d[3 + i] = rec[i];              // bitOff, nBits | hdrBits, bandType[16], preAdj | nBands | flags
            const uint64_t hOff = sd.streamOff + 2;
            const uint32_t hLen = sd.hdrLen == 1 ? 1u : 16u;
#pragma unroll
            for (uint32_t i = 0 ; i < 16 ; ++i)
                if (i < hLen && hOff + i < blobLen)
                    h[i >> 2] |= static_cast<uint32_t>(blob[hOff + i]) << (8 * (i & 3));
        }
        ps[0] = reinterpret_cast<const uint4 *>(&cs[lane])[0];
        ps[1] = make_uint4(d[0], d[1], d[2], d[3]);
        ps[2] = make_uint4(d[4], d[5], d[6], d[7]);
        ps[3] = make_uint4(d[8], d[9], static_cast<uint32_t>(sl.poolOff) | (static_cast<uint32_t>(sl.bpl) << 16), sl.nextJob);
        ps[4] = make_uint4(h[0], h[1], h[2], h[3]);
    }
    // the lane's own record: first band and split record of the frame's q-th unpack lane (zero where there is none)
    {
        constexpr int SUB = 64 / FPW;
        const int sI = lane % FPW, q = lane / FPW;
        const DcsSlot sl = cs[sI];
        uint32_t r0 = 0, r1 = 0;
        if (q >= 1 && q < SUB && !(sl.flags & DCS_SLOT_EMPTY) && sl.nSrc != 0 && sl.bpl != 0)
        {
            const DcsPlanSrc sd = srcs[sl.firstSrc];
            const int nb16 = sd.nBands < 16 ? sd.nBands : 16;
            const int nbEnd = dcsDealEnd(sd.format, sd.nBands);
            const int base = dcsLaneFirstBand(sd.format, q, sl.bpl, nbEnd);
            r0 = 0x8000u;                       // bitDelta bit 15: no bands for this lane
            const uint32_t *mid = reinterpret_cast<const uint32_t *>(&records[sd.record].split[14]);
            if (q == SUB - 1 && dcsMid15(sd.format, sl.bpl, nb16, mid[0] >> 16))
            {
                // the second half of band 15 (1994+, one band per lane)
                r0 = mid[0] >> 16;
                r1 = ((mid[1] & 0x3FFu) | DCS_SPLIT_MID15 | (15u << 12)) << 16;
            }
            else if (base >= nbEnd)
                ;
            else if (base >= 16)
            {
                // OS93a Type 1, bands 16 and 17: their records travel in the frame record's bandType bytes (at byte 8 of
                // the record, so dword-aligned)
                const uint32_t *sp = reinterpret_cast<const uint32_t *>(records[sd.record].bandType) + 2 * (base - 16);
                r0 = sp[0];
                r1 = (sp[1] & 0x0DFFFFFFu) | (DCS_SPLIT_BASE16 << 16) | (static_cast<uint32_t>(base - 16) << 28);
            }
            else
            {
                const uint32_t *sp = reinterpret_cast<const uint32_t *>(&records[sd.record].split[base - 1]);
                r0 = sp[0];
                r1 = (sp[1] & 0x0FFFFFFFu) | (static_cast<uint32_t>(base) << 28);
            }
        }
        if (split4)
            reinterpret_cast<uint32_t *>(pkg + dcsPkgOffSplit(FPW))[lane] = (r0 & 0xFFFFu) | (r1 & 0xFFFF0000u);
        else
            reinterpret_cast<uint2 *>(pkg + dcsPkgOffSplit(FPW))[lane] = make_uint2(r0, r1);
    }
    // (the bytes between the split records and the image's 128-byte boundary)
    {
        const uint32_t padFrom = dcsPkgOffSplit(FPW) + 64u * dcsPkgSplitBytes(layout), padTo = dcsPkgOffPool(FPW, layout);
        for (uint32_t i = padFrom / 4 + static_cast<uint32_t>(lane) ; i < padTo / 4 ; i += 64)
            reinterpret_cast<uint32_t *>(pkg)[i] = 0;
    }
}
}   // namespace

// ---------------------------------------------------------------------------------------------------------
// The planner on the device, for lists of WHOLE STREAMS (the pipeline's third step onto the device: the index records never
// leave it, and the host neither waits for them nor plans).  The job list of such a list is regular -- stream k's frames
// f = 0 .. nFrames + extraFrames - 1 one after the other, each the successor of the one before -- so the chunk plan is
// arithmetic: chunk c holds jobs c * FPW .. c * FPW + FPW - 1, a frame whose predecessor lies in the chunk before imports
// its tail from there.  One thread per chunk writes the chunk's slots (run placement as the host planner does it,
// dcs_plan.cpp: placeFrame) and the source digests of its frames.  What the arithmetic plan cannot express is reported in
// a flag word and the list then takes the host planner's path: a chunk whose compressed bytes overflow the bit pool (the
// host planner closes such a chunk early), a stream whose frames run past its buffer.  A stream the index pass stopped
// early (nValidFrames < nFrames) needs no flag: its remaining frames are silent here as there.
// ---------------------------------------------------------------------------------------------------------
namespace {
template <int FPW>
__global__ __launch_bounds__(256) void dcsPlanKernel(const DcsPlanStream *streams, uint32_t nStreams, uint32_t extraFrames, uint32_t nJobs,
                                                      const DcsFrameIndex *records, const DcsStreamInfo *infos,
                                                      DcsSlot *slots, DcsPlanSrc *srcs, uint32_t *flagWord, uint32_t *hostFlag, uint32_t fpc)
{
    // fpc: frames a chunk holds, FPW or -- for a list whose frames are too large for FPW of them to share the bit pool -- fewer (the
    // chunk's other slots stay empty): chunk c holds jobs c * fpc .. c * fpc + fpc - 1
    const uint32_t nChunks = (nJobs + fpc - 1) / fpc;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nChunks)
        return;
    // the stream of the chunk's first job: the last stream whose first job is not behind it
    uint32_t lo = 0, hi = nStreams - 1;
    const uint32_t j0 = c * fpc;
    while (lo < hi)
    {
        const uint32_t mid = (lo + hi + 1) / 2;
        if (streams[mid].firstJob <= j0) lo = mid; else hi = mid - 1;
    }
    uint32_t k = lo;
    DcsPlanStream st = streams[k];
    uint32_t nValid = min(static_cast<uint32_t>(infos[k].nValidFrames), st.nFrames);
    uint32_t flags = 0;

    DcsSlot out[FPW];
    uint32_t nRuns = 0, runUse = 0, poolUse = 0, curStart = 0, curN = 0, curOff = 0;    // cur*: the chunk's last run
    const DcsSlot empty{ 0xFFFFFFFFu, DCS_NO_PREV_SLOT, DCS_SLOT_EMPTY, 0, 0, 0, DCS_PREV_NONE, 0, 0, 0, 0, 0, 0, 0 };
    for (int p = 0 ; p < FPW ; ++p)
    {
        const uint32_t j = j0 + static_cast<uint32_t>(p);
        if (j >= nJobs || static_cast<uint32_t>(p) >= fpc) { out[p] = empty; continue; }
        while (k + 1 < nStreams && j >= streams[k + 1].firstJob)
        {
            ++k;
            st = streams[k];
            nValid = min(static_cast<uint32_t>(infos[k].nValidFrames), st.nFrames);
        }
        const uint32_t f = j - st.firstJob, framesOut = st.nFrames + extraFrames;
        const bool has = f < nValid;
        if (f == 0)
        {
            // (what counts is the bits the frames occupy, not nBytes, which includes the reference reader's look-ahead)
            const DcsStreamInfo in = infos[k];
            if (in.nFrames == 0 || 2u + static_cast<uint32_t>(in.hdrLen) + (in.payloadBits + 7) / 8 > st.len)
                flags |= DCS_PLAN_TRUNCATED;
        }
        DcsSlot sl{ j, DCS_NO_PREV_SLOT, 0, static_cast<uint8_t>(has ? 1 : 0),
                    static_cast<uint8_t>((has ? (f == 0 ? st.volShift0 : st.volShiftN) : 8) | (st.xform << 4)),
                    has ? st.firstRecord + f : 0u, f == 0 ? DCS_PREV_NONE : j - 1, 0, 0, 0, 0, 0, 0, 0 };
        if (f != 0)
        {
            if (p != 0)
                sl.prevSlot = static_cast<uint8_t>(p - 1);
            else
            {
                sl.flags |= DCS_SLOT_IMPORT;            // the chunk before publishes the tail (its last frame is this one's predecessor)
                sl.prevJob = c - 1;
            }
        }
        if ((static_cast<uint32_t>(p) == fpc - 1 || j + 1 == nJobs) && f + 1 < framesOut && j + 1 < nJobs)
        {
            sl.flags |= DCS_SLOT_EXPORT;
            sl.nextJob = j + 1;                     // (the stream's next frame: the first job of the next chunk)
        }
        if (f + 1 == framesOut)
            sl.flags |= DCS_SLOT_KEEP_TAIL;         // the last frame of its chain (dcs_plan.cpp)
        if (has)
        {
            const uint32_t *rec = reinterpret_cast<const uint32_t *>(&records[st.firstRecord + f]);
            const uint32_t bitOff = rec[0], nBits = rec[1] & 0xFFFFu, nBands = (rec[6] >> 16) & 0xFFu, fl = rec[6] >> 24;
            const uint32_t sub = 64 / FPW, nb16 = nBands < 16 ? nBands : 16;
            const uint32_t bpl = (nb16 + sub - 1) / sub;
            sl.bpl = (fl & DCS_IDX_SERIAL) ? 0 : static_cast<uint8_t>(bpl < 1 ? 1 : bpl);
            // where the frame's bytes go in the pool: it extends the chunk's last run or opens a new one (placeFrame, dcs_plan.cpp)
            const uint64_t bitPos = (st.streamOff + 2 + st.hdrLen) * 8 + bitOff;
            const uint32_t s0 = static_cast<uint32_t>(bitPos >> 5);
            const uint32_t n = dcsPoolDwords(st.streamOff, st.hdrLen, bitOff, nBits);
            if (nRuns != 0 && s0 >= curStart && s0 <= curStart + curN)
                curN = max(curN, s0 + n - curStart);
            else
            {
                curStart = s0; curN = n; curOff = runUse;
                ++nRuns;
            }
            runUse = curOff + ((curN + 3) & ~3u);
            sl.poolOff = static_cast<uint16_t>(curOff + (s0 - curStart));
            poolUse += (n + 3) & ~3u;
            srcs[st.firstRecord + f] = DcsPlanSrc{ st.streamOff, bitOff, static_cast<uint16_t>(nBits), st.hdrLen, static_cast<uint8_t>(nBands),
                                                   static_cast<uint8_t>(fl), st.format, f == 0 ? st.mixMul0 : st.mixMulN, st.firstRecord + f };
        }
        out[p] = sl;
        if (has)
            for (int r = 0 ; r <= p ; ++r)                  // slot k carries run k (constant subscripts: the slots stay in registers)
                if (static_cast<uint32_t>(r) + 1 == nRuns)
                {
                    out[r].runStartDw = curStart;
                    out[r].runNDw = static_cast<uint16_t>(curN);
                    out[r].runPoolOff = static_cast<uint16_t>(curOff);
                }
    }
    if (poolUse > dcsPoolCapacity(FPW) || runUse > dcsPoolCapacity(FPW))
        flags |= DCS_PLAN_POOL_OVERFLOW;
    for (int p = 0 ; p < FPW ; ++p)
        slots[static_cast<size_t>(c) * FPW + p] = out[p];
    if (flags != 0)
    {
        const uint32_t seen = atomicOr(flagWord, flags);
        // ... and straight into the batch's pinned staging word, so that no copy kernel has to bring it down behind the decode
        // launch.  A plain store of everything this thread knows raised (its own bits and those the device word held before):
        // non-zero is all the host needs for the PCM to be safe; concurrent writers can still hide each other's bits, which
        // is why the host retries only on "overflow and NOT truncated" and treats the word as "at least these" 
        if (hostFlag != nullptr)
            __hip_atomic_store(hostFlag, seen | flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}   // namespace

// plan on the host from source digests, pack on the device: `dRecords` (the index records as the device index pass
// wrote them) and `dBlob` (the streams as uploaded for it) must stay valid until the pack kernel has run, i.e. until the
// batch's stream has been waited for once
static DcsStatus createBatchOnDevice(DcsCtx *ctx, const DcsFrameJob *jobs, uint32_t nJobs, const DcsPlanSrc *srcs, uint32_t nSrcs,
                                     const DcsFrameIndex *dRecords, const uint8_t *dBlob, uint64_t blobLen,
                                     hipStream_t stream, bool handoff, DcsBatch **out)
{
    *out = nullptr;
    DcsBatch *b = new (std::nothrow) DcsBatch;
    if (b == nullptr)
        return DCS_ERR_NO_MEMORY;
    b->ctx = ctx;
    b->stream = stream ? stream : ctx->stream;
    b->nJobs = nJobs; b->nSrcs = nSrcs;
    bool all94 = nJobs != 0;
    for (uint32_t j = 0 ; j < nJobs && all94 ; ++j)
        all94 = jobs[j].xform == DCS_XFORM_94;
    b->fpw = chooseFpw(ctx, nJobs, all94);
    uint64_t payloadBits = 0, hdrBytes = 0, lastOff = ~0ull;
    for (uint32_t k = 0 ; k < nSrcs ; ++k)
    {
        payloadBits += srcs[k].nBits;
        if (srcs[k].format == DCS_FMT_93A_T1)
            b->flags |= DCS_BATCH_HAS_93A_T1;
        if (srcs[k].streamOff != lastOff)       // (a list's frames come stream by stream)
        {
            lastOff = srcs[k].streamOff;
            hdrBytes += 2u + srcs[k].hdrLen;
        }
    }
    const uint64_t payload = (payloadBits + 7) / 8, pcm = static_cast<uint64_t>(nJobs) * DCS_FRAME_SAMPLES * 2;
    b->algoBytes = payload + hdrBytes + static_cast<uint64_t>(nSrcs) * 56u + pcm;
    b->abiBytes = payload + static_cast<uint64_t>(nSrcs) * sizeof(DcsSrcDesc) + static_cast<uint64_t>(nJobs) * sizeof(DcsFrameJob) + pcm;

    thread_local std::vector<DcsSlot> slots;
    const bool ranges = (tlsXcdRanges || ctx->xcdRanges) && !ctx->noXcdRanges;
    b->nChunks = dcsPlanChunksLite(jobs, nJobs, srcs, b->fpw, slots, handoff, ctx->framesPerChunk, !ranges);
    b->imgDw = dcsImageDwords(slots.data(), b->nChunks, b->fpw);
    {
        bool all94 = nSrcs != 0;
        for (uint32_t k = 0 ; k < nSrcs && all94 ; ++k)
            all94 = srcs[k].format >= DCS_FMT_94_T0;
        if (all94)
            b->imgDw |= DCS_PKG_SPLIT4;
    }
    if (ranges && handoff)
        b->flags |= DCS_BATCH_XCD_RANGES;
    dcsShuffleChunks(slots, b->nChunks, b->fpw, ctx->shuffleSeed);          // (test hook)
    const size_t pkgBytes = static_cast<size_t>(b->nChunks) * dcsPkgStride(b->fpw, b->imgDw);
    void *stage = nullptr;
    size_t stageBytes = 0;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        b->planSlotsCap = sizeof(DcsSlot) * slots.size();
        b->planSrcsCap = sizeof(DcsPlanSrc) * (nSrcs ? nSrcs : 1);
        HIPCHK(ctx, cacheAlloc(ctx, false, &b->dPlanSlots, b->planSlotsCap));
        HIPCHK(ctx, cacheAlloc(ctx, false, &b->dPlanSrcs, b->planSrcsCap));
        // (through pinned staging: a copy from pageable memory holds the calling thread until the stream gets to it)
        stageBytes = b->planSlotsCap + sizeof(DcsPlanSrc) * nSrcs;
        HIPCHK(ctx, cacheAlloc(ctx, true, &stage, stageBytes));
        memcpy(stage, slots.data(), b->planSlotsCap);
        if (nSrcs)
            memcpy(static_cast<uint8_t *>(stage) + b->planSlotsCap, srcs, sizeof(DcsPlanSrc) * nSrcs);
        HIPCHK(ctx, copyByKernel(b->stream, b->dPlanSlots, stage, b->planSlotsCap));
        if (nSrcs)
            HIPCHK(ctx, copyByKernel(b->stream, b->dPlanSrcs, static_cast<uint8_t *>(stage) + b->planSlotsCap, sizeof(DcsPlanSrc) * nSrcs));
        b->cap[9] = pkgBytes; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dPackages), b->cap[9]));
        const uint32_t blocks = (b->nChunks + 3) / 4;       // (the pack kernel writes every byte of the packages: nothing to clear)
        const DcsSlot *dS = static_cast<const DcsSlot *>(b->dPlanSlots);
        const DcsPlanSrc *dP = static_cast<const DcsPlanSrc *>(b->dPlanSrcs);
        if (b->fpw == 16)
            hipLaunchKernelGGL(dcsPackKernel<16>, dim3(blocks), dim3(256), 0, b->stream, dS, b->nChunks, dP, dRecords, dBlob, blobLen, b->dPackages, b->imgDw);
        else if (b->fpw == 8)
            hipLaunchKernelGGL(dcsPackKernel<8>, dim3(blocks), dim3(256), 0, b->stream, dS, b->nChunks, dP, dRecords, dBlob, blobLen, b->dPackages, b->imgDw);
        else
            hipLaunchKernelGGL(dcsPackKernel<4>, dim3(blocks), dim3(256), 0, b->stream, dS, b->nChunks, dP, dRecords, dBlob, blobLen, b->dPackages, b->imgDw);
        HIPCHK(ctx, hipGetLastError());
        b->cap[4] = sizeof(int16_t) * DCS_FRAME_SAMPLES * nJobs; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dPcm), b->cap[4]));
        b->cap[5] = sizeof(uint32_t) * nJobs; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dErr), b->cap[5]));
        // (no tails kept: nothing reads them on this path, and a buffer from the recycling cache would carry another list's -- ADVICE r5)
        HIPCHK(ctx, hipMemsetAsync(b->dErr, 0, sizeof(uint32_t) * nJobs, b->stream));
        b->cap[8] = sizeof(unsigned long long) * 16 * (b->nChunks + 1); HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dHandoff), b->cap[8]));
        HIPCHK(ctx, hipMemsetAsync(b->dHandoff, 0, b->cap[8], b->stream));
        HIPCHK(ctx, hipEventCreate(&b->ev0));
        HIPCHK(ctx, hipEventCreate(&b->ev1));
        HIPCHK(ctx, hipEventCreateWithFlags(&b->evDone, hipEventDisableTiming | (tlsBlockingWaits ? hipEventBlockingSync : 0u)));
        // (nothing is waited for here: the decode launch follows the pack kernel on the same stream)
        return DCS_OK;
    }();
    b->hStage = stage; b->hStageCap = stageBytes;       // (in use until the uploads have run: released with the batch)
    if (st != DCS_OK)
    {
        (void)streamWait(ctx, b->stream);
        dcs_batch_destroy(b);
        return st;
    }
    *out = b;
    return DCS_OK;
}

namespace {
// three ranges cleared by one launch: a (16-byte units), b (dwords), c (16-byte units)
__global__ __launch_bounds__(256) void dcsClear3Kernel(uint4 *a, size_t nA16, uint32_t *b, size_t nB4, uint4 *c, size_t nC16)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x, t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = t ; i < nA16 ; i += stride) a[i] = z;
    for (size_t i = t ; i < nB4 ; i += stride) b[i] = 0;
    for (size_t i = t ; i < nC16 ; i += stride) c[i] = z;
}
}   // namespace

// What a planned-on-device batch queues in front of its decode launch: error words and hand-off words cleared,
// one thread per chunk plans (dcsPlanKernel), one wavefront per chunk packs (dcsPackKernel).  `between` (optional) is
// recorded between planner and packer (the device-path timing, dcs_device_path_run).
static DcsStatus queuePlanAndPack(DcsBatch *b, uint32_t nStreams, uint32_t extraFrames, const DcsFrameIndex *dRecords, const DcsStreamInfo *dInfos,
                                  const uint8_t *dBlob, uint64_t blobLen, hipEvent_t between)
{
    DcsCtx *ctx = b->ctx;
    const uint32_t nJobs = b->nJobs;
    // error words and hand-off words cleared by ONE kernel (three hipMemsetAsync were three dispatches in a chain of
    // ten per list; epoch 0 = never written; the planner's flag word lies behind the last chunk's hand-off words)
    {
        // (the packages are no longer among them: the pack kernel writes every byte of a package itself)
        const size_t errBytes = sizeof(uint32_t) * nJobs, hoBytes = b->cap[8];
        const size_t total16 = (errBytes + 15) / 16 + hoBytes / 16;
        const unsigned blocks = static_cast<unsigned>(std::min<size_t>((total16 + 255) / 256, 2048));
        hipLaunchKernelGGL(dcsClear3Kernel, dim3(blocks), dim3(256), 0, b->stream, static_cast<uint4 *>(nullptr), static_cast<size_t>(0),
                           reinterpret_cast<uint32_t *>(b->dErr), errBytes / 4, reinterpret_cast<uint4 *>(b->dHandoff), hoBytes / 16);
        HIPCHK(ctx, hipGetLastError());
    }
    uint32_t *flagWord = reinterpret_cast<uint32_t *>(b->dHandoff + static_cast<size_t>(b->nChunks) * 16) + 2;
    const uint32_t planBlocks = (b->nChunks + 63) / 64;         // (one wavefront per workgroup: a thread's work is serial, the chunks should spread over the CUs)
    DcsSlot *dS = static_cast<DcsSlot *>(b->dPlanSlots);
    DcsPlanSrc *dP = static_cast<DcsPlanSrc *>(b->dPlanSrcs);
    const DcsPlanStream *dT = static_cast<const DcsPlanStream *>(b->dTable);
    const uint32_t blocks = (b->nChunks + 3) / 4;
    if (b->fpw == 16)
        hipLaunchKernelGGL(dcsPlanKernel<16>, dim3(planBlocks), dim3(64), 0, b->stream, dT, nStreams, extraFrames, nJobs, dRecords, dInfos, dS, dP, flagWord, static_cast<uint32_t *>(b->hStage), b->planFpc);
    else if (b->fpw == 8)
        hipLaunchKernelGGL(dcsPlanKernel<8>, dim3(planBlocks), dim3(64), 0, b->stream, dT, nStreams, extraFrames, nJobs, dRecords, dInfos, dS, dP, flagWord, static_cast<uint32_t *>(b->hStage), b->planFpc);
    else
        hipLaunchKernelGGL(dcsPlanKernel<4>, dim3(planBlocks), dim3(64), 0, b->stream, dT, nStreams, extraFrames, nJobs, dRecords, dInfos, dS, dP, flagWord, static_cast<uint32_t *>(b->hStage), b->planFpc);
    HIPCHK(ctx, hipGetLastError());
    if (between != nullptr)
        HIPCHK(ctx, hipEventRecord(between, b->stream));
    if (b->fpw == 16)
        hipLaunchKernelGGL(dcsPackKernel<16>, dim3(blocks), dim3(256), 0, b->stream, dS, b->nChunks, dP, dRecords, dBlob, blobLen, b->dPackages, b->imgDw);
    else if (b->fpw == 8)
        hipLaunchKernelGGL(dcsPackKernel<8>, dim3(blocks), dim3(256), 0, b->stream, dS, b->nChunks, dP, dRecords, dBlob, blobLen, b->dPackages, b->imgDw);
    else
        hipLaunchKernelGGL(dcsPackKernel<4>, dim3(blocks), dim3(256), 0, b->stream, dS, b->nChunks, dP, dRecords, dBlob, blobLen, b->dPackages, b->imgDw);
    HIPCHK(ctx, hipGetLastError());
    return DCS_OK;
}

// Plan AND pack on the device (dcsPlanKernel above): nothing of the list's index results is needed on the host.  `table`
// describes the streams (host memory; copied), dRecords / dInfos are what the index kernel wrote or is still writing on
// `stream`, dBlob the streams as uploaded.  The planner's flag word (DCS_PLAN_*) is copied to *flagOut (pinned memory of the
// batch) behind the decode launch by batchQueuePlanFlag; a non-zero flag means the PCM of this batch is not to be used.
static DcsStatus createBatchPlannedOnDevice(DcsCtx *ctx, const DcsPlanStream *table, uint32_t nStreams, uint32_t extraFrames, uint32_t nJobs,
                                            uint32_t nRecords, bool all94, bool has93aT1, uint64_t payloadBytes, const DcsFrameIndex *dRecords,
                                            const DcsStreamInfo *dInfos, const uint8_t *dBlob, uint64_t blobLen, hipStream_t stream, DcsBatch **out,
                                            int framesPerChunk = 0)
{
    *out = nullptr;
    if (nStreams == 0 || nJobs == 0)
        return DCS_ERR_INVALID_ARG;
    DcsBatch *b = new (std::nothrow) DcsBatch;
    if (b == nullptr)
        return DCS_ERR_NO_MEMORY;
    b->ctx = ctx;
    b->stream = stream ? stream : ctx->stream;
    b->nJobs = nJobs; b->nSrcs = nRecords;
    b->fpw = chooseFpw(ctx, nJobs, all94);
    if (has93aT1)
        b->flags |= DCS_BATCH_HAS_93A_T1;
    static const bool xcdRanges = getenv("DCS_PIPE_XCD_RANGES") == nullptr || atoi(getenv("DCS_PIPE_XCD_RANGES")) != 0;
    if (xcdRanges && !ctx->noXcdRanges)
        b->flags |= DCS_BATCH_XCD_RANGES;       // (the arithmetic plan IS chain order: chunk c takes its tail from chunk c - 1)
    const uint64_t pcm = static_cast<uint64_t>(nJobs) * DCS_FRAME_SAMPLES * 2;
    b->algoBytes = payloadBytes + static_cast<uint64_t>(nRecords) * 56u + pcm;
    b->abiBytes = payloadBytes + static_cast<uint64_t>(nRecords) * sizeof(DcsSrcDesc) + static_cast<uint64_t>(nJobs) * sizeof(DcsFrameJob) + pcm;
    // (framesPerChunk: fewer frames per chunk than the kernel variant has slots -- the pipeline's second attempt for a list whose
    // frames are too large for a full chunk's bit pool; the context's diagnostic setting applies as well)
    {
        int fpc = framesPerChunk > 0 ? framesPerChunk : ctx->framesPerChunk;
        b->planFpc = static_cast<uint32_t>(fpc >= 1 && fpc < b->fpw ? fpc : b->fpw);
    }
    b->nChunks = (nJobs + b->planFpc - 1) / b->planFpc;
    b->imgDw = dcsPoolCapacity(b->fpw) | (all94 ? DCS_PKG_SPLIT4 : 0u);     // (a plan made on the device: the full image, dcs_common.h)
    const size_t pkgBytes = static_cast<size_t>(b->nChunks) * dcsPkgStride(b->fpw, b->imgDw);
    void *dTable = nullptr;
    const size_t tableBytes = sizeof(DcsPlanStream) * nStreams;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        b->planSlotsCap = sizeof(DcsSlot) * static_cast<size_t>(b->nChunks) * static_cast<size_t>(b->fpw);
        b->planSrcsCap = sizeof(DcsPlanSrc) * (nRecords ? nRecords : 1);
        HIPCHK(ctx, cacheAlloc(ctx, false, &b->dPlanSlots, b->planSlotsCap));
        HIPCHK(ctx, cacheAlloc(ctx, false, &b->dPlanSrcs, b->planSrcsCap));
        // the stream table through pinned staging (a copy from pageable memory holds the calling thread), its device copy
        // behind it in the same allocation is not needed: the table is small, it rides in the slots' buffer's neighbour
        b->hStageCap = tableBytes + 16;
        HIPCHK(ctx, cacheAlloc(ctx, true, &b->hStage, b->hStageCap));
        memcpy(static_cast<uint8_t *>(b->hStage) + 16, table, tableBytes);
        memset(b->hStage, 0, 16);
        b->cap[2] = tableBytes;
        HIPCHK(ctx, cacheAlloc(ctx, false, &dTable, tableBytes));
        b->dTable = dTable;
        HIPCHK(ctx, copyByKernel(b->stream, dTable, static_cast<uint8_t *>(b->hStage) + 16, tableBytes));
        b->cap[9] = pkgBytes; HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dPackages), b->cap[9]));
        // (PCM and error words in one allocation: one copy brings both down)
        b->errJoined = true;
        b->cap[4] = sizeof(int16_t) * DCS_FRAME_SAMPLES * nJobs + sizeof(uint32_t) * nJobs; b->cap[5] = 0;
        HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dPcm), b->cap[4]));
        b->dErr = reinterpret_cast<uint32_t *>(b->dPcm + static_cast<size_t>(DCS_FRAME_SAMPLES) * nJobs);
        // (no tails kept: nothing reads them on this path, and a buffer from the recycling cache would carry another list's -- ADVICE r5)
        b->cap[8] = sizeof(unsigned long long) * 16 * (b->nChunks + 1); HIPCHK(ctx, cacheAlloc(ctx, false, reinterpret_cast<void **>(&b->dHandoff), b->cap[8]));
        {
            const DcsStatus sq = queuePlanAndPack(b, nStreams, extraFrames, dRecords, dInfos, dBlob, blobLen, nullptr);
            if (sq != DCS_OK)
                return sq;
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventCreate(&b->ev0));
        HIPCHK(ctx, hipEventCreate(&b->ev1));
        HIPCHK(ctx, hipEventCreateWithFlags(&b->evDone, hipEventDisableTiming | (tlsBlockingWaits ? hipEventBlockingSync : 0u)));
        return DCS_OK;
    }();
    if (st != DCS_OK)
    {
        (void)streamWait(ctx, b->stream);
        dcs_batch_destroy(b);
        return st;
    }
    *out = b;
    return DCS_OK;
}

// queue the copy of the planner's flag word into the batch's pinned staging (behind whatever runs on the batch's stream);
// read it with batchPlanFlag once the stream has been waited for
static DcsStatus batchQueuePlanFlag(DcsBatch *b)
{
    // (nothing to queue since round 4: the planner stores a non-zero flag into the pinned word itself)
    (void)b;
    return DCS_OK;
}
static uint32_t batchPlanFlag(const DcsBatch *b) { return *static_cast<const volatile uint32_t *>(b->hStage); }

// Diagnostic / test entry: the packages of `jobs` as the DEVICE packer lays them out (plan from source digests on the
// host, pack kernel on the device), for comparison with dcs_pack_chunks, the host packer.  out = NULL to size.
extern "C" DcsStatus dcs_pack_chunks_device(DcsCtx *ctx, const DcsFrameJob *jobs, uint32_t nJobs, const DcsSrcDesc *srcs, uint32_t nSrcs,
                                            const uint8_t *blob, size_t blobLen, int fpw,
                                            uint8_t *out, size_t cap, uint32_t *nChunksOut, uint32_t *packageBytesOut)
{
    if (ctx == nullptr || jobs == nullptr || srcs == nullptr || blob == nullptr || nChunksOut == nullptr || !(fpw == 4 || fpw == 8 || fpw == 16))
        return DCS_ERR_INVALID_ARG;
    std::vector<DcsPlanSrc> ps(nSrcs);
    std::vector<DcsFrameIndex> recs(nSrcs);
    for (uint32_t k = 0 ; k < nSrcs ; ++k)
    {
        const DcsSrcDesc &sd = srcs[k];
        ps[k] = DcsPlanSrc{ sd.streamOff, sd.idx.bitOff, sd.idx.nBits, sd.hdrLen, sd.idx.nBands, sd.idx.flags, sd.format, sd.mixMul, k };
        recs[k] = sd.idx;
    }
    std::vector<DcsSlot> slots;
    uint32_t imgDw = 0;
    const uint32_t nChunks = dcsPlanChunksCappedLite(jobs, nJobs, ps.data(), fpw, slots, true, 0, true, false, &imgDw, DCS_MI355X_WAVE_PLACES);
    *nChunksOut = nChunks;
    if (dcsAllSources94(jobs, nJobs, srcs))
        imgDw |= DCS_PKG_SPLIT4;
    if (packageBytesOut != nullptr)
        *packageBytesOut = dcsPkgStride(fpw, imgDw);
    if (out == nullptr)
        return DCS_OK;
    const size_t pkgBytes = static_cast<size_t>(nChunks) * dcsPkgStride(fpw, imgDw);
    if (cap < pkgBytes)
        return DCS_ERR_CAPACITY;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void *dSlots = nullptr, *dPs = nullptr, *dRecs = nullptr, *dBlob = nullptr, *dPkg = nullptr;
    const size_t blobAlloc = ((blobLen + 3) & ~size_t(3)) + 64;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipMalloc(&dSlots, sizeof(DcsSlot) * slots.size()));
        HIPCHK(ctx, hipMalloc(&dPs, sizeof(DcsPlanSrc) * (nSrcs ? nSrcs : 1)));
        HIPCHK(ctx, hipMalloc(&dRecs, sizeof(DcsFrameIndex) * (nSrcs ? nSrcs : 1)));
        HIPCHK(ctx, hipMalloc(&dBlob, blobAlloc));
        HIPCHK(ctx, hipMalloc(&dPkg, pkgBytes));
        HIPCHK(ctx, hipMemsetAsync(dBlob, 0, blobAlloc, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(dPkg, 0xA5, pkgBytes, ctx->stream));     // (the pack kernel writes every byte: a pattern it must leave nothing of)
        HIPCHK(ctx, hipMemcpyAsync(dSlots, slots.data(), sizeof(DcsSlot) * slots.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(dPs, ps.data(), sizeof(DcsPlanSrc) * nSrcs, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(dRecs, recs.data(), sizeof(DcsFrameIndex) * nSrcs, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(dBlob, blob, blobLen, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t blocks = (nChunks + 3) / 4;
        const DcsSlot *dS = static_cast<const DcsSlot *>(dSlots);
        const DcsPlanSrc *dP = static_cast<const DcsPlanSrc *>(dPs);
        const DcsFrameIndex *dR = static_cast<const DcsFrameIndex *>(dRecs);
        const uint8_t *dB = static_cast<const uint8_t *>(dBlob);
        uint8_t *dO = static_cast<uint8_t *>(dPkg);
        if (fpw == 16)     hipLaunchKernelGGL(dcsPackKernel<16>, dim3(blocks), dim3(256), 0, ctx->stream, dS, nChunks, dP, dR, dB, blobLen, dO, imgDw);
        else if (fpw == 8) hipLaunchKernelGGL(dcsPackKernel<8>, dim3(blocks), dim3(256), 0, ctx->stream, dS, nChunks, dP, dR, dB, blobLen, dO, imgDw);
        else               hipLaunchKernelGGL(dcsPackKernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, dS, nChunks, dP, dR, dB, blobLen, dO, imgDw);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(out, dPkg, pkgBytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return DCS_OK;
    }();
    (void)hipStreamSynchronize(ctx->stream);
    for (void *q : { dSlots, dPs, dRecs, dBlob, dPkg })
        if (q) (void)hipFree(q);
    return st;
}

extern "C" DcsStatus dcs_batch_create(DcsCtx *ctx,
                                      const uint8_t *blob, size_t blobLen,
                                      const DcsSrcDesc *srcs, uint32_t nSrcs,
                                      const DcsFrameJob *jobs, uint32_t nJobs,
                                      const int16_t *tailsIn, uint32_t nTailsIn,
                                      DcsBatch **out)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    struct Resident { bool old; Resident() : old(tlsResidentBatch) { tlsResidentBatch = true; } ~Resident() { tlsResidentBatch = old; } } resident;
    return createBatch(ctx, blob, blobLen, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, nullptr, ctx->handoff, out);
}

template <int FPW>
static hipError_t launch(const DcsKernelArgs &args, hipStream_t stream, int numCUs)
{
    uint32_t blocks = (args.nChunks + dcsk::kWavesPerBlock - 1) / dcsk::kWavesPerBlock;
    if (args.flags & DCS_BATCH_XCD_RANGES)
        blocks = (blocks + 7u) / 8u * 8u;           // eight ranges of equal length; the padding workgroups find no chunk and leave
    // (priorities, dcs_kernels.hip.h: four wavefronts of this kernel per SIMD, CUs x 16 resident at once; with no more than CUs x 4
    // chunks every wavefront has a SIMD to itself and there is nothing to arrange)
    const uint32_t cus = static_cast<uint32_t>(numCUs);
    uint32_t flags = args.flags;
    if (args.nChunks > cus * 4u && cus % 8u == 0 && cus / 8u < 256u)
        flags |= DCS_BATCH_PACED | ((cus / 8u) << DCS_BATCH_CUS8_SHIFT);
    dcsk::dcsDecodeKernel<FPW><<<dim3(blocks), dim3(64 * dcsk::kWavesPerBlock), dcsk::ldsBytes(FPW), stream>>>(
        args.packages, args.tables, args.nChunks, flags, args.epoch, args.nJobs, args.pcm, args.handoff, args.err, args.tailsOut,
        args.blob, args.blobLen, args.srcs, args.tailsIn, args.debug);
    return hipGetLastError();
}

// one launch, without the completion event (the callers below record it once per call)
static DcsStatus launchOnce(DcsBatch *b, hipStream_t stream)
{
    DcsCtx *ctx = b->ctx;
    b->settled = false;
    if (++b->epoch > DCS_EPOCH_MAX)
        b->epoch = 1;                   // (0 marks words no launch has written; after 2^31 launches of ONE batch the count starts over)
    const DcsKernelArgs args = kernelArgs(b);
    hipError_t e;
    e = (b->fpw == 16) ? launch<16>(args, stream, ctx->numCUs) : (b->fpw == 8) ? launch<8>(args, stream, ctx->numCUs) : launch<4>(args, stream, ctx->numCUs);
    if (e != hipSuccess)
    {
        setError(ctx, std::string("kernel launch failed: ") + hipGetErrorString(e));
        return DCS_ERR_HIP;
    }
    return DCS_OK;
}

static DcsStatus markLaunched(DcsBatch *b, hipStream_t stream)
{
    b->settled = false;
    HIPCHK(b->ctx, hipEventRecord(b->evDone, stream));
    b->launched = true;
    return DCS_OK;
}

extern "C" DcsStatus dcs_batch_run(DcsBatch *b, void *hipStream)
{
    if (b == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = hipStream ? static_cast<hipStream_t>(hipStream) : b->stream;
    const DcsStatus st = launchOnce(b, stream);
    return st != DCS_OK ? st : markLaunched(b, stream);
}

// `count` launches back to back without returning to the caller in between (a caller that drives the steps from an
// interpreted language would otherwise be slower per launch than the kernel of a small batch)
extern "C" DcsStatus dcs_batch_run_many(DcsBatch *b, void *hipStream, int count)
{
    if (b == nullptr || count < 0)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = hipStream ? static_cast<hipStream_t>(hipStream) : b->stream;
    for (int i = 0 ; i < count ; ++i)
    {
        const DcsStatus st = launchOnce(b, stream);
        if (st != DCS_OK)
            return st;
    }
    return count ? markLaunched(b, stream) : DCS_OK;
}

extern "C" DcsStatus dcs_batch_time(DcsBatch *b, void *hipStream, int iters, float *avgMs)
{
    if (b == nullptr || iters < 1 || avgMs == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = hipStream ? static_cast<hipStream_t>(hipStream) : b->stream;
    HIPCHK(ctx, hipEventRecord(b->ev0, stream));
    for (int i = 0 ; i < iters ; ++i)
    {
        DcsStatus st = launchOnce(b, stream);
        if (st != DCS_OK)
            return st;
    }
    HIPCHK(ctx, hipEventRecord(b->ev1, stream));
    {
        const DcsStatus st = markLaunched(b, stream);
        if (st != DCS_OK)
            return st;
    }
    HIPCHK(ctx, hipEventSynchronize(b->ev1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *avgMs = ms / static_cast<float>(iters);
    return DCS_OK;
}

// `iters` launches dealt round-robin to `n` resident batches on one stream, bracketed by HIP events: with batches whose
// packages and PCM together exceed the 256 MB Infinity Cache no launch finds its inputs or its outputs' lines in it
extern "C" DcsStatus dcs_batch_time_rotating(DcsBatch *const *batches, uint32_t n, void *hipStream, int iters, float *avgMs)
{
    if (batches == nullptr || n == 0 || iters < 1 || avgMs == nullptr)
        return DCS_ERR_INVALID_ARG;
    for (uint32_t k = 0 ; k < n ; ++k)
        if (batches[k] == nullptr || batches[k]->ctx != batches[0]->ctx)
            return DCS_ERR_INVALID_ARG;
    DcsBatch *b0 = batches[0];
    DcsCtx *ctx = b0->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = hipStream ? static_cast<hipStream_t>(hipStream) : b0->stream;
    HIPCHK(ctx, hipEventRecord(b0->ev0, stream));
    for (int i = 0 ; i < iters ; ++i)
    {
        const DcsStatus st = launchOnce(batches[static_cast<uint32_t>(i) % n], stream);
        if (st != DCS_OK)
            return st;
    }
    HIPCHK(ctx, hipEventRecord(b0->ev1, stream));
    for (uint32_t k = 0 ; k < n ; ++k)
    {
        const DcsStatus st = markLaunched(batches[k], stream);
        if (st != DCS_OK)
            return st;
    }
    HIPCHK(ctx, hipEventSynchronize(b0->ev1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, b0->ev0, b0->ev1));
    *avgMs = ms / static_cast<float>(iters);
    return DCS_OK;
}

extern "C" DcsStatus dcs_batch_sync(DcsBatch *b)
{
    if (b == nullptr)
        return DCS_ERR_INVALID_ARG;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    HIPCHK(b->ctx, waitLaunched(b));
    HIPCHK(b->ctx, streamWait(b->ctx, b->stream));
    b->settled = true;
    return DCS_OK;
}

extern "C" DcsStatus dcs_batch_download(DcsBatch *b, int16_t *pcmOut, uint32_t *errOut, int16_t *tailsOut)
{
    if (b == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, waitLaunched(b));
    HIPCHK(ctx, streamWait(b->ctx, b->stream));
    b->settled = true;          // (the copies below are synchronous)
    if (pcmOut)
        HIPCHK(ctx, hipMemcpy(pcmOut, b->dPcm, sizeof(int16_t) * DCS_FRAME_SAMPLES * b->nJobs, hipMemcpyDeviceToHost));
    if (errOut)
        HIPCHK(ctx, hipMemcpy(errOut, b->dErr, sizeof(uint32_t) * b->nJobs, hipMemcpyDeviceToHost));
    if (tailsOut && b->dTailsOut)
        HIPCHK(ctx, hipMemcpy(tailsOut, b->dTailsOut, sizeof(int16_t) * 16 * b->nJobs, hipMemcpyDeviceToHost));
    return DCS_OK;
}

// PCM and error words in PINNED host memory owned by the batch (valid until it is destroyed or run again):
// the device-to-host copy runs at link speed and the caller reads the result in place.
extern "C" DcsStatus dcs_batch_download_view(DcsBatch *b, const int16_t **pcmOut, const uint32_t **errOut)
{
    if (b == nullptr || pcmOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t pcmBytes = sizeof(int16_t) * DCS_FRAME_SAMPLES * b->nJobs, errBytes = sizeof(uint32_t) * b->nJobs;
    // (the error words lie behind the PCM on the device: mirror that in pinned memory and bring both down in one copy -- unless
    // dcs_batch_download has given this batch separate host buffers before)
    const bool joined = b->errJoined && (b->hPcm == nullptr
                                         || b->hErr == reinterpret_cast<uint32_t *>(b->hPcm + static_cast<size_t>(DCS_FRAME_SAMPLES) * b->nJobs));
    if (b->hPcm == nullptr)
    {
        b->hCap[0] = joined ? pcmBytes + errBytes : pcmBytes;
        HIPCHK(ctx, cacheAlloc(ctx, true, reinterpret_cast<void **>(&b->hPcm), b->hCap[0]));
        if (joined)
        {
            b->hErr = reinterpret_cast<uint32_t *>(b->hPcm + static_cast<size_t>(DCS_FRAME_SAMPLES) * b->nJobs);
            b->hCap[1] = 0;
        }
    }
    if (b->hErr == nullptr && errOut != nullptr)
    {
        b->hCap[1] = errBytes;
        HIPCHK(ctx, cacheAlloc(ctx, true, reinterpret_cast<void **>(&b->hErr), errBytes));
    }
    if (b->launched)
        HIPCHK(ctx, hipStreamWaitEvent(b->stream, b->evDone, 0));      // the copies follow the last launch, whatever stream it ran on
    const size_t firstBytes = joined ? pcmBytes + errBytes : pcmBytes;
    if (b->downByKernel)
    {
        HIPCHK(ctx, copyByKernel(b->stream, b->hPcm, b->dPcm, firstBytes, b->downBlocks));
        if (errOut != nullptr && !joined)
            HIPCHK(ctx, copyByKernel(b->stream, b->hErr, b->dErr, errBytes, b->downBlocks));
    }
    else
    {
        HIPCHK(ctx, hipMemcpyAsync(b->hPcm, b->dPcm, firstBytes, hipMemcpyDeviceToHost, b->stream));
        if (errOut != nullptr && !joined)
            HIPCHK(ctx, hipMemcpyAsync(b->hErr, b->dErr, errBytes, hipMemcpyDeviceToHost, b->stream));
    }
    HIPCHK(ctx, streamWait(b->ctx, b->stream));
    b->settled = true;          // (uploads and pack kernels ran on b->stream before, the launches are behind evDone)
    *pcmOut = b->hPcm;
    if (errOut != nullptr)
        *errOut = b->hErr;
    return DCS_OK;
}

extern "C" void *dcs_batch_device_pcm(DcsBatch *b) { return b ? b->dPcm : nullptr; }

#ifdef DCS_STAMPS
// diagnostic builds only: copy the per-chunk phase stamps (16 x uint64 per chunk) to the host
extern "C" int dcs_debug_stamps(DcsBatch *b, unsigned long long *out, uint32_t capChunks)
{
    if (b == nullptr || b->dDebug == nullptr) return -1;
    (void)streamWait(b->ctx, b->stream);
    const uint32_t n = b->nChunks < capChunks ? b->nChunks : capChunks;
    (void)hipMemcpy(out, b->dDebug, sizeof(unsigned long long) * 16 * n, hipMemcpyDeviceToHost);
    return static_cast<int>(n);
}
#endif
extern "C" uint64_t dcs_batch_algorithmic_bytes(const DcsBatch *b) { return b ? b->algoBytes : 0; }
extern "C" uint64_t dcs_batch_abi_bytes(const DcsBatch *b) { return b ? b->abiBytes : 0; }
extern "C" uint32_t dcs_batch_num_chunks(const DcsBatch *b) { return b ? b->nChunks : 0; }
extern "C" int dcs_batch_frames_per_wave(const DcsBatch *b) { return b ? b->fpw : 0; }
extern "C" uint32_t dcs_batch_num_jobs(const DcsBatch *b) { return b ? b->nJobs : 0; }

// ---------------------------------------------------------------------------------------------------------
// The context's LIVE decoder: dcs_decode_batch for a caller that comes back every few frames (DCSDecoderHIP's sample pump, the
// sequencer, small one-shot calls).  Round 5's one-shot call made a batch object per call: a planner and packer run into freshly
// borrowed buffers, three events, three memsets, an upload, the launch and two synchronous downloads -- 80 us for ONE frame, when
// the reference's whole pump takes 4 us a frame.  Here everything that can outlive a call does:
//   * one pinned arena for what goes up (the external tails, the descriptors of multi-channel frames, the chunk packages) and one
//     for what comes down (PCM, error words, every frame's tail), grown geometrically, never given back before the context goes;
//   * small batches are not copied at all: the kernel reads its packages from the pinned arena and writes its PCM into the
//     pinned arena over the link (a package is read once, 16 bytes per lane; the PCM of a few dozen frames is a few kilobytes),
//     so a call is ONE launch and ONE wait.  Larger ones get one copy up, one copy down (PCM, error words and tails are one
//     block), queued with the launch and waited for once;
//   * the hand-off words live as long as the context and are told apart by a launch counter (epoch) that only grows, so nothing
//     is cleared per call; the kernel writes every error word itself;
//   * the streams behind the SECOND and later sources of multi-channel frames stay resident: a caller that names its blob
//     (blobId != 0, append-only under that name) has every byte of it uploaded once, when it first appears.
// Results are handed out as pointers into the pinned arena, valid until the context's next live call.
// ---------------------------------------------------------------------------------------------------------
struct DcsLive
{
    uint8_t *hUp = nullptr, *dUp = nullptr, *hDown = nullptr, *dDown = nullptr, *dBlob = nullptr;
    size_t hUpCap = 0, dUpCap = 0, hDownCap = 0, dDownCap = 0, blobCap = 0;
    unsigned long long *dHandoff = nullptr;
    size_t handoffChunks = 0;
    uint32_t epoch = 0;
    size_t blobResident = 0;            // bytes of the named blob that are on the device
    size_t blobDirty = 0;               // bytes of the device blob written since it was last cleared
    uint64_t blobId = 0;
    std::vector<DcsSlot> slots;
    // what a call may leave to the link instead of a copy (bytes up, frames down); DCS_LIVE_ZC_UP_KB / DCS_LIVE_ZC_DOWN_FRAMES.
    // Measured (tools/live_sweep.py, profiles/r06_live_sweep.txt): up to 2 000 frames a call no copy pays in either direction
    // (1 frame 21 us, 64 frames 26.5 against 36-39 with copies, 2 000 frames 176 against 187); beyond a megabyte the copy engines take over.
    size_t zcUpBytes = size_t(1) << 20;
    uint32_t zcDownFrames = 2048;
    // DCS_LIVE_STATS=1: where the calls' time went, printed when the context goes
    struct { double validateUs = 0, planUs = 0, packUs = 0, queueUs = 0, waitUs = 0; unsigned long long calls = 0, frames = 0; } stats;
};
static const bool g_liveStats = getenv("DCS_LIVE_STATS") != nullptr && atoi(getenv("DCS_LIVE_STATS")) != 0;

static void liveDestroy(DcsCtx *ctx)
{
    DcsLive *l = ctx->live;
    if (l == nullptr)
        return;
    (void)hipStreamSynchronize(ctx->stream);
    if (g_liveStats)
        fprintf(stderr, "live decoder of context %p: %llu calls, %llu frames; validate %.1f us, plan %.1f, pack %.1f, queue %.1f, wait %.1f\n", static_cast<void *>(ctx),
                l->stats.calls, l->stats.frames, l->stats.validateUs, l->stats.planUs, l->stats.packUs, l->stats.queueUs, l->stats.waitUs);
    if (l->hUp) (void)hipHostFree(l->hUp);
    if (l->hDown) (void)hipHostFree(l->hDown);
    for (void *p : { static_cast<void *>(l->dUp), static_cast<void *>(l->dDown), static_cast<void *>(l->dBlob), static_cast<void *>(l->dHandoff) })
        if (p) (void)hipFree(p);
    delete l;
    ctx->live = nullptr;
}

// room for `bytes` in one of the live arenas (pinned or device); what it held is not kept
static hipError_t liveRoom(uint8_t **buf, size_t *cap, size_t bytes, bool pinned, size_t first = 0)
{
    if (bytes <= *cap)
        return hipSuccess;
    // (`first`: what the arena starts with.  A decoder's look-ahead grows 64 -> 512 -> 4 096 frames within its first three calls, and
    // every step used to free and pin the arenas again -- 0.7-1.5 ms of a new context's first stream, NOTES 44)
    if (*buf == nullptr && bytes < first)
        bytes = first;
    if (*buf != nullptr)
    {
        (void)(pinned ? hipHostFree(*buf) : hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    const size_t want = std::max((bytes + 65535) & ~size_t(65535), *cap * 2);
    void *p = nullptr;
    hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    if (e != hipSuccess && want != ((bytes + 65535) & ~size_t(65535)))
    {
        (void)hipGetLastError();
        e = pinned ? hipHostMalloc(&p, (bytes + 65535) & ~size_t(65535), hipHostMallocDefault) : hipMalloc(&p, (bytes + 65535) & ~size_t(65535));
    }
    if (e != hipSuccess)
        return e;
    *buf = static_cast<uint8_t *>(p);
    *cap = want;
    return hipSuccess;
}

// the pinned arenas' first size: what a look-ahead of 4 096 frames needs (packages up ~ 340 B a frame, PCM + error word + tail down)
static const size_t kLiveFirstUp = size_t(3) << 19, kLiveFirstDown = size_t(9) << 18;
// batches beyond this go the resident-batch way (buffers from the context's bounded cache): the live arenas never shrink
static const uint32_t kLiveMaxJobs = 1u << 17;

static DcsStatus decodeLive(DcsCtx *ctx, const uint8_t *blob, size_t blobLen, uint64_t blobId,
                            const DcsSrcDesc *srcs, uint32_t nSrcs, const DcsFrameJob *jobs, uint32_t nJobs,
                            const int16_t *tailsIn, uint32_t nTailsIn,
                            const int16_t **pcmOut, const uint32_t **errOut, const int16_t **tailsOut)
{
    uint64_t payloadBits = 0;
    uint32_t batchFlags = 0;
    const double tv0 = g_liveStats ? hipchkNow() : 0.0;
    {
        const DcsStatus vst = validateBatch(ctx, blobLen, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, &payloadBits, &batchFlags);
        if (vst != DCS_OK)
            return vst;
    }
    const double tv1 = g_liveStats ? hipchkNow() : 0.0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->live == nullptr)
    {
        ctx->live = new (std::nothrow) DcsLive;
        if (ctx->live == nullptr)
            return DCS_ERR_NO_MEMORY;
        if (const char *v = getenv("DCS_LIVE_ZC_UP_KB")) ctx->live->zcUpBytes = static_cast<size_t>(atol(v)) << 10;
        if (const char *v = getenv("DCS_LIVE_ZC_DOWN_FRAMES")) ctx->live->zcDownFrames = static_cast<uint32_t>(atol(v));
    }
    DcsLive *l = ctx->live;
    bool all94 = true, multi = false;
    for (uint32_t j = 0 ; j < nJobs ; ++j)
    {
        all94 = all94 && jobs[j].xform == DCS_XFORM_94;
        multi = multi || jobs[j].nSrc > 1;
    }
    const int fpw = chooseFpw(ctx, nJobs, all94);
    const bool split4 = dcsAllSources94(jobs, nJobs, srcs);

    // the streams behind further sources: resident under the caller's name for the blob, else uploaded for this call
    if (multi)
    {
        const size_t need = ((blobLen + 3) & ~size_t(3)) + 64;          // zero tail: the bit reader looks past the end
        const bool reuse = blobId != 0 && blobId == l->blobId && blobLen >= l->blobResident && need <= l->blobCap;
        if (!reuse)
        {
            if (need > l->blobCap)
            {
                HIPCHK(ctx, liveRoom(&l->dBlob, &l->blobCap, blobId != 0 ? need * 2 : need, false));
                HIPCHK(ctx, hipMemsetAsync(l->dBlob, 0, l->blobCap, ctx->stream));
            }
            else if (l->blobDirty != 0)                                 // (what another blob left there must not show behind this one's end)
                HIPCHK(ctx, hipMemsetAsync(l->dBlob, 0, std::min(l->blobCap, ((l->blobDirty + 3) & ~size_t(3)) + 64), ctx->stream));
            l->blobDirty = 0;
            l->blobResident = 0;
            l->blobId = blobId;
        }
        if (blobLen > l->blobResident)
        {
            HIPCHK(ctx, hipMemcpyAsync(l->dBlob + l->blobResident, blob + l->blobResident, blobLen - l->blobResident, hipMemcpyHostToDevice, ctx->stream));
            l->blobDirty = std::max(l->blobDirty, blobLen);
            if (blobId != 0)
                l->blobResident = blobLen;                              // (an unnamed blob is nobody's next time)
        }
    }

    const size_t pcmBytes = static_cast<size_t>(nJobs) * DCS_FRAME_SAMPLES * sizeof(int16_t), errBytes = static_cast<size_t>(nJobs) * sizeof(uint32_t);
    const size_t downBytes = pcmBytes + errBytes + static_cast<size_t>(nJobs) * 16 * sizeof(int16_t);
    const bool zcDown = nJobs <= l->zcDownFrames;
    HIPCHK(ctx, liveRoom(&l->hDown, &l->hDownCap, downBytes, true, kLiveFirstDown));
    if (!zcDown)
        HIPCHK(ctx, liveRoom(&l->dDown, &l->dDownCap, downBytes, false));

    {
        const bool handoff = ctx->handoff;
        const double tp0 = g_liveStats ? hipchkNow() : 0.0;
        const uint32_t nChunks = dcsPlanChunks(jobs, nJobs, srcs, fpw, l->slots, handoff, ctx->framesPerChunk, false, true, ctx->shuffleSeed);
        const double tp1 = g_liveStats ? hipchkNow() : 0.0;
        const uint32_t layout = dcsImageDwords(l->slots.data(), nChunks, fpw) | (split4 ? DCS_PKG_SPLIT4 : 0u);

        // what goes up, in one block: external tails | descriptors (only multi-channel frames read them) | chunk packages
        const size_t tailBytes = static_cast<size_t>(nTailsIn) * 16 * sizeof(int16_t);
        const size_t offSrcs = (tailBytes + 255) & ~size_t(255);
        const size_t srcBytes = multi ? static_cast<size_t>(nSrcs) * sizeof(DcsSrcDesc) : 0;
        const size_t offPkg = (offSrcs + srcBytes + 255) & ~size_t(255);
        const size_t pkgBytes = static_cast<size_t>(nChunks) * dcsPkgStride(fpw, layout);
        const size_t upBytes = offPkg + pkgBytes;
        HIPCHK(ctx, liveRoom(&l->hUp, &l->hUpCap, upBytes, true, kLiveFirstUp));
        if (tailBytes)
            memcpy(l->hUp, tailsIn, tailBytes);
        if (srcBytes)
            memcpy(l->hUp + offSrcs, srcs, srcBytes);
        dcsBuildPackages(l->slots.data(), nChunks, fpw, srcs, blob, blobLen, l->hUp + offPkg, layout);
        const double tp2 = g_liveStats ? hipchkNow() : 0.0;
        const bool zcUp = upBytes <= l->zcUpBytes;
        if (!zcUp)
        {
            HIPCHK(ctx, liveRoom(&l->dUp, &l->dUpCap, upBytes, false));
            HIPCHK(ctx, hipMemcpyAsync(l->dUp, l->hUp, upBytes, hipMemcpyHostToDevice, ctx->stream));
        }
        if (nChunks + 1 > l->handoffChunks)
        {
            if (l->dHandoff) (void)hipFree(l->dHandoff);
            l->dHandoff = nullptr;
            l->handoffChunks = std::max<size_t>(size_t(nChunks) + 1, std::max<size_t>(l->handoffChunks * 2, 1024));
            HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&l->dHandoff), l->handoffChunks * 16 * sizeof(unsigned long long)));
            HIPCHK(ctx, hipMemsetAsync(l->dHandoff, 0, l->handoffChunks * 16 * sizeof(unsigned long long), ctx->stream));
        }
        if (++l->epoch > DCS_EPOCH_MAX)
        {
            // the launch counter has come round: words of 2^32 launches ago must not pass for this launch's
            HIPCHK(ctx, hipMemsetAsync(l->dHandoff, 0, l->handoffChunks * 16 * sizeof(unsigned long long), ctx->stream));
            l->epoch = 1;
        }
        uint8_t *up = zcUp ? l->hUp : l->dUp, *down = zcDown ? l->hDown : l->dDown;
        DcsKernelArgs args;
        args.blob = multi ? l->dBlob : nullptr;
        args.blobLen = multi ? blobLen : 0;
        args.srcs = multi ? reinterpret_cast<const DcsSrcDesc *>(up + offSrcs) : nullptr;
        args.packages = up + offPkg;
        args.nChunks = nChunks;
        args.nJobs = nJobs;
        args.pcm = reinterpret_cast<int16_t *>(down);
        args.err = reinterpret_cast<uint32_t *>(down + pcmBytes);
        args.tailsIn = nTailsIn ? reinterpret_cast<const int16_t *>(up) : nullptr;
        args.tailsOut = reinterpret_cast<int16_t *>(down + pcmBytes + errBytes);
        args.tables = ctx->dTables;
        args.debug = nullptr;
        args.handoff = l->dHandoff;
        args.epoch = l->epoch;
        args.flags = batchFlags | (layout << DCS_BATCH_IMG_SHIFT);
        const hipError_t e = fpw == 16 ? launch<16>(args, ctx->stream, ctx->numCUs) : fpw == 8 ? launch<8>(args, ctx->stream, ctx->numCUs) : launch<4>(args, ctx->stream, ctx->numCUs);
        if (e != hipSuccess)
        {
            setError(ctx, std::string("kernel launch failed: ") + hipGetErrorString(e));
            (void)streamWait(ctx, ctx->stream);
            return DCS_ERR_HIP;
        }
        if (!zcDown)
            HIPCHK(ctx, hipMemcpyAsync(l->hDown, l->dDown, downBytes, hipMemcpyDeviceToHost, ctx->stream));
        const double tp3 = g_liveStats ? hipchkNow() : 0.0;
        HIPCHK(ctx, streamWait(ctx, ctx->stream));
        if (g_liveStats)
        {
            l->stats.planUs += tp1 - tp0; l->stats.packUs += tp2 - tp1; l->stats.queueUs += tp3 - tp2; l->stats.waitUs += hipchkNow() - tp3;
            l->stats.validateUs += tv1 - tv0; l->stats.calls += 1; l->stats.frames += nJobs;
        }
    }
    if (pcmOut) *pcmOut = reinterpret_cast<const int16_t *>(l->hDown);
    if (errOut) *errOut = reinterpret_cast<const uint32_t *>(l->hDown + pcmBytes);
    if (tailsOut) *tailsOut = reinterpret_cast<const int16_t *>(l->hDown + pcmBytes + errBytes);
    return DCS_OK;
}

extern "C" DcsStatus dcs_decode_batch_live(DcsCtx *ctx, const uint8_t *blob, size_t blobLen, uint64_t blobId,
                                           const DcsSrcDesc *srcs, uint32_t nSrcs, const DcsFrameJob *jobs, uint32_t nJobs,
                                           const int16_t *tailsIn, uint32_t nTailsIn,
                                           const int16_t **pcmOut, const uint32_t **errOut, const int16_t **tailsOut)
{
    if (ctx == nullptr || jobs == nullptr || nJobs == 0 || (nSrcs != 0 && (srcs == nullptr || blob == nullptr)))
        return DCS_ERR_INVALID_ARG;
    if (nJobs > kLiveMaxJobs)
    {
        setError(ctx, "dcs_decode_batch_live: more than 131 072 frames in one call (use dcs_decode_batch or a resident batch)");
        return DCS_ERR_CAPACITY;
    }
    return decodeLive(ctx, blob, blobLen, blobId, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, pcmOut, errOut, tailsOut);
}

extern "C" DcsStatus dcs_decode_batch(DcsCtx *ctx,
                                      const uint8_t *blob, size_t blobLen,
                                      const DcsSrcDesc *srcs, uint32_t nSrcs,
                                      const DcsFrameJob *jobs, uint32_t nJobs,
                                      const int16_t *tailsIn, uint32_t nTailsIn,
                                      int16_t *pcmOut, uint32_t *errOut, int16_t *tailsOut)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    if (jobs != nullptr && nJobs != 0 && nJobs <= kLiveMaxJobs && !(nSrcs != 0 && (srcs == nullptr || blob == nullptr)))
    {
        // the context's live decoder: no batch object, nothing allocated, one launch, one wait; the results are copied out
        const int16_t *pcm = nullptr, *tails = nullptr;
        const uint32_t *err = nullptr;
        const DcsStatus st = decodeLive(ctx, blob, blobLen, 0, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, &pcm, &err, &tails);
        if (st != DCS_OK)
            return st;
        if (pcmOut) memcpy(pcmOut, pcm, static_cast<size_t>(nJobs) * DCS_FRAME_SAMPLES * sizeof(int16_t));
        if (errOut) memcpy(errOut, err, static_cast<size_t>(nJobs) * sizeof(uint32_t));
        if (tailsOut) memcpy(tailsOut, tails, static_cast<size_t>(nJobs) * 16 * sizeof(int16_t));
        return DCS_OK;
    }
    // more than the live decoder takes: a batch object with buffers from the context's bounded cache, run once
    // a caller that asks for tailsOut gets the tail EVERY frame leaves (the sequencer resumes from any tick of a batch)
    struct Tails { bool old; explicit Tails(bool all) : old(tlsKeepAllTails) { tlsKeepAllTails = all; } ~Tails() { tlsKeepAllTails = old; } } tails(tailsOut != nullptr);
    DcsBatch *b = nullptr;
    DcsStatus st = createBatch(ctx, blob, blobLen, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, nullptr, ctx->handoff, &b);
    if (st != DCS_OK)
        return st;
    st = dcs_batch_run(b, nullptr);
    if (st == DCS_OK)
        st = dcs_batch_download(b, pcmOut, errOut, tailsOut);
    dcs_batch_destroy(b);
    return st;
}


// ---------------------------------------------------------------------------------------------------------
// Index pass on the device: one wavefront per stream (dcs_index_wave.hip.h).  The walk is serial from frame to frame,
// so a launch takes as long as its longest stream; four streams share a workgroup (and its LDS copy of the tables).
// ---------------------------------------------------------------------------------------------------------
static hipError_t launchIndexWave(hipStream_t stream, uintptr_t blobBase, const DcsStreamLoc *dLocs, uint32_t nStreams, const DcsDevTables *dTables,
                                  DcsFrameIndex *dOut, DcsStreamInfo *dInfos, DcsFrameDigest *dDigest, const dcsidx::StreamOut *dOuts = nullptr)
{
    if (nStreams == 0)          // (an empty round is no launch: a zero-block grid is an error to some runtimes)
        return hipSuccess;
    const uint32_t blocks = (nStreams + dcsidx::kWaves - 1) / dcsidx::kWaves;
    static std::atomic<uint32_t> launches{ 0 };         // (the walks of a launch pace themselves against each other, not against other launches')
    hipLaunchKernelGGL(dcsidx::dcsIndexWaveKernel, dim3(blocks), dim3(dcsidx::kWaves * 64), 0, stream, blobBase, dLocs, nStreams, dTables,
                       dOut, dInfos, dDigest, dOuts, launches.fetch_add(1, std::memory_order_relaxed));
    return hipGetLastError();
}

static hipError_t launchIndex(DcsCtx *ctx)
{
    return launchIndexWave(ctx->stream, reinterpret_cast<uintptr_t>(ctx->dIdxBlob), ctx->dIdxLocs, ctx->idxStreams, ctx->dTables, ctx->dIdxOut,
                           ctx->dIdxInfos, nullptr);
}

extern "C" DcsStatus dcs_index_streams_gpu(DcsCtx *ctx, const uint8_t *blob, size_t blobLen,
                                           const DcsStreamLoc *streams, uint32_t nStreams,
                                           DcsFrameIndex *out, uint64_t outCap, DcsStreamInfo *infos)
{
    if (ctx == nullptr)
        return DCS_ERR_INVALID_ARG;
    if (blob == nullptr || streams == nullptr || nStreams == 0 || out == nullptr || infos == nullptr)
    {
        setError(ctx, "dcs_index_streams_gpu: null argument or no streams");
        return DCS_ERR_INVALID_ARG;
    }
    // every stream must lie in the blob and its records (at most its U16 frame count) in `out`
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        const DcsStreamLoc &l = streams[k];
        if (l.len < 3 || l.off > blobLen || l.len > blobLen - l.off || l.os < DCS_OS93A || l.os > DCS_OS95)
        {
            setError(ctx, "dcs_index_streams_gpu: stream " + std::to_string(k) + " outside the blob or bad OS version");
            return DCS_ERR_INVALID_ARG;
        }
        const uint64_t nf = (static_cast<uint64_t>(blob[l.off]) << 8) | blob[l.off + 1];
        if (l.firstRecord > outCap || nf > outCap - l.firstRecord)
        {
            setError(ctx, "dcs_index_streams_gpu: records of stream " + std::to_string(k) + " do not fit in out");
            return DCS_ERR_CAPACITY;
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (void *p : { static_cast<void *>(ctx->dIdxBlob), static_cast<void *>(ctx->dIdxLocs),
                     static_cast<void *>(ctx->dIdxOut), static_cast<void *>(ctx->dIdxInfos) })
        if (p) (void)hipFree(p);
    ctx->dIdxBlob = nullptr; ctx->dIdxLocs = nullptr; ctx->dIdxOut = nullptr; ctx->dIdxInfos = nullptr;
    ctx->idxStreams = 0;
    const size_t blobAlloc = (blobLen + 3 + 4) & ~size_t(3);
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dIdxBlob), blobAlloc));
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dIdxLocs), sizeof(DcsStreamLoc) * nStreams));
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dIdxOut), sizeof(DcsFrameIndex) * (outCap ? outCap : 1)));
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dIdxInfos), sizeof(DcsStreamInfo) * nStreams));
    HIPCHK(ctx, hipMemsetAsync(reinterpret_cast<uint8_t *>(ctx->dIdxBlob) + (blobAlloc - 8), 0, 8, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dIdxBlob, blob, blobLen, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dIdxLocs, streams, sizeof(DcsStreamLoc) * nStreams, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->dIdxOut, 0, sizeof(DcsFrameIndex) * (outCap ? outCap : 1), ctx->stream));
    ctx->idxBlobLen = blobLen;
    ctx->idxBlobDw = blobAlloc / 4;
    ctx->idxStreams = nStreams;
    ctx->idxCap = outCap;
    HIPCHK(ctx, launchIndex(ctx));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->dIdxOut, sizeof(DcsFrameIndex) * outCap, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(infos, ctx->dIdxInfos, sizeof(DcsStreamInfo) * nStreams, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DCS_OK;
}

#ifdef DCS_IDX_STAMPS
// diagnostic build: the index kernel's cycle accumulators since the last call (tools/index_stamps.py)
extern "C" int dcs_debug_index_stamps(unsigned long long *out12)
{
    unsigned long long zero[16] = { 0 };
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(dcsidx::g_idxStamps), 12 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(dcsidx::g_idxStamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" DcsStatus dcs_index_streams_gpu_time(DcsCtx *ctx, int iters, float *avgMs)
{
    if (ctx == nullptr || avgMs == nullptr || iters < 1)
        return DCS_ERR_INVALID_ARG;
    if (ctx->idxStreams == 0)
    {
        setError(ctx, "dcs_index_streams_gpu_time: no resident index inputs");
        return DCS_ERR_INVALID_ARG;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0 ; i < iters ; ++i)
        HIPCHK(ctx, launchIndex(ctx));
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avgMs = ms / static_cast<float>(iters);
    return DCS_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Shader clock under load, for pricing a kernel in cycles (bench.py's VALU-issue figure): every SIMD of the chip runs
// a dependent integer chain for a few hundred microseconds, each workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around it, and the median ratio is the clock the chip held.  Not part of the decode path.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void dcsClockKernel(unsigned long long *out, int iters, uint32_t seed)
{
    uint32_t x = seed + threadIdx.x, y = blockIdx.x;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0 ; i < iters ; ++i)
    {
        x = x * 0x9E3779B1u + y;
        y = (y ^ (x >> 7)) + 0x7F4A7C15u;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0)
    {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = (r1 - r0) + ((x ^ y) == 0x12345u ? 1u : 0u);     // (keeps the chain alive)
    }
}
}   // namespace

extern "C" DcsStatus dcs_ctx_clock_mhz(DcsCtx *ctx, float *mhzOut)
{
    if (ctx == nullptr || mhzOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint32_t blocks = static_cast<uint32_t>(ctx->numCUs) * 4;
    unsigned long long *d = nullptr;
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long) * 2 * blocks));
    std::vector<unsigned long long> h(2 * static_cast<size_t>(blocks));
    hipError_t e = hipSuccess;
    for (int rep = 0 ; rep < 2 && e == hipSuccess ; ++rep)      // (the first launch only warms the chip up)
    {
        hipLaunchKernelGGL(dcsClockKernel, dim3(blocks), dim3(256), 0, ctx->stream, d, 8000, 12345u + rep);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess)
    {
        setError(ctx, std::string("clock probe failed: ") + hipGetErrorString(e));
        return DCS_ERR_HIP;
    }
    std::vector<double> ratio;
    for (uint32_t k = 0 ; k < blocks ; ++k)
        if (h[2 * k + 1] != 0)
            ratio.push_back(static_cast<double>(h[2 * k]) / static_cast<double>(h[2 * k + 1]));
    if (ratio.empty())
        return DCS_ERR_HIP;
    std::nth_element(ratio.begin(), ratio.begin() + static_cast<long>(ratio.size() / 2), ratio.end());
    *mhzOut = static_cast<float>(ratio[ratio.size() / 2] * 100.0);
    return DCS_OK;
}

// device memory -> pinned host memory, the way the pipelines bring PCM down (hipMemcpyAsync on a stream, one copy of 64 MB at
// a time, five of them timed by HIP events): GB/s.  Not part of the decode path: what the link allows, to hold a sustained
// end-to-end rate against (480 bytes of PCM per frame have to cross it).
extern "C" DcsStatus dcs_ctx_link_rate(DcsCtx *ctx, float *gbpsOut)
{
    if (ctx == nullptr || gbpsOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = size_t(64) << 20;
    void *d = nullptr, *h = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, cacheAlloc(ctx, false, &d, bytes));
        HIPCHK(ctx, cacheAlloc(ctx, true, &h, bytes));
        HIPCHK(ctx, hipEventCreate(&e0));
        HIPCHK(ctx, hipEventCreate(&e1));
        HIPCHK(ctx, hipMemsetAsync(d, 0x5A, bytes, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));       // (first use of the path)
        HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
        for (int i = 0 ; i < 5 ; ++i)
            HIPCHK(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
        *gbpsOut = static_cast<float>(5.0 * static_cast<double>(bytes) / (static_cast<double>(ms) * 1e6));
        return DCS_OK;
    }();
    if (st != DCS_OK)
        (void)hipStreamSynchronize(ctx->stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) cacheFree(ctx, false, d, bytes);
    if (h) cacheFree(ctx, true, h, bytes);
    return st;
}

// What a synchronous call cannot get under on this box, for holding the one-shot calls against (bench.py one_shot): an empty kernel
// launched on the context's stream and waited for, and the same followed by the copy of nFrames x 516 bytes (PCM, error word, tail)
// into pinned memory; medians over `iters` rounds, microseconds of host time.  Not part of the decode path.
namespace {
__global__ void dcsEmptyKernel(uint32_t *p) { if (p != nullptr && threadIdx.x == 1u << 20) *p = 0; }
}   // namespace
extern "C" DcsStatus dcs_ctx_call_floor(DcsCtx *ctx, uint32_t nFrames, int iters, float *launchWaitUs, float *launchCopyWaitUs)
{
    if (ctx == nullptr || iters < 1 || nFrames == 0 || nFrames > (1u << 17))
        return DCS_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(nFrames) * 516;
    void *d = nullptr, *h = nullptr;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, cacheAlloc(ctx, false, &d, bytes));
        HIPCHK(ctx, cacheAlloc(ctx, true, &h, bytes));
        std::vector<double> a, b;
        for (int i = 0 ; i < iters + 3 ; ++i)
        {
            const double t0 = hipchkNow();
            hipLaunchKernelGGL(dcsEmptyKernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<uint32_t *>(nullptr));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            const double t1 = hipchkNow();
            hipLaunchKernelGGL(dcsEmptyKernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<uint32_t *>(nullptr));
            HIPCHK(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            const double t2 = hipchkNow();
            if (i >= 3) { a.push_back(t1 - t0); b.push_back(t2 - t1); }
        }
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        if (launchWaitUs) *launchWaitUs = static_cast<float>(a[a.size() / 2]);
        if (launchCopyWaitUs) *launchCopyWaitUs = static_cast<float>(b[b.size() / 2]);
        return DCS_OK;
    }();
    if (d) cacheFree(ctx, false, d, bytes);
    if (h) cacheFree(ctx, true, h, bytes);
    return st;
}

#include "dcs_pipeline.hip.h"
#include "dcs_device_path.hip.h"
#include "dcs_node.hip.h"

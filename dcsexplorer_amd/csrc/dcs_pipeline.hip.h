// dcs_pipeline.hip.h -- lists of streams in flight: the host preparation of one list (index pass, mixing parameters,
// chunk planner, packer) runs while the GPU decodes another and a third comes back into pinned memory.  Included at
// the end of dcs_runtime.hip (it uses the runtime's batch internals).
//
// The reference decodes its batch job (`--extract-streams`, DCSExplorer.cpp:1742-1907) one stream after the other on
// one thread; here a caller submits lists of whole streams and collects their PCM in submission order.
//
// Shapes, chosen at creation:
//   index pass on the host pool (default): `depth` worker threads each take a submitted list through all its stages on
//     a HIP stream of their own, so the stages of different lists overlap by themselves.
//   index pass on the device (DCS_PIPE_INDEX_ON_DEVICE): the index pass is most of a list's host work (tools/hostbench.cpp:
//     35 of 45 CPU-milliseconds for 65 536 frames), and the host has no cores to spare.  A worker uploads its list's streams
//     (stage A) and hands the list to an INDEXER thread, which walks the streams of the lists that are waiting -- one
//     wavefront per stream, dcs_index_wave.hip.h; a round of up to 2 048 streams takes 2.4 to 2.7 ms whatever their number,
//     every list's records go to buffers of its own -- copies the records back and passes the lists on; a worker then
//     builds, plans, packs, decodes and downloads (stage B).
//   ... and the packer on the device too (DCS_PIPE_PACK_ON_DEVICE): the records do not come back at all.  The indexer
//     copies back an 8-byte digest per frame (bit offset, bit count, band count, flags), the worker plans from that, and
//     a pack kernel assembles the packages from the records and streams that are already resident.
//   ... and the planner (DCS_PIPE_PLAN_ON_DEVICE): nothing of the index results comes back.  A list of whole streams has a
//     regular job list, so one thread per chunk works the chunk plan out (dcsPlanKernel); a worker queues planner, packer,
//     decode kernel and the copy down behind the indexers' round and sleeps until the PCM is there.  Lists the arithmetic
//     plan cannot serve are decoded by the host-planned path (DcsPipelineResult.path).
// The context keeps a pipeline of its own for dcs_decode_streams on a large list (dcsDecodeStreamsInParts below).
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>
#include <atomic>
#include <pthread.h>

struct DcsPipeline
{
    struct Job
    {
        const DcsStreamRef *streams = nullptr;
        uint32_t nStreams = 0, extraFrames = 0;
        std::vector<uint32_t> firstJob;         // first output frame of each stream, and the total
        DcsBatch *batch = nullptr;
        const int16_t *pcm = nullptr;
        const uint32_t *err = nullptr;
        DcsStatus status = DCS_OK;
        bool done = false;
        double hostMs = 0, deviceMs = 0;        // wall time of the threads that worked on the list: host preparation /
                                                // uploads + kernels + downloads (the device index pass counts here)
        // ---- index pass on the device
        bool onDevice = false;                  // records came from the device; the streams lie in hBlob
        uint32_t path = 0;                      // DCS_PIPE_*: the stages of THIS list that ran on the device
        uint8_t *hBlob = nullptr;               // the streams as uploaded, end to end (pinned); the packer reads them
        size_t hBlobCap = 0, hBlobLen = 0;
        void *dBlob = nullptr;                  // the same on the device, for the walk only
        size_t dBlobCap = 0;
        std::vector<DcsStreamLoc> locs;         // offsets relative to the list's blob
        std::vector<uint64_t> firstRecord, streamOff;
        uint64_t totalRec = 0;
        void *hRec = nullptr, *hInfo = nullptr; // records (or, packing on the device, their digests) and stream summaries
        size_t recBytes = 0, infoBytes = 0;     //   as they come back (pinned)
        hipEvent_t uploaded = nullptr;
        // results copied into caller memory by the worker (dcs_decode_streams in parts): optional
        int16_t *pcmDst = nullptr;
        uint32_t *errDst = nullptr;
        // index records the submitter already has (host memory that outlives the job): the index pass is skipped
        const DcsFrameIndex *preRecords = nullptr;
        const uint64_t *preFirstRecord = nullptr;
        const DcsStreamInfo *preInfos = nullptr;
        double tSubmit = 0, tTaken = 0, tQueuedForIndex = 0, tIndexStart = 0, tIndexed = 0, tStageB = 0, tDone = 0;     // (DCS_PIPE_TRACE)
        // what the index round writes for this list (device; fixed sizes per list, so the context's cache serves them): the
        // records -- which stay resident for the device packer --, their digests, the stream summaries
        void *dRec = nullptr, *dDigest = nullptr, *dInfo = nullptr;
        size_t dRecBytes = 0, dDigestBytes = 0;
        const DcsFrameIndex *dRecords = nullptr;    // = dRec once the round has run
        // planner on the device: the list's stream locations and result addresses as the index kernel takes them
    };
    typedef std::shared_ptr<Job> JobPtr;

    DcsCtx *ctx = nullptr;
    int depth = 0;
    size_t roundGather = 0;                 // > 1: an index round waits (briefly) until so many lists are there
    uint32_t flags = 0;                             // DCS_PIPE_*
    std::mutex m;
    std::condition_variable work, indexWork, finished, room;
    std::deque<JobPtr> fresh;                       // submitted, not yet taken by a worker
    std::deque<JobPtr> toIndex;                     // uploaded, waiting for the indexer
    std::deque<JobPtr> indexed;                     // records are back: ready for stage B
    std::deque<JobPtr> order;                       // submitted, not yet collected (submission order)
    JobPtr held;                                    // the list whose result the caller is reading
    std::vector<std::thread> workers;
    std::vector<std::thread> indexers;
    int nWorkers = 0, nUploaders = 0;               // (device index pass: the first nUploaders workers do stage A only)
    std::vector<hipStream_t> streams;               // one per worker, and one more for each indexer
    bool quit = false;
    // dcsDecodeStreamsInParts with the walk shared between host and device: when the last list of either kind was finished
    double lastHostWalkedDone = 0, lastDeviceWalkedDone = 0;
};

static double nowMs()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// DCS_PIPE_TRACE=2: what every pipeline thread did and when (tools/pipe_threads.py reads it from stderr)
// internal flag: the pipeline's threads wait by polling inside the runtime (shortest latency) instead of napping
static constexpr uint32_t kPipeLatency = 0x100u;

static void pipeLog(const char *who, int id, const char *what, double t0, double t1, size_t q1 = 0, size_t q2 = 0)
{
    static const bool on = getenv("DCS_PIPE_TRACE") != nullptr && atoi(getenv("DCS_PIPE_TRACE")) >= 2;
    if (on)
        fprintf(stderr, "pipe thread: %s %d %s %.3f %.3f %zu %zu\n", who, id, what, t0, t1, q1, q2);
}

static void pipelineFreeIndexBuffers(DcsPipeline *p, DcsPipeline::Job *job, bool keepHostBlob)
{
    DcsCtx *ctx = p->ctx;
    job->dRecords = nullptr;
    if (job->dRec) { cacheFree(ctx, false, job->dRec, job->dRecBytes); job->dRec = nullptr; }
    if (job->dDigest) { cacheFree(ctx, false, job->dDigest, job->dDigestBytes); job->dDigest = nullptr; }
    if (job->dInfo) { cacheFree(ctx, false, job->dInfo, job->infoBytes); job->dInfo = nullptr; }
    if (job->dBlob) { cacheFree(ctx, false, job->dBlob, job->dBlobCap); job->dBlob = nullptr; }
    if (job->hRec) { cacheFree(ctx, true, job->hRec, job->recBytes); job->hRec = nullptr; }
    if (job->hInfo) { cacheFree(ctx, true, job->hInfo, job->infoBytes); job->hInfo = nullptr; }
    if (job->hBlob && !keepHostBlob) { cacheFree(ctx, true, job->hBlob, job->hBlobCap); job->hBlob = nullptr; }
    if (job->uploaded) { (void)hipEventDestroy(job->uploaded); job->uploaded = nullptr; }
}

static void pipelineRelease(DcsPipeline *p, DcsPipeline::JobPtr &job)
{
    if (job)
    {
        if (job->batch)
        {
            dcs_batch_destroy(job->batch);
            job->batch = nullptr;
        }
        pipelineFreeIndexBuffers(p, job.get(), false);
    }
    job.reset();
}

static void pipelineFinish(DcsPipeline *p, const DcsPipeline::JobPtr &job, DcsStatus st)
{
    {
        std::lock_guard<std::mutex> lk(p->m);
        job->status = st;
        job->done = true;
        (job->preRecords != nullptr ? p->lastHostWalkedDone : p->lastDeviceWalkedDone) = nowMs();
    }
    p->finished.notify_all();
}

// stage A (device index pass): lay the list's streams end to end in pinned memory and send them up
static DcsStatus pipelineUpload(DcsPipeline *p, DcsPipeline::Job *job, hipStream_t stream)
{
    DcsCtx *ctx = p->ctx;
    const uint32_t n = job->nStreams;
    job->locs.resize(n); job->firstRecord.resize(n); job->streamOff.resize(n);
    size_t blobLen = 0;
    uint64_t totalRec = 0;
    for (uint32_t k = 0 ; k < n ; ++k)
    {
        const DcsStreamRef &sr = job->streams[k];
        if (sr.data == nullptr || sr.len < 3 || sr.os < DCS_OS93A || sr.os > DCS_OS95)
            return DCS_ERR_INVALID_ARG;
        const uint32_t nFrames = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
        if (nFrames == 0)
            return DCS_ERR_BAD_STREAM;
        // no stream is longer than its header plus nFrames maximal frames (the caller's buffer may be the rest of a ROM)
        const size_t most = 2 + 16 + (static_cast<size_t>(nFrames) * DCS_MAX_FRAME_BITS + 7) / 8 + 8;
        const size_t len = sr.len < most ? sr.len : most;
        blobLen = (blobLen + 3) & ~size_t(3);
        job->locs[k].off = blobLen; job->locs[k].len = static_cast<uint32_t>(len); job->locs[k].os = sr.os;
        job->locs[k].firstRecord = totalRec;
        job->streamOff[k] = blobLen;
        job->firstRecord[k] = totalRec;
        blobLen += len;
        totalRec += nFrames;
    }
    job->totalRec = totalRec;
    job->hBlobLen = blobLen;
    job->hBlobCap = ((blobLen + 3) & ~size_t(3)) + 64;          // zero tail: the walk prefetches, the packer copies whole dwords
    job->dBlobCap = job->hBlobCap;
    job->recBytes = ((p->flags & DCS_PIPE_PACK_ON_DEVICE) ? sizeof(DcsFrameDigest) : sizeof(DcsFrameIndex)) * totalRec;
    job->infoBytes = sizeof(DcsStreamInfo) * n;
    const double tu0 = nowMs();
    HIPCHK(ctx, cacheAlloc(ctx, true, reinterpret_cast<void **>(&job->hBlob), job->hBlobCap));
    const bool planOnDevice = (p->flags & DCS_PIPE_PLAN_ON_DEVICE) != 0;       // (then nothing of the index pass comes back to the host)
    if (!planOnDevice)
    {
        HIPCHK(ctx, cacheAlloc(ctx, true, &job->hRec, job->recBytes));
        HIPCHK(ctx, cacheAlloc(ctx, true, &job->hInfo, job->infoBytes));
    }
    HIPCHK(ctx, cacheAlloc(ctx, false, &job->dBlob, job->dBlobCap));
    job->dRecBytes = sizeof(DcsFrameIndex) * (totalRec ? totalRec : 1);
    job->dDigestBytes = sizeof(DcsFrameDigest) * (totalRec ? totalRec : 1);
    HIPCHK(ctx, cacheAlloc(ctx, false, &job->dRec, job->dRecBytes));
    if ((p->flags & DCS_PIPE_PACK_ON_DEVICE) && !planOnDevice)
        HIPCHK(ctx, cacheAlloc(ctx, false, &job->dDigest, job->dDigestBytes));
    HIPCHK(ctx, cacheAlloc(ctx, false, &job->dInfo, job->infoBytes));
    const double tu1 = nowMs();
    memset(job->hBlob + blobLen, 0, job->hBlobCap - blobLen);
    for (uint32_t k = 0 ; k < n ; ++k)
    {
        const DcsStreamLoc &l = job->locs[k];
        if (k + 1 < n)          // (the alignment gap in front of the next stream)
            memset(job->hBlob + l.off + l.len, 0, static_cast<size_t>(job->locs[k + 1].off - l.off) - l.len);
        memcpy(job->hBlob + l.off, job->streams[k].data, l.len);
    }
    const double tu2 = nowMs();
    HIPCHK(ctx, hipEventCreateWithFlags(&job->uploaded, hipEventDisableTiming));
    // The streams go up through EIGHT workgroups (DCS_PIPE_UP_BLOCKS overrides).  A copy kernel that reads pinned host memory with hundreds of
    // workgroups -- 2.4 MB a list: 586 of them -- keeps that many wavefronts stalled on PCIe reads whose completions travel the
    // direction the PCM's writes need: the PCM of the lists further along came down at 70-75 % of the link's rate.  With 4 to 20
    // workgroups (an upload then takes 0.15 ms instead of 0.05) the link runs at 93-97 %: 0.74 -> 0.58 ms per list sustained (round 4).
    // (a list far larger than those measured gets more of them, one per 300 KB up to 32, so that its upload stays shorter than its PCM's way down)
    static const unsigned upBlocksEnv = getenv("DCS_PIPE_UP_BLOCKS") != nullptr ? static_cast<unsigned>(std::max(1, atoi(getenv("DCS_PIPE_UP_BLOCKS")))) : 0u;
    const unsigned upBlocks = upBlocksEnv != 0 ? upBlocksEnv : static_cast<unsigned>(std::min<size_t>(32, std::max<size_t>(8, job->hBlobCap / (300u << 10))));
    HIPCHK(ctx, copyByKernel(stream, job->dBlob, job->hBlob, job->hBlobCap, (p->flags & kPipeLatency) ? 1024u : upBlocks));
    HIPCHK(ctx, hipEventRecord(job->uploaded, stream));
    if (getenv("DCS_PIPE_TRACE"))
        fprintf(stderr, "pipe upload: allocs %.2f, memcpy %.2f, hip calls %.2f\n", tu1 - tu0, tu2 - tu1, nowMs() - tu2);
    return DCS_OK;
}

// the indexer: ONE launch of the index kernel over the streams of every list that is waiting
static void pipelineIndexer(DcsPipeline *p, int which)
{
    pthread_setname_np(pthread_self(), "dcs-indexer");
    tlsBlockingWaits = (p->flags & kPipeLatency) == 0;      // (the context's own pipeline serves ONE waiting caller: its threads poll)
    DcsCtx *ctx = p->ctx;
    (void)hipSetDevice(ctx->device);
    const hipStream_t stream = p->streams[static_cast<size_t>(p->nWorkers + which)];
    // A round takes the lists that are waiting, up to about an eighth of the chip's wavefront slots (one wavefront walks one
    // stream, for milliseconds): the decode kernels of the lists further along must find room next to it, and beyond that
    // size a round's time grows with its streams anyway (2 048 streams x 256 frames 2.7 ms, 8 192 6.0 ms).
    size_t maxRoundStreams = static_cast<size_t>(ctx->numCUs) * 8;
    if (const char *e = getenv("DCS_PIPE_ROUND_STREAMS"))
        maxRoundStreams = static_cast<size_t>(std::max(1, atoi(e)));
    void *hTable = nullptr, *dTable = nullptr;      // the round's stream locations and result addresses, as uploaded
    size_t tableCap = 0;
    for (;;)
    {
        std::vector<DcsPipeline::JobPtr> jobs;
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->indexWork.wait(lk, [&] { return p->quit || !p->toIndex.empty(); });
            // (a caller that submits the parts of ONE list wants them in one round: wait a moment for the rest)
            if (p->roundGather > 1 && !p->quit)
                p->indexWork.wait_for(lk, std::chrono::microseconds(400), [&] { return p->quit || p->toIndex.size() >= p->roundGather; });
            if (p->quit && p->toIndex.empty())
            {
                if (hTable) (void)hipHostFree(hTable);
                if (dTable) (void)hipFree(dTable);
                return;
            }
            size_t roundStreams = 0;
            while (!p->toIndex.empty() && (jobs.empty() || roundStreams + p->toIndex.front()->nStreams <= maxRoundStreams))
            {
                roundStreams += p->toIndex.front()->nStreams;
                jobs.push_back(p->toIndex.front());
                p->toIndex.pop_front();
            }
        }
        if (jobs.empty())       // (the other indexer took every list while this one dropped the mutex in the gather wait)
            continue;
        const double t0 = nowMs();
        for (const DcsPipeline::JobPtr &j : jobs) j->tIndexStart = t0;
        // stream locations with ABSOLUTE device addresses (the kernel's blob base is address 0); every list's results go to
        // buffers of its own, so a round allocates nothing but its table of locations (kept from round to round)
        uint32_t nStreams = 0;
        for (const DcsPipeline::JobPtr &j : jobs) nStreams += j->nStreams;
        const size_t locBytes = sizeof(DcsStreamLoc) * nStreams, tableBytes = locBytes + sizeof(dcsidx::StreamOut) * nStreams;
        const bool packOnDevice = (p->flags & DCS_PIPE_PACK_ON_DEVICE) != 0;
        DcsStatus st = [&]() -> DcsStatus {
            if (tableCap < tableBytes)
            {
                if (hTable) (void)hipHostFree(hTable);
                if (dTable) (void)hipFree(dTable);
                hTable = nullptr; dTable = nullptr; tableCap = 0;
                const size_t want = tableBytes * 2;
                HIPCHK(ctx, hipHostMalloc(&hTable, want, hipHostMallocDefault));
                HIPCHK(ctx, hipMalloc(&dTable, want));
                tableCap = want;
            }
            DcsStreamLoc *locs = static_cast<DcsStreamLoc *>(hTable);
            dcsidx::StreamOut *outs = reinterpret_cast<dcsidx::StreamOut *>(static_cast<uint8_t *>(hTable) + locBytes);
            uint32_t k = 0;
            for (const DcsPipeline::JobPtr &j : jobs)
                for (uint32_t i = 0 ; i < j->nStreams ; ++i, ++k)
                {
                    const DcsStreamLoc &l = j->locs[i];
                    locs[k] = l;
                    locs[k].off = reinterpret_cast<uint64_t>(j->dBlob) + l.off;
                    outs[k].records = static_cast<DcsFrameIndex *>(j->dRec) + l.firstRecord;
                    outs[k].digest = j->dDigest != nullptr ? static_cast<DcsFrameDigest *>(j->dDigest) + l.firstRecord : nullptr;
                    outs[k].info = static_cast<DcsStreamInfo *>(j->dInfo) + i;
                }
            for (const DcsPipeline::JobPtr &j : jobs)
                HIPCHK(ctx, hipStreamWaitEvent(stream, j->uploaded, 0));
            HIPCHK(ctx, copyByKernel(stream, dTable, hTable, tableBytes));
            HIPCHK(ctx, launchIndexWave(stream, 0, static_cast<const DcsStreamLoc *>(dTable), nStreams, ctx->dTables, nullptr, nullptr, nullptr,
                                        reinterpret_cast<const dcsidx::StreamOut *>(static_cast<const uint8_t *>(dTable) + locBytes)));
            for (const DcsPipeline::JobPtr &j : jobs)
                if (j->hRec != nullptr)         // (planner on the device: the records stay where they are)
                {
                    HIPCHK(ctx, hipMemcpyAsync(j->hRec, packOnDevice ? j->dDigest : j->dRec, j->recBytes, hipMemcpyDeviceToHost, stream));
                    HIPCHK(ctx, hipMemcpyAsync(j->hInfo, j->dInfo, j->infoBytes, hipMemcpyDeviceToHost, stream));
                }
            HIPCHK(ctx, streamWait(ctx, stream));
            return DCS_OK;
        }();
        if (st != DCS_OK)
            (void)streamWait(ctx, stream);
        if (packOnDevice && st == DCS_OK)
            for (const DcsPipeline::JobPtr &j : jobs)
                j->dRecords = static_cast<const DcsFrameIndex *>(j->dRec);      // (they stay until the list has packed)
        const double dt = nowMs() - t0;
        pipeLog("indexer", which, "round", t0, nowMs(), jobs.size(), nStreams);
        {
            std::lock_guard<std::mutex> lk(p->m);
            for (const DcsPipeline::JobPtr &j : jobs)
            {
                j->deviceMs += dt;
                j->status = st;
                j->tIndexed = nowMs();
                p->indexed.push_back(j);
            }
        }
        p->work.notify_all();
    }
}

// Why a pipeline's batches are launched in XCD ranges (DCS_BATCH_XCD_RANGES, dcs_common.h).  A decode kernel's wavefronts may wait
// for a tail another of ITS wavefronts publishes (dcs_kernels.hip.h: hand-off).  Workgroups are dispatched in index order and a
// producer lies in a lower-numbered chunk, so within ONE kernel that has the chip to itself every wait is for a wavefront that is
// resident or through.  Two such kernels side by side break that: workgroups go round-robin to the eight XCDs, each XCD fills its
// places on its own, and an XCD can be full of kernel B's waiting consumers whose producers sit undispatched on another XCD that is
// full of kernel A's waiting consumers -- whose producers wait for a place on the first.  Nothing moves until the bound of the wait
// (500 ms) flags the frames and the lists are decoded again: seen in round 4 as two lists in a hundred of 608 011 frames (76 000
// chunks each) and once in 400 lists of 65 536; short of that, the kernels of lists in flight held each other up for most of a
// millisecond (0.94 ms per decode kernel in the pipeline against 0.05 alone).  With chain order and XCD ranges a consumer's producer
// is dispatched before it on the consumer's own XCD, so no wait depends on a place becoming free.  (ONE decode stream per device,
// which also rules the circle out, was measured first: 0.92 ms per list instead of 0.60 -- lists waiting behind each other's packers.)

// How a list's PCM comes down: by the runtime's copy (hipMemcpyAsync into pinned memory, which this runtime does with a blit
// kernel of its own), or -- DCS_PIPE_DOWN_BLOCKS=n, and always for a pipeline with ONE waiting caller (the context's own), where
// hipMemcpyAsync now and then holds the calling thread for 7 ms (profiles/NOTES.md 17) -- by dcsCopyKernel with at most n workgroups.
// Measured in round 4 (NOTES 24): with the uploads throttled (pipelineUpload) both ways down run at 93-97 % of the link.
static void pipelineDownPolicy(const DcsPipeline *p, DcsBatch *b)
{
    static const int downBlocksEnv = getenv("DCS_PIPE_DOWN_BLOCKS") != nullptr ? atoi(getenv("DCS_PIPE_DOWN_BLOCKS")) : 0;
    const bool latency = (p->flags & kPipeLatency) != 0;
    b->downByKernel = latency || downBlocksEnv > 0;
    b->downBlocks = latency ? 1024u : static_cast<unsigned>(std::max(1, downBlocksEnv));
}

// stage B: from index records (device path) or from scratch (host pool) to PCM in pinned memory
static DcsStatus pipelineDecode(DcsPipeline *p, DcsPipeline::Job *job, hipStream_t stream)
{
    const double t0 = nowMs();
    DcsStatus st = DCS_OK;
    bool fromDevice = false;
    // the batch description is needed only until the batch exists: one per worker thread, its memory kept from list to
    // list (fresh multi-megabyte vectors for every list cost more in page faults than everything else the host does)
    thread_local DcsBuiltStreams scratch;
    DcsBuiltStreams &built = scratch;
    const bool packOnDevice = (p->flags & DCS_PIPE_PACK_ON_DEVICE) != 0 && job->dRecords != nullptr;
    thread_local DcsBuiltPlan planScratch;
    if (job->hRec != nullptr)
    {
        // a stream whose frames run past its buffer reads the missing bytes as zero, which streams laid end to end
        // cannot express: such a list (truncated input) takes the host path.  What counts is the bits the frames
        // occupy, not nBytes, which includes the reference reader's look-ahead of up to three bytes (:1509).
        const DcsStreamInfo *infos = static_cast<const DcsStreamInfo *>(job->hInfo);
        fromDevice = true;
        for (uint32_t k = 0 ; k < job->nStreams && fromDevice ; ++k)
            fromDevice = infos[k].nFrames != 0
                      && 2u + static_cast<size_t>(infos[k].hdrLen) + (static_cast<size_t>(infos[k].payloadBits) + 7) / 8 <= job->locs[k].len;
        if (fromDevice && packOnDevice)
        {
            const DcsDigested in{ static_cast<const DcsFrameDigest *>(job->hRec), job->firstRecord.data(), infos, job->streamOff.data(), 0 };
            st = dcsBuildPlanFromDigest(job->streams, job->nStreams, job->extraFrames, in, planScratch);
        }
        else if (fromDevice)
        {
            const DcsPreIndexed pre{ static_cast<const DcsFrameIndex *>(job->hRec), job->firstRecord.data(), infos, job->streamOff.data() };
            st = dcsBuildStreams(job->streams, job->nStreams, job->extraFrames, built, false, false, &pre);
        }
    }
    if (st == DCS_OK && !fromDevice)
    {
        if (job->preRecords != nullptr)
        {
            const DcsPreIndexed pre{ job->preRecords, job->preFirstRecord, job->preInfos, nullptr };
            st = dcsBuildStreams(job->streams, job->nStreams, job->extraFrames, built, false, false, &pre);
        }
        else
            st = dcsBuildStreams(job->streams, job->nStreams, job->extraFrames, built, false, false);
    }
    const bool devicePacked = fromDevice && packOnDevice;
    job->onDevice = fromDevice;
    job->path = fromDevice ? (DCS_PIPE_INDEX_ON_DEVICE | (devicePacked ? DCS_PIPE_PACK_ON_DEVICE : 0u)) : 0u;
    job->firstJob = devicePacked ? planScratch.firstJob : built.firstJob;
    const size_t nJobsBuilt = devicePacked ? planScratch.jobs.size() : built.jobs.size();
    const uint8_t *blob = fromDevice ? job->hBlob : built.blob.data();
    const size_t blobLen = fromDevice ? job->hBlobLen : built.blob.size();
    double t1 = nowMs(), t2 = t1;
    if (st == DCS_OK)
    {
        const bool handoff = p->ctx->handoff;
        const DcsBuiltStreams &B = built;
        if (devicePacked)
            st = createBatchOnDevice(p->ctx, planScratch.jobs.data(), static_cast<uint32_t>(planScratch.jobs.size()), planScratch.srcs.data(),
                                     static_cast<uint32_t>(planScratch.srcs.size()), job->dRecords, static_cast<const uint8_t *>(job->dBlob),
                                     job->hBlobLen, stream, handoff, &job->batch);
        else
            st = createBatch(p->ctx, blob, blobLen, B.srcs.data(), static_cast<uint32_t>(B.srcs.size()),
                             B.jobs.data(), static_cast<uint32_t>(B.jobs.size()), nullptr, 0, stream, handoff, &job->batch);
        t2 = nowMs();
        if (st == DCS_OK) st = dcs_batch_run(job->batch, nullptr);
        // (DCS_PIPE_TRACE=3 waits for the kernels first, so that the thread log tells them from the copies)
        const double tk0 = nowMs();
        static const bool splitWait = getenv("DCS_PIPE_TRACE") != nullptr && atoi(getenv("DCS_PIPE_TRACE")) >= 3;
        if (st == DCS_OK && splitWait) st = dcs_batch_sync(job->batch);
        const double tk1 = nowMs();
        if (st == DCS_OK) pipelineDownPolicy(p, job->batch);
        if (st == DCS_OK) st = dcs_batch_download_view(job->batch, &job->pcm, &job->err);
        pipeLog("worker", 0, "kernels", tk0, tk1);
        pipeLog("worker", 0, "download", tk1, nowMs());
    }
    // A failed list may have left its pack kernel (which reads dBlob and the round's records) in flight: nothing of it
    // runs any more when those buffers go back to the cache, which knows nothing of streams.  (A list that succeeded has
    // waited for its PCM, which the worker's stream delivers after everything else.)
    if (st != DCS_OK)
    {
        if (job->batch != nullptr) { dcs_batch_destroy(job->batch); job->batch = nullptr; }
        (void)streamWait(p->ctx, stream);
    }
    pipelineFreeIndexBuffers(p, job, false);        // (the packages are on the device: the streams are no longer needed)
    if (st == DCS_OK && job->pcmDst != nullptr)
    {
        memcpy(job->pcmDst, job->pcm, sizeof(int16_t) * DCS_FRAME_SAMPLES * nJobsBuilt);
        if (job->errDst != nullptr)
            memcpy(job->errDst, job->err, sizeof(uint32_t) * nJobsBuilt);
    }
    const double t3 = nowMs();
    if (getenv("DCS_PIPE_TRACE"))
        fprintf(stderr, "pipe list: upload %.2f ms, index launch %.2f ms (records from the %s) | build %.2f create %.2f run+download %.2f\n",
                job->hostMs, job->deviceMs, fromDevice ? "device" : "host pool", t1 - t0, t2 - t1, t3 - t2);
    job->hostMs += t2 - t0;
    job->deviceMs += t3 - t2;
    (void)t1;
    return st;
}

// The device planner's stream table (DcsPlanStream, 40 bytes a stream): what the host knows of every stream of a list of
// whole streams without walking it -- where it lies in the list's blob, its layout, frame count, and the mixing parameters
// of frame 0 and of every later frame.  Also the first output frame of every stream (n + 1 entries) and the list's totals.
static DcsStatus planTableFor(const DcsStreamRef *streams, uint32_t n, uint32_t extraFrames, const DcsStreamLoc *locs, const uint64_t *firstRecord,
                              std::vector<DcsPlanStream> &table, std::vector<uint32_t> &firstJob, uint64_t *nJobsOut, uint64_t *payloadOut,
                              bool *all94Out, bool *has93aOut)
{
    table.resize(n);
    firstJob.resize(static_cast<size_t>(n) + 1);
    uint64_t nJobs = 0, payload = 0;
    bool all94 = true, has93a = false;
    for (uint32_t k = 0 ; k < n ; ++k)
    {
        const DcsStreamRef &sr = streams[k];
        const uint8_t *d = sr.data;
        const uint32_t len = locs[k].len;
        const uint32_t nFrames = (static_cast<uint32_t>(d[0]) << 8) | d[1];
        const bool typeBit = (d[2] & 0x80) != 0;
        const uint32_t h12 = (len > 3 ? d[3] : 0u) | (len > 4 ? d[4] : 0u);
        const DcsOsVersion os = static_cast<DcsOsVersion>(sr.os);
        DcsPlanStream &t = table[k];
        t = DcsPlanStream{};
        t.hdrLen = (os == DCS_OS93A && typeBit) ? 1 : 16;
        t.format = static_cast<uint8_t>(os == DCS_OS93A ? (typeBit ? DCS_FMT_93A_T1 : DCS_FMT_93_T0)
                                      : os == DCS_OS93B ? (typeBit ? DCS_FMT_93B_T1 : DCS_FMT_93_T0)
                                      : !typeBit ? DCS_FMT_94_T0 : (h12 & 0x80) == 0 ? DCS_FMT_94_T1_S0 : DCS_FMT_94_T1_S3);
        t.xform = (os == DCS_OS93A || os == DCS_OS93B) ? DCS_XFORM_93 : DCS_XFORM_94;
        all94 = all94 && t.xform == DCS_XFORM_94;
        has93a = has93a || t.format == DCS_FMT_93A_T1;
        t.streamOff = locs[k].off;
        t.len = len;
        t.firstRecord = static_cast<uint32_t>(firstRecord[k]);
        t.firstJob = static_cast<uint32_t>(nJobs);
        t.nFrames = nFrames;
        uint16_t mm[2]; uint8_t vs[2];
        const DcsStatus st = dcs_stream_params_from(os, sr.volume, sr.level, sr.channelVolume, 0x7FFF, 2, mm, vs);    // frame 0, and every later frame
        if (st != DCS_OK)
            return st;
        t.mixMul0 = mm[0]; t.mixMulN = mm[1]; t.volShift0 = vs[0]; t.volShiftN = vs[1];
        firstJob[k] = static_cast<uint32_t>(nJobs);
        nJobs += nFrames + extraFrames;
        payload += len;
    }
    firstJob[n] = static_cast<uint32_t>(nJobs);
    *nJobsOut = nJobs; *payloadOut = payload; *all94Out = all94; *has93aOut = has93a;
    return DCS_OK;
}

// Planner on the device (DCS_PIPE_PLAN_ON_DEVICE), stage B: the list's index records are on the device (an indexer's round
// put them there); planner, packer and decode kernels and the PCM's way down are queued on the worker's stream, and the one
// wait is for the PCM.  Returns DCS_OK with *served = false when the arithmetic plan cannot serve the list (DCS_PLAN_*):
// the caller then takes the host-planned path.
static DcsStatus pipelineDecodePlanned(DcsPipeline *p, DcsPipeline::Job *job, hipStream_t stream, bool *served)
{
    DcsCtx *ctx = p->ctx;
    *served = false;
    const double t0 = nowMs();
    DcsStatus st = DCS_OK;
    const uint32_t n = job->nStreams;
    // what the host knows of every stream without walking it
    thread_local std::vector<DcsPlanStream> table;
    uint64_t nJobs = 0, payload = 0;
    bool all94 = true, has93a = false;
    st = planTableFor(job->streams, n, job->extraFrames, job->locs.data(), job->firstRecord.data(), table, job->firstJob, &nJobs, &payload, &all94, &has93a);
    if (st != DCS_OK)
        return st;
    if (nJobs > 0xFFFFFFFFull || job->totalRec > 0xFFFFFFFFull)
        return DCS_ERR_CAPACITY;
    // A chunk whose frames' compressed bytes overflow the kernel's bit pool (224 bytes per slot) is what the arithmetic plan cannot
    // close early as the host planner does: the list is planned again with fewer frames per chunk -- three quarters, then half of the
    // slots -- before the host path gets it (one stream of large frames among 600 would otherwise cost the whole list the device).
    double t1 = t0, t2 = t0;
    uint32_t flag = 0;
    const int fullFpw = chooseFpw(ctx, static_cast<uint32_t>(nJobs), all94);
    const int tries[3] = { 0, fullFpw * 3 / 4, fullFpw / 2 };
    for (int attempt = 0 ; attempt < 3 ; ++attempt)
    {
        if (job->batch != nullptr) { dcs_batch_destroy(job->batch); job->batch = nullptr; }
        st = createBatchPlannedOnDevice(ctx, table.data(), n, job->extraFrames, static_cast<uint32_t>(nJobs), static_cast<uint32_t>(job->totalRec), all94,
                                        has93a, payload, static_cast<const DcsFrameIndex *>(job->dRec), static_cast<const DcsStreamInfo *>(job->dInfo),
                                        static_cast<const uint8_t *>(job->dBlob), job->hBlobLen, stream, &job->batch, tries[attempt]);
        t1 = nowMs();
        pipeLog("worker", 0, "plan-queue", t0, t1);
        if (st == DCS_OK) st = dcs_batch_run(job->batch, nullptr);
        if (st == DCS_OK) st = batchQueuePlanFlag(job->batch);
        t2 = nowMs();
        pipeLog("worker", 0, "run-queue", t1, t2);
        if (st == DCS_OK) pipelineDownPolicy(p, job->batch);
        if (st == DCS_OK) st = dcs_batch_download_view(job->batch, &job->pcm, &job->err);
        pipeLog("worker", 0, "download", t2, nowMs());
        flag = 0;
        if (st == DCS_OK)
            flag = batchPlanFlag(job->batch);
        // again with fewer slots only for an overflow, and not when the list is (also) truncated: that one is the host's whatever the plan
        if (st != DCS_OK || (flag & DCS_PLAN_POOL_OVERFLOW) == 0 || (flag & DCS_PLAN_TRUNCATED) != 0)
            break;
    }
    job->hostMs += t1 - t0;
    job->deviceMs += nowMs() - t1;
    if (st != DCS_OK || flag != 0)
    {
        // not served (or failed): nothing of this attempt stays
        if (job->batch != nullptr) { dcs_batch_destroy(job->batch); job->batch = nullptr; }
        (void)streamWait(ctx, stream);
        pipelineFreeIndexBuffers(p, job, false);
        return st;
    }
    job->onDevice = true;
    job->path = DCS_PIPE_INDEX_ON_DEVICE | DCS_PIPE_PACK_ON_DEVICE | DCS_PIPE_PLAN_ON_DEVICE;
    pipelineFreeIndexBuffers(p, job, false);
    if (job->pcmDst != nullptr)
    {
        memcpy(job->pcmDst, job->pcm, sizeof(int16_t) * DCS_FRAME_SAMPLES * nJobs);
        if (job->errDst != nullptr)
            memcpy(job->errDst, job->err, sizeof(uint32_t) * nJobs);
    }
    *served = true;
    return DCS_OK;
}

static void pipelineWorker(DcsPipeline *p, int id)
{
    pthread_setname_np(pthread_self(), "dcs-worker");
    static const bool xcdRanges = getenv("DCS_PIPE_XCD_RANGES") == nullptr || atoi(getenv("DCS_PIPE_XCD_RANGES")) != 0;     // (0: an experiment switch)
    tlsXcdRanges = xcdRanges;                               // (this thread's batches: chain order, launched in XCD ranges; see above)
    tlsBlockingWaits = (p->flags & kPipeLatency) == 0;      // (the context's own pipeline serves ONE waiting caller: its threads poll)
    (void)hipSetDevice(p->ctx->device);
    const hipStream_t stream = p->streams[id];
    const bool deviceIndex = (p->flags & DCS_PIPE_INDEX_ON_DEVICE) != 0;
    for (;;)
    {
        DcsPipeline::JobPtr job;
        bool stageB = false;
        {
            // With the index pass on the device the first `nUploaders` workers take ONLY fresh lists (stage A: lay the streams
            // out, send them up) and the others only indexed ones (stage B, which ends in a wait of milliseconds for the PCM).
            // When every worker took whatever was there, indexed lists first, the pipeline fell into lock step: all lists in
            // flight reached stage B together, the fresh ones behind them waited 20 ms for a worker, the indexers ran dry and
            // then walked everything in a burst (round 4, DCS_PIPE_TRACE=2: rounds of 8 lists back to back, then 20 ms of nothing).
            const bool takesFresh = !deviceIndex || id < p->nUploaders, takesIndexed = !deviceIndex || id >= p->nUploaders;
            std::unique_lock<std::mutex> lk(p->m);
            p->work.wait(lk, [&] { return p->quit || (takesIndexed && !p->indexed.empty()) || (takesFresh && !p->fresh.empty()); });
            if (takesIndexed && !p->indexed.empty())            // lists that are further along come first
            {
                job = p->indexed.front(); p->indexed.pop_front();
                stageB = true;
            }
            else if (takesFresh && !p->fresh.empty())
            {
                job = p->fresh.front(); p->fresh.pop_front();
            }
            else
                return;                         // quit
        }
        if (!stageB && deviceIndex)
        {
            const double t0 = nowMs();
            job->tTaken = t0;
            const DcsStatus st = pipelineUpload(p, job.get(), stream);
            job->hostMs += nowMs() - t0;
            pipeLog("worker", id, "upload", t0, nowMs());
            if (st != DCS_OK)
            {
                pipelineFreeIndexBuffers(p, job.get(), false);
                pipelineFinish(p, job, st);
                continue;
            }
            if (job->preRecords != nullptr && (p->flags & DCS_PIPE_PLAN_ON_DEVICE))
            {
                // The host pool has walked this list already (dcsDecodeStreamsInParts, the walk shared with the device): its
                // records and stream summaries -- the very bytes the index kernel would have written -- go up behind the streams
                // and the list joins the indexed ones.  (Pinned memory of the caller's; this worker's stream is waited for, as an
                // indexer waits for its round, because stage B runs on another stream.)
                DcsStatus s2 = [&]() -> DcsStatus {
                    DcsCtx *ctx = p->ctx;
                    HIPCHK(ctx, copyByKernel(stream, job->dRec, job->preRecords + job->preFirstRecord[0], sizeof(DcsFrameIndex) * job->totalRec, 64u));
                    HIPCHK(ctx, copyByKernel(stream, job->dInfo, job->preInfos, job->infoBytes, 1u));
                    HIPCHK(ctx, streamWait(ctx, stream));
                    return DCS_OK;
                }();
                const double t1 = nowMs();
                {
                    std::lock_guard<std::mutex> lk(p->m);
                    job->dRecords = static_cast<const DcsFrameIndex *>(job->dRec);
                    job->status = s2;
                    job->tQueuedForIndex = job->tIndexStart = job->tIndexed = t1;
                    p->indexed.push_back(job);
                }
                p->work.notify_all();
                continue;
            }
            {
                std::lock_guard<std::mutex> lk(p->m);
                job->tQueuedForIndex = nowMs();
                p->toIndex.push_back(job);
            }
            p->indexWork.notify_all();
            continue;
        }
        DcsStatus st = job->status;             // (the indexer's)
        job->tStageB = nowMs();
        if (st == DCS_OK && (p->flags & DCS_PIPE_PLAN_ON_DEVICE))
        {
            bool served = false;
            st = pipelineDecodePlanned(p, job.get(), stream, &served);
            if (st == DCS_OK && !served)
            {
                pipeLog("worker", id, "not-served", job->tStageB, nowMs());
                st = pipelineDecode(p, job.get(), stream);       // (host index pass, host planner: serves every list)
            }
        }
        else if (st == DCS_OK)
            st = pipelineDecode(p, job.get(), stream);
        pipeLog("worker", id, "stageB", job->tStageB, nowMs());
        job->tDone = nowMs();
        if (getenv("DCS_PIPE_TRACE") && job->tIndexed != 0)
        {
            fprintf(stderr, "pipe life: submit->taken %.2f, upload %.2f, wait for indexer %.2f, index round %.2f, wait for worker %.2f, stage B %.2f\n",
                    job->tTaken - job->tSubmit, job->tQueuedForIndex - job->tTaken, job->tIndexStart - job->tQueuedForIndex,
                    job->tIndexed - job->tIndexStart, job->tStageB - job->tIndexed, job->tDone - job->tStageB);
        }
        pipelineFreeIndexBuffers(p, job.get(), false);      // (a list whose indexer failed still holds them)
        pipelineFinish(p, job, st);
    }
}

static DcsStatus pipelineCreate(DcsCtx *ctx, int depth, uint32_t flags, DcsPipeline **out);

extern "C" DcsStatus dcs_pipeline_create(DcsCtx *ctx, int depth, uint32_t flags, DcsPipeline **out)
{
    if ((flags & ~(DCS_PIPE_INDEX_ON_DEVICE | DCS_PIPE_PACK_ON_DEVICE | DCS_PIPE_PLAN_ON_DEVICE)) != 0)
        return DCS_ERR_INVALID_ARG;
    return pipelineCreate(ctx, depth, flags, out);
}

// (flags may carry kPipeLatency, which the public entry does not accept)
static DcsStatus pipelineCreate(DcsCtx *ctx, int depth, uint32_t flags, DcsPipeline **out)
{
    if (ctx == nullptr || out == nullptr || depth < 1 || depth > 64)
        return DCS_ERR_INVALID_ARG;
    if (flags & DCS_PIPE_PLAN_ON_DEVICE)
        flags |= DCS_PIPE_PACK_ON_DEVICE;
    if (flags & DCS_PIPE_PACK_ON_DEVICE)
        flags |= DCS_PIPE_INDEX_ON_DEVICE;           // (the packer works from the records the index pass leaves on the device)
    *out = nullptr;
    DcsPipeline *p = new (std::nothrow) DcsPipeline;
    if (p == nullptr)
        return DCS_ERR_NO_MEMORY;
    p->ctx = ctx;
    p->depth = depth;
    p->flags = flags;
    if (hipSetDevice(ctx->device) != hipSuccess)
    {
        delete p;
        setError(ctx, "dcs_pipeline_create: hipSetDevice failed");
        return DCS_ERR_HIP;
    }
    // with the index pass on the device a worker holds a list only while it works on it, so there need not be one per
    // list in flight: as many as the host has cores, and a few more for the ones that wait for a copy.  With the packer on
    // the device as well a list costs a worker under a millisecond of its own work, and what more workers add is contention
    // inside the HIP runtime: measured with 32 lists in flight on 16 CPUs, 4 to 6 workers 1.65-1.95 ms per list at 7-9 CPU-ms,
    // 10 workers 1.8-2.4, 20 workers 2.1-2.5 at 18-22 CPU-ms (tools/pipe_trace.py, round 2).  Planner on the device too: a
    // worker spends a third of a millisecond on a list and then sleeps until its PCM is down; 0.70-0.77 ms per list with 6,
    // 8, 12, 16 or 24 of them (round 3: the link is what bounds it).  Round 5 (tools/thread_cpu.py, three interleaved rounds of 1 500
    // lists): TWO of them 0.59-0.63 ms per list at 0.43-0.55 CPU-ms, three 0.65 / 0.67, eight 0.63-0.66 / 0.74-0.87 -- every worker has a
    // HIP stream of its own, the runtime spreads a process's streams over GPU_MAX_HW_QUEUES (8) hardware queues per device, and once
    // streams share queues a thread of the RUNTIME orders them on the host: two pipelines with twelve streams each on one device (two
    // contexts of a node, or two ranks, sharing a card) kept that thread 65 % busy, 0.45 CPU-ms per list; with two stage-B workers a
    // pipeline has six streams.  (The context's own pipeline, which serves one waiting caller's parts side by side, keeps eight.)
    int nWorkers = (flags & DCS_PIPE_PLAN_ON_DEVICE)  ? std::min(depth, (flags & kPipeLatency) ? 8 : 2)
                 : (flags & DCS_PIPE_PACK_ON_DEVICE)  ? std::min(depth, std::max(4, dcs_host_threads() / 3))
                 : (flags & DCS_PIPE_INDEX_ON_DEVICE) ? std::min(depth, dcs_host_threads() + 4) : depth;
    if (const char *w = getenv("DCS_PIPE_WORKERS"))
        nWorkers = std::max(1, std::min(64, atoi(w)));
    int nUploaders = 0;
    if (flags & DCS_PIPE_INDEX_ON_DEVICE)
    {
        // stage A costs a worker a quarter to half a millisecond per list: two of them keep up with any rate the link allows
        nUploaders = (flags & kPipeLatency) ? std::min(depth, 4) : depth >= 4 ? 2 : 1;      // (one waiting caller: its parts go up side by side)
        if (const char *u = getenv("DCS_PIPE_UPLOADERS"))
            nUploaders = std::max(1, std::min(8, atoi(u)));
        nWorkers += nUploaders;
    }
    p->nUploaders = nUploaders;
    int prioLeast = 0, prioGreatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prioLeast, &prioGreatest);
    p->nWorkers = nWorkers;
    // two indexers, so that one round's walk runs while the other round's lists gather and its records come back
    const int nIndexers = (flags & DCS_PIPE_INDEX_ON_DEVICE) ? 2 : 0;
    for (int i = 0 ; i < nWorkers + nIndexers ; ++i)
    {
        hipStream_t s = nullptr;
        // The indexers' streams get the lowest and the highest priority: not for the priority, but because streams of
        // different priorities never share a hardware queue.  Their launches run for milliseconds (a walk is serial per
        // stream, on a handful of lanes), and neither a worker's copies and 40-microsecond kernels nor the other
        // indexer's launch must queue up behind one.
        // (DCS_PIPE_INDEXER_PRIO=ab, a / b one of l(east) g(reatest) n(ormal): the two indexers' priorities, an experiment switch)
        static const char *prioEnv = getenv("DCS_PIPE_INDEXER_PRIO");
        auto prioOf = [&](char c, int dflt) { return c == 'l' ? prioLeast : c == 'g' ? prioGreatest : c == 'n' ? (prioLeast + prioGreatest) / 2 : dflt; };
        const int idxPrio = i == nWorkers ? prioOf(prioEnv && prioEnv[0] ? prioEnv[0] : 0, prioLeast)
                                          : prioOf(prioEnv && prioEnv[0] && prioEnv[1] ? prioEnv[1] : 0, prioGreatest);
        const hipError_t e = i >= nWorkers ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, idxPrio)
                                           : hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess)
        {
            for (hipStream_t t : p->streams) (void)hipStreamDestroy(t);
            delete p;
            setError(ctx, "dcs_pipeline_create: hipStreamCreate failed");
            return DCS_ERR_HIP;
        }
        p->streams.push_back(s);
    }
    for (int i = 0 ; i < nWorkers ; ++i)
        p->workers.emplace_back(pipelineWorker, p, i);
    for (int i = 0 ; i < nIndexers ; ++i)
        p->indexers.emplace_back(pipelineIndexer, p, i);
    *out = p;
    return DCS_OK;
}

extern "C" void dcs_pipeline_destroy(DcsPipeline *p)
{
    printWaitStats("dcs_pipeline_destroy");
    if (p == nullptr)
        return;
    {
        std::unique_lock<std::mutex> lk(p->m);
        // lists still in flight are finished first (their streams belong to the caller)
        p->finished.wait(lk, [&] { for (const DcsPipeline::JobPtr &j : p->order) if (!j->done) return false; return true; });
        p->quit = true;
    }
    p->work.notify_all();
    p->indexWork.notify_all();
    for (std::thread &w : p->workers)
        w.join();
    for (std::thread &t : p->indexers)
        t.join();
    (void)hipSetDevice(p->ctx->device);
    pipelineRelease(p, p->held);
    for (DcsPipeline::JobPtr &j : p->order)
        pipelineRelease(p, j);
    for (hipStream_t s : p->streams)
        (void)hipStreamDestroy(s);
    delete p;
}

static DcsStatus pipelineSubmit(DcsPipeline *p, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames,
                                int16_t *pcmDst, uint32_t *errDst, const DcsFrameIndex *preRecords = nullptr,
                                const uint64_t *preFirstRecord = nullptr, const DcsStreamInfo *preInfos = nullptr)
{
    if (p == nullptr || streams == nullptr || nStreams == 0)
        return DCS_ERR_INVALID_ARG;
    DcsPipeline::JobPtr job = std::make_shared<DcsPipeline::Job>();
    job->streams = streams; job->nStreams = nStreams; job->extraFrames = extraFrames;
    job->pcmDst = pcmDst; job->errDst = errDst;
    job->preRecords = preRecords; job->preFirstRecord = preFirstRecord; job->preInfos = preInfos;
    const double ts0 = nowMs();
    {
        std::unique_lock<std::mutex> lk(p->m);
        // at most `depth` lists between submit and collect (each holds device and pinned buffers)
        p->room.wait(lk, [&] { return static_cast<int>(p->order.size()) < p->depth; });
        job->tSubmit = nowMs();
        pipeLog("caller", 0, "submit-wait", ts0, job->tSubmit);
        p->fresh.push_back(job);
        p->order.push_back(job);
    }
    p->work.notify_all();           // (workers have roles: the one woken must be one that takes fresh lists)
    return DCS_OK;
}

extern "C" DcsStatus dcs_pipeline_submit(DcsPipeline *p, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames)
{
    return pipelineSubmit(p, streams, nStreams, extraFrames, nullptr, nullptr);
}

extern "C" DcsStatus dcs_pipeline_collect(DcsPipeline *p, DcsPipelineResult *out)
{
    if (p == nullptr || out == nullptr)
        return DCS_ERR_INVALID_ARG;
    const double tc0 = nowMs();
    (void)hipSetDevice(p->ctx->device);
    const double tc1 = nowMs();
    pipelineRelease(p, p->held);                    // the previous result's buffers go back to the context's cache
    pipeLog("caller", 0, "set-device", tc0, tc1);
    pipeLog("caller", 0, "release", tc1, nowMs());
    DcsPipeline::JobPtr job;
    {
        std::unique_lock<std::mutex> lk(p->m);
        if (p->order.empty())
            return DCS_ERR_INVALID_ARG;             // nothing submitted
        job = p->order.front();
        p->finished.wait(lk, [&] { return job->done; });
        p->order.pop_front();
    }
    p->room.notify_one();
    p->held = job;
    memset(out, 0, sizeof(*out));
    out->status = job->status;
    out->nStreams = job->nStreams;
    out->hostMs = static_cast<float>(job->hostMs);
    out->path = job->path;
    out->deviceMs = static_cast<float>(job->deviceMs);
    if (job->status == DCS_OK)
    {
        out->pcm = job->pcm;
        out->err = job->err;
        out->frameOffsets = job->firstJob.data();
        out->nFrames = job->firstJob.empty() ? 0u : job->firstJob.back();
    }
    return job->status;
}

// dcs_decode_streams for a LARGE list: cut into parts (contiguous stream ranges balanced by frames) that go through a
// pipeline the context keeps for the purpose (index pass on the host pool), so that the planner, packer and upload of
// one part run while another part is indexed and a third decodes and comes back; every worker copies its part's PCM
// straight into the caller's buffer.  One synchronous call for the caller, the same PCM -- 15 ms become 5 for 65 536
// frames.  *handled = false: the list is small, the caller takes the direct path.
DcsStatus dcsDecodeStreamsInParts(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames,
                                  int16_t *pcmOut, size_t pcmCapFrames, uint32_t *frameOffsets, uint32_t *errOut, bool *handled)
{
    *handled = false;
    constexpr uint32_t kParts = 8;
    if (nStreams < 4 * kParts)
        return DCS_OK;
    std::vector<uint32_t> frames(nStreams);
    std::vector<uint64_t> first(static_cast<size_t>(nStreams) + 1, 0);
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        if (streams[k].data == nullptr || streams[k].len < 3)
            return DCS_OK;                          // (the direct path reports it)
        const uint32_t nf = (static_cast<uint32_t>(streams[k].data[0]) << 8) | streams[k].data[1];
        if (nf == 0)
            return DCS_OK;
        frames[k] = nf + extraFrames;
        first[k + 1] = first[k] + frames[k];
    }
    if (first[nStreams] < 32768 || first[nStreams] > pcmCapFrames || first[nStreams] > 0xFFFFFFFFull)
        return DCS_OK;
    const bool onDevice = ctx->largeListOnDevice;
    if (ctx->internalPipe == nullptr)
    {
        const DcsStatus st = pipelineCreate(ctx, static_cast<int>(kParts), (onDevice ? DCS_PIPE_ALL_ON_DEVICE : 0u) | kPipeLatency, &ctx->internalPipe);
        if (st != DCS_OK)
            return DCS_OK;                          // (no pipeline: the direct path still works)
        if (onDevice)
            ctx->internalPipe->roundGather = kParts;
    }
    *handled = true;
    // The parts taper: the call ends one part's latency (build, create, upload, kernel, download) after the index pass
    // has reached the list's last stream, so the last parts are small -- 4 4 4 4 3 2 2 1 twenty-fourths of the frames
    // (equal parts: 4.6 ms for 65 536 frames, of which 1.45 behind the index pass).
    // (the walk shared with the device: the device walks the LAST part(s), and what it does behind its walk -- planner, packer,
    // decode, the PCM's way down -- is the call's tail: those parts are the small ones)
    static const uint32_t kWeightHost[kParts] = { 4, 4, 4, 4, 3, 2, 2, 1 }, kWeightDevice[kParts] = { 3, 3, 3, 3, 3, 3, 3, 3 },
                          kWeightShared[kParts] = { 4, 4, 4, 3, 3, 3, 2, 1 };
    const uint32_t *kWeight = !onDevice ? kWeightHost : ctx->largeListShared ? kWeightShared : kWeightDevice;
    uint32_t cut[kParts + 1];
    {
        uint32_t wSum = 0, wAcc = 0;
        for (uint32_t r = 0 ; r < kParts ; ++r)
            wSum += kWeight[r];
        cut[0] = 0;
        for (uint32_t r = 1 ; r < kParts ; ++r)
        {
            wAcc += kWeight[r - 1];
            const uint64_t target = first[nStreams] * wAcc / wSum;
            uint32_t k = static_cast<uint32_t>(std::lower_bound(first.begin(), first.end(), target) - first.begin());
            // every part keeps at least one stream (nStreams >= 4 * kParts)
            k = std::max(k, cut[r - 1] + 1);
            k = std::min(k, nStreams - (kParts - r));
            cut[r] = k;
        }
        cut[kParts] = nStreams;
    }
    DcsStatus st = DCS_OK;
    // the index pass over the WHOLE list in one region of the host pool (eight regions of 32 streams each would balance
    // badly over the pool's threads), then the parts go to the workers with their records
    // (only the host-index path needs them: 148 B per frame, kept per calling thread; the device path leaves them empty)
    thread_local std::vector<DcsFrameIndex> records;
    std::vector<DcsStreamInfo> infos;
    std::vector<uint64_t> firstRecord;
    if (!onDevice)
    {
        infos.resize(nStreams);
        firstRecord.resize(nStreams);
        uint64_t nRec = 0;
        for (uint32_t k = 0 ; k < nStreams ; ++k)
        {
            firstRecord[k] = nRec;
            nRec += frames[k] - extraFrames;
        }
        if (records.size() < nRec)
            records.resize(nRec);
    }
    const double tI0 = nowMs();
    // a part goes to the workers the moment the last of its streams has been indexed (by whichever pool thread that was):
    // planner, packer, upload and decode of the early parts run under the index pass of the late ones
    std::atomic<uint32_t> left[kParts];
    std::atomic<uint32_t> submitted{ 0 };
    std::atomic<int> submitError{ DCS_OK };
    std::vector<uint8_t> partOf(nStreams);
    for (uint32_t r = 0 ; r < kParts ; ++r)
    {
        left[r].store(cut[r + 1] - cut[r]);
        for (uint32_t k = cut[r] ; k < cut[r + 1] ; ++k)
            partOf[k] = static_cast<uint8_t>(r);
    }
    DcsFrameIndex *const recs = records.data();     // (`records` is thread-local: the pool threads must not name it)
    const std::function<void(uint32_t)> done = [&, recs](uint32_t k) {
        const uint32_t r = partOf[k];
        if (left[r].fetch_sub(1) != 1)
            return;
        const uint64_t f0 = first[cut[r]];
        const DcsStatus s1 = pipelineSubmit(ctx->internalPipe, streams + cut[r], cut[r + 1] - cut[r], extraFrames,
                                            pcmOut + f0 * DCS_FRAME_SAMPLES, errOut ? errOut + f0 : nullptr,
                                            recs, firstRecord.data() + cut[r], infos.data() + cut[r]);
        if (s1 == DCS_OK)
            submitted.fetch_add(1);
        else
        {
            int expected = DCS_OK;
            submitError.compare_exchange_strong(expected, s1);
        }
    };
    const int idxThreads = 0;        // (all of the pool: leaving a quarter of the CPUs to the workers was measured, 6.8 against 5.8 ms)
    void *sharedPinned = nullptr;
    size_t sharedPinnedBytes = 0;
    uint32_t hostParts = 0;
    double tCall0 = nowMs();
    if (onDevice)
    {
        // The walk SHARED between host and device (dcs_ctx_set_large_list_path 2, the default; round 5).  The index kernel takes as
        // long as its longest stream whatever the number of streams (a lone wavefront per stream: 1.9 ms for 256 frames), and until
        // it is through nothing of the list can be decoded; the host pool walks a stream fourteen times faster and delivers the
        // list's FIRST parts while the device walks the last ones: their planner, packer, decode and -- what counts -- their PCM's
        // way down run under the index kernel.  The host's records are the index kernel's, byte for byte (tests/test_gpu_parity.py),
        // so they go up and take its place (pipelineWorker).  How many parts the host takes follows what was measured: the side
        // that finished later in the last call gets less in the next.
        hostParts = ctx->largeListShared ? static_cast<uint32_t>(std::max(0, std::min(static_cast<int>(kParts) - 1, ctx->sharedHostParts))) : 0u;
        //  The pool runs ONE parallel region at a time: with several contexts decoding large lists at once (dcs_node,
        // dcs_decode_streams_sharded: a thread per device) the shared walks would queue up one behind the other, the last device
        // waiting (N - 1) host walks before its first part is even submitted.  A context that finds the pool taken walks its whole
        // list on its own device (mode 1's 3.3 ms, whatever the others do).  A share that has fallen to nothing is tried again
        // with one part every sixteenth call.
        if (hostParts != 0 && dcsIndexPoolBusy())
            hostParts = 0;
        if (ctx->largeListShared && ctx->sharedHostParts == 0 && (++ctx->sharedProbe & 15) == 0 && !dcsIndexPoolBusy())
            hostParts = 1;
        const uint32_t hostStreams = cut[hostParts];
        const uint64_t hostRecs = first[hostStreams] - static_cast<uint64_t>(hostStreams) * extraFrames;
        DcsFrameIndex *hRecs = nullptr;
        DcsStreamInfo *hInfos = nullptr;
        if (hostParts != 0)
        {
            sharedPinnedBytes = sizeof(DcsFrameIndex) * hostRecs + sizeof(DcsStreamInfo) * hostStreams + 64;
            if (cacheAlloc(ctx, true, &sharedPinned, sharedPinnedBytes) != hipSuccess)
            {
                (void)hipGetLastError();
                sharedPinned = nullptr;
                hostParts = 0;
            }
            else
            {
                hRecs = static_cast<DcsFrameIndex *>(sharedPinned);
                hInfos = reinterpret_cast<DcsStreamInfo *>(static_cast<uint8_t *>(sharedPinned) + ((sizeof(DcsFrameIndex) * hostRecs + 15) & ~size_t(15)));
                firstRecord.resize(hostStreams);
                uint64_t nRec = 0;
                for (uint32_t k = 0 ; k < hostStreams ; ++k)
                {
                    firstRecord[k] = nRec;
                    nRec += frames[k] - extraFrames;
                }
            }
        }
        {
            std::lock_guard<std::mutex> lk(ctx->internalPipe->m);
            ctx->internalPipe->roundGather = kParts - hostParts;           // (the index round waits until its parts are all there)
            ctx->internalPipe->lastHostWalkedDone = ctx->internalPipe->lastDeviceWalkedDone = 0;
        }
        tCall0 = nowMs();
        // the device's parts first: index walk, planner, packer and decode on the device, the workers copy the PCM out
        for (uint32_t r = hostParts ; r < kParts && st == DCS_OK ; ++r)
        {
            const uint64_t f0 = first[cut[r]];
            st = pipelineSubmit(ctx->internalPipe, streams + cut[r], cut[r + 1] - cut[r], extraFrames,
                                pcmOut + f0 * DCS_FRAME_SAMPLES, errOut ? errOut + f0 : nullptr);
            if (st == DCS_OK)
                submitted.fetch_add(1);
        }
        if (hostParts != 0 && st == DCS_OK)
        {
            // ... and the host's: a part goes to the pipeline the moment the pool has walked the last of its streams
            const std::function<void(uint32_t)> walked = [&, hRecs, hInfos](uint32_t k) {
                const uint32_t r = partOf[k];
                if (left[r].fetch_sub(1) != 1)
                    return;
                const uint64_t f0 = first[cut[r]];
                const DcsStatus s1 = pipelineSubmit(ctx->internalPipe, streams + cut[r], cut[r + 1] - cut[r], extraFrames,
                                                    pcmOut + f0 * DCS_FRAME_SAMPLES, errOut ? errOut + f0 : nullptr,
                                                    hRecs, firstRecord.data() + cut[r], hInfos + cut[r]);
                if (s1 == DCS_OK)
                    submitted.fetch_add(1);
                else
                {
                    int expected = DCS_OK;
                    submitError.compare_exchange_strong(expected, s1);
                }
            };
            st = dcsIndexStreamsNotify(streams, hostStreams, idxThreads, hRecs, firstRecord.data(), hInfos, &walked);
            if (st == DCS_OK)
                st = static_cast<DcsStatus>(submitError.load());
        }
    }
    else
    {
        st = dcsIndexStreamsNotify(streams, nStreams, idxThreads, recs, firstRecord.data(), infos.data(), &done);
        if (st == DCS_OK)
            st = static_cast<DcsStatus>(submitError.load());
    }
    const double tI1 = nowMs();
    for (uint32_t r = 0, n = submitted.load() ; r < n ; ++r)
    {
        DcsPipelineResult res;
        const DcsStatus s1 = dcs_pipeline_collect(ctx->internalPipe, &res);
        if (st == DCS_OK)
            st = s1;
    }
    if (onDevice && hostParts != 0)
    {
        // who finished later?  (the pool's parts are submitted as they are walked, so both times are of lists of this call)
        double tHost, tDev;
        {
            std::lock_guard<std::mutex> lk(ctx->internalPipe->m);
            tHost = ctx->internalPipe->lastHostWalkedDone - tCall0;
            tDev = ctx->internalPipe->lastDeviceWalkedDone - tCall0;
        }
        if (st == DCS_OK && tHost > 0 && tDev > 0)
        {
            if (tHost > tDev + 0.15 && ctx->sharedHostParts > 0)
                ctx->sharedHostParts -= 1;
            else if (tDev > tHost + 0.15 && ctx->sharedHostParts < static_cast<int>(kParts) - 1)
                ctx->sharedHostParts += 1;
        }
        if (getenv("DCS_PIPE_TRACE"))
            fprintf(stderr, "decode in parts: %u of %u parts walked by the host pool, done at %.2f ms; the device's at %.2f ms; next call %d\n",
                    hostParts, kParts, tHost, tDev, ctx->sharedHostParts);
    }
    if (sharedPinned != nullptr)
        cacheFree(ctx, true, sharedPinned, sharedPinnedBytes);
    if (getenv("DCS_PIPE_TRACE"))
        fprintf(stderr, "decode in parts: index %.2f ms, parts %.2f ms\n", tI1 - tI0, nowMs() - tI1);
    if (frameOffsets != nullptr)
        for (uint32_t k = 0 ; k <= nStreams ; ++k)
            frameOffsets[k] = static_cast<uint32_t>(first[k]);
    return st;
}

// dcs_pipeline.hip.h -- batches in flight: the host preparation of batch k+1 (index pass, mixing parameters, chunk
// planner, packer) runs while the GPU decodes batch k and batch k-1 comes back into pinned memory.  Included at the end
// of dcs_runtime.hip (it uses the runtime's batch internals).
//
// The reference decodes its batch job (`--extract-streams`, DCSExplorer.cpp:1742-1907) one stream after the other on
// one thread; here a caller submits lists of whole streams and collects their PCM in submission order.  `depth` worker
// threads each take a submitted job through all its stages on a HIP stream of their own, so the stages of different
// jobs overlap by themselves: a worker that waits for its kernel or its copy leaves the host cores to the others.
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>

struct DcsPipeline
{
    struct Job
    {
        const DcsStreamRef *streams = nullptr;
        uint32_t nStreams = 0, extraFrames = 0;
        DcsBuiltStreams built;
        DcsBatch *batch = nullptr;
        const int16_t *pcm = nullptr;
        const uint32_t *err = nullptr;
        DcsStatus status = DCS_OK;
        bool done = false;
        double hostMs = 0, deviceMs = 0;        // preparation / upload + kernel + download, as the worker saw them
        // index pass on the device: the streams as uploaded for it (pinned), kept for the packer
        uint8_t *hBlob = nullptr;
        size_t hBlobCap = 0, hBlobLen = 0;
    };
    DcsCtx *ctx = nullptr;
    int depth = 0;
    uint32_t flags = 0;                             // DCS_PIPE_*
    std::mutex m;
    std::condition_variable work, finished, room;
    std::deque<std::shared_ptr<Job>> queue;         // submitted, not yet taken by a worker
    std::deque<std::shared_ptr<Job>> order;         // submitted, not yet collected (submission order)
    std::shared_ptr<Job> held;                      // the job whose result the caller is reading
    std::vector<std::thread> workers;
    std::vector<hipStream_t> streams;
    bool quit = false;
};

static double nowMs()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void pipelineRelease(DcsPipeline *p, std::shared_ptr<DcsPipeline::Job> &job)
{
    if (job && job->batch)
    {
        dcs_batch_destroy(job->batch);
        job->batch = nullptr;
    }
    if (job && job->hBlob)
    {
        cacheFree(p->ctx, true, job->hBlob, job->hBlobCap);
        job->hBlob = nullptr;
    }
    job.reset();
}

// The index pass of one list on the device (dcsIndexKernel, one lane per stream) instead of the host pool: the streams
// go up as they are, the records come back, and the host is left with parameters, planner and packer.  A lane walks a
// frame some 50 times slower than a host core does, so one list takes longer this way -- but the GPU has lanes to spare
// and the host has not: with enough lists in flight the walks of different lists overlap each other and the decode
// kernels, and the host cores, which the index pass otherwise keeps busy nine tenths of the time, prepare other lists
// meanwhile.  Returns DCS_OK with *usable = false when the list has to take the host path (a stream that runs past
// its buffer: its missing bytes read as zero, which the streams laid end to end cannot express).
static DcsStatus pipelineIndexOnDevice(DcsPipeline *p, DcsPipeline::Job *job, hipStream_t stream, bool *usable,
                                       double *deviceMs)
{
    *usable = false;
    DcsCtx *ctx = p->ctx;
    const uint32_t n = job->nStreams;
    std::vector<DcsStreamLoc> locs(n);
    std::vector<uint64_t> firstRecord(n), streamOff(n);
    uint64_t totalRec = 0;
    size_t blobLen = 0;
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
        locs[k].off = blobLen; locs[k].len = static_cast<uint32_t>(len); locs[k].os = sr.os; locs[k].firstRecord = totalRec;
        streamOff[k] = blobLen;
        firstRecord[k] = totalRec;
        blobLen += len;
        totalRec += nFrames;
    }
    const size_t blobCap = ((blobLen + 3 + 4) & ~size_t(3)) + 64;       // (zero tail: the packer copies whole dwords)
    const size_t recBytes = sizeof(DcsFrameIndex) * totalRec, infoBytes = sizeof(DcsStreamInfo) * n;
    void *hRec = nullptr, *hInfo = nullptr;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, cacheAlloc(ctx, true, reinterpret_cast<void **>(&job->hBlob), blobCap));
        job->hBlobCap = blobCap; job->hBlobLen = blobLen;
        HIPCHK(ctx, cacheAlloc(ctx, true, &hRec, recBytes));
        HIPCHK(ctx, cacheAlloc(ctx, true, &hInfo, infoBytes));
        return DCS_OK;
    }();
    if (st == DCS_OK)
    {
        memset(job->hBlob + blobLen, 0, blobCap - blobLen);
        for (uint32_t k = 0 ; k < n ; ++k)
        {
            if (k + 1 < n)      // (the alignment gap in front of the next stream)
                memset(job->hBlob + locs[k].off + locs[k].len, 0, static_cast<size_t>(locs[k + 1].off - locs[k].off) - locs[k].len);
            memcpy(job->hBlob + locs[k].off, job->streams[k].data, locs[k].len);
        }
        const double t0 = nowMs();
        st = gpuIndexOnStream(ctx, stream, job->hBlob, blobLen, locs.data(), n, static_cast<DcsFrameIndex *>(hRec), totalRec,
                              static_cast<DcsStreamInfo *>(hInfo));
        *deviceMs += nowMs() - t0;
    }
    if (st == DCS_OK)
    {
        const DcsStreamInfo *infos = static_cast<const DcsStreamInfo *>(hInfo);
        bool ok = true;
        for (uint32_t k = 0 ; k < n && ok ; ++k)
            ok = infos[k].nFrames != 0 && static_cast<size_t>(infos[k].nBytes) <= locs[k].len;
        if (ok)
        {
            const DcsPreIndexed pre{ static_cast<const DcsFrameIndex *>(hRec), firstRecord.data(), infos, streamOff.data() };
            st = dcsBuildStreams(job->streams, n, job->extraFrames, job->built, false, false, &pre);
            *usable = st == DCS_OK;
        }
    }
    cacheFree(ctx, true, hRec, recBytes);
    cacheFree(ctx, true, hInfo, infoBytes);
    return st;
}

static void pipelineWorker(DcsPipeline *p, int id)
{
    (void)hipSetDevice(p->ctx->device);
    for (;;)
    {
        std::shared_ptr<DcsPipeline::Job> job;
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->work.wait(lk, [&] { return p->quit || !p->queue.empty(); });
            if (p->quit && p->queue.empty())
                return;
            job = p->queue.front();
            p->queue.pop_front();
        }
        const double t0 = nowMs();
        double indexDeviceMs = 0;
        bool onDevice = false;
        DcsStatus st = DCS_OK;
        if (p->flags & DCS_PIPE_INDEX_ON_DEVICE)
            st = pipelineIndexOnDevice(p, job.get(), p->streams[id], &onDevice, &indexDeviceMs);
        if (st == DCS_OK && !onDevice)
        {
            job->built = DcsBuiltStreams();
            st = dcsBuildStreams(job->streams, job->nStreams, job->extraFrames, job->built, false, false);
        }
        const uint8_t *blob = onDevice ? job->hBlob : job->built.blob.data();
        const size_t blobLen = onDevice ? job->hBlobLen : job->built.blob.size();
        double t1 = nowMs(), t2 = t1;
        // (second attempt, tails by re-decoding the predecessor, only after a lost tail: see dcs_decode_batch)
        for (int attempt = 0 ; st == DCS_OK && attempt < 2 ; ++attempt)
        {
            const bool handoff = p->ctx->handoff && attempt == 0;
            const DcsBuiltStreams &B = job->built;
            st = createBatch(p->ctx, blob, blobLen, B.srcs.data(), static_cast<uint32_t>(B.srcs.size()),
                             B.jobs.data(), static_cast<uint32_t>(B.jobs.size()), nullptr, 0, p->streams[id], handoff, &job->batch);
            t2 = nowMs();
            if (st == DCS_OK) st = dcs_batch_run(job->batch, nullptr);
            if (st == DCS_OK) st = dcs_batch_download_view(job->batch, &job->pcm, &job->err);
            bool lost = false;
            if (st == DCS_OK && handoff)
                for (size_t j = 0 ; j < B.jobs.size() && !lost ; ++j)
                    lost = (job->err[j] & DCS_FRAME_TAIL_LOST) != 0;
            if (!lost)
                break;
            dcs_batch_destroy(job->batch);
            job->batch = nullptr;
        }
        const double t3 = nowMs();
        job->hostMs = (t2 - t0) - indexDeviceMs;
        job->deviceMs = (t3 - t2) + indexDeviceMs;
        {
            std::lock_guard<std::mutex> lk(p->m);
            job->status = st;
            job->done = true;
        }
        p->finished.notify_all();
    }
}

extern "C" DcsStatus dcs_pipeline_create(DcsCtx *ctx, int depth, uint32_t flags, DcsPipeline **out)
{
    if (ctx == nullptr || out == nullptr || depth < 1 || depth > 64 || (flags & ~DCS_PIPE_INDEX_ON_DEVICE) != 0)
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;
    DcsPipeline *p = new (std::nothrow) DcsPipeline;
    if (p == nullptr)
        return DCS_ERR_NO_MEMORY;
    p->ctx = ctx;
    p->depth = depth;
    p->flags = flags;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int i = 0 ; i < depth ; ++i)
    {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
        {
            for (hipStream_t t : p->streams) (void)hipStreamDestroy(t);
            delete p;
            ctx->lastError = "dcs_pipeline_create: hipStreamCreate failed";
            return DCS_ERR_HIP;
        }
        p->streams.push_back(s);
    }
    for (int i = 0 ; i < depth ; ++i)
        p->workers.emplace_back(pipelineWorker, p, i);
    *out = p;
    return DCS_OK;
}

extern "C" void dcs_pipeline_destroy(DcsPipeline *p)
{
    if (p == nullptr)
        return;
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->quit = true;
    }
    p->work.notify_all();
    for (std::thread &w : p->workers)
        w.join();
    (void)hipSetDevice(p->ctx->device);
    pipelineRelease(p, p->held);
    for (std::shared_ptr<DcsPipeline::Job> &j : p->order)
        pipelineRelease(p, j);
    for (hipStream_t s : p->streams)
        (void)hipStreamDestroy(s);
    delete p;
}

extern "C" DcsStatus dcs_pipeline_submit(DcsPipeline *p, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames)
{
    if (p == nullptr || streams == nullptr || nStreams == 0)
        return DCS_ERR_INVALID_ARG;
    std::shared_ptr<DcsPipeline::Job> job = std::make_shared<DcsPipeline::Job>();
    job->streams = streams; job->nStreams = nStreams; job->extraFrames = extraFrames;
    {
        std::unique_lock<std::mutex> lk(p->m);
        // at most `depth` jobs between submit and collect (each holds device and pinned buffers)
        p->room.wait(lk, [&] { return static_cast<int>(p->order.size()) < p->depth; });
        p->queue.push_back(job);
        p->order.push_back(job);
    }
    p->work.notify_one();
    return DCS_OK;
}

extern "C" DcsStatus dcs_pipeline_collect(DcsPipeline *p, DcsPipelineResult *out)
{
    if (p == nullptr || out == nullptr)
        return DCS_ERR_INVALID_ARG;
    (void)hipSetDevice(p->ctx->device);
    pipelineRelease(p, p->held);                    // the previous result's buffers go back to the context's cache
    std::shared_ptr<DcsPipeline::Job> job;
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
    out->deviceMs = static_cast<float>(job->deviceMs);
    if (job->status == DCS_OK)
    {
        out->pcm = job->pcm;
        out->err = job->err;
        out->frameOffsets = job->built.firstJob.data();
        out->nFrames = static_cast<uint32_t>(job->built.jobs.size());
    }
    return job->status;
}

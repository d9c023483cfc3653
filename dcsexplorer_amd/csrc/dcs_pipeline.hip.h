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
    };
    DcsCtx *ctx = nullptr;
    int depth = 0;
    std::mutex m;
    std::condition_variable work, finished, room;
    std::deque<std::shared_ptr<Job>> queue;         // submitted, not yet taken by a worker
    std::deque<std::shared_ptr<Job>> order;         // submitted, not yet collected (submission order)
    std::shared_ptr<Job> held;                      // the job whose result the caller is reading
    std::vector<std::thread> workers;
    std::vector<hipStream_t> streams;
    bool quit = false;
};

static void pipelineRelease(DcsPipeline *p, std::shared_ptr<DcsPipeline::Job> &job)
{
    if (job && job->batch)
    {
        dcs_batch_destroy(job->batch);
        job->batch = nullptr;
    }
    job.reset();
}

static double nowMs()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
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
        DcsStatus st = dcsBuildStreams(job->streams, job->nStreams, job->extraFrames, job->built, false, false);
        double t1 = nowMs(), t2 = t1;
        // (second attempt, tails by re-decoding the predecessor, only after a lost tail: see dcs_decode_batch)
        for (int attempt = 0 ; st == DCS_OK && attempt < 2 ; ++attempt)
        {
            const bool handoff = p->ctx->handoff && attempt == 0;
            const DcsBuiltStreams &B = job->built;
            st = createBatch(p->ctx, B.blob.data(), B.blob.size(), B.srcs.data(), static_cast<uint32_t>(B.srcs.size()),
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
        job->hostMs = (t1 - t0) + (t2 - t1);
        job->deviceMs = t3 - t2;
        {
            std::lock_guard<std::mutex> lk(p->m);
            job->status = st;
            job->done = true;
        }
        p->finished.notify_all();
    }
}

extern "C" DcsStatus dcs_pipeline_create(DcsCtx *ctx, int depth, DcsPipeline **out)
{
    if (ctx == nullptr || out == nullptr || depth < 1 || depth > 16)
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;
    DcsPipeline *p = new (std::nothrow) DcsPipeline;
    if (p == nullptr)
        return DCS_ERR_NO_MEMORY;
    p->ctx = ctx;
    p->depth = depth;
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

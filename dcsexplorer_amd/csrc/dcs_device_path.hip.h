// dcs_device_path.hip.h -- the WHOLE device path of a list of streams, resident: stream bytes in HBM -> PCM in HBM.
// Included at the end of dcs_runtime.hip, behind dcs_pipeline.hip.h (it uses the runtime's batch internals and the
// pipeline's stream table).
//
// dcs_batch_* times the decode kernel over a batch that has been indexed, planned and packed beforehand.  What one GPU does
// from the bytes of whole streams -- the reference's GetStreamInfo walk (DCSDecoderNative.cpp:1486-1537) to find every
// frame, then DecompressFrame + TransformFrame for each (:1546-1589, :272-278) -- is four kernels on one HIP stream:
//     dcsIndexWaveKernel (one wavefront per stream)  ->  dcsPlanKernel (one thread per chunk)
//         ->  dcsPackKernel (one wavefront per chunk)  ->  dcsDecodeKernel<FPW>
// with nothing crossing PCIe: the streams were uploaded when the object was created, the index records, the plan, the
// packages and the PCM stay in HBM.  dcs_device_path_run queues `iters` such passes back to back and reports the time of a
// pass, and the time of every kernel by HIP events around it.  This is what the pipeline queues per list
// (DCS_PIPE_PLAN_ON_DEVICE), without the pipeline's threads and copies.
#pragma once

struct DcsDevicePath
{
    DcsCtx *ctx = nullptr;
    uint32_t nStreams = 0, extraFrames = 0;
    uint64_t totalRec = 0, nJobs = 0;
    std::vector<uint32_t> firstJob;
    void *dBlob = nullptr, *dRec = nullptr, *dInfo = nullptr, *dLocs = nullptr;
    size_t blobCap = 0, recBytes = 0, infoBytes = 0, locBytes = 0, blobLen = 0;
    DcsBatch *batch = nullptr;
    hipEvent_t ev[6] = { nullptr };
};

extern "C" void dcs_device_path_destroy(DcsDevicePath *d)
{
    if (d == nullptr)
        return;
    DcsCtx *ctx = d->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (d->batch) dcs_batch_destroy(d->batch);
    if (d->dBlob) cacheFree(ctx, false, d->dBlob, d->blobCap);
    if (d->dRec) cacheFree(ctx, false, d->dRec, d->recBytes);
    if (d->dInfo) cacheFree(ctx, false, d->dInfo, d->infoBytes);
    if (d->dLocs) cacheFree(ctx, false, d->dLocs, d->locBytes);
    for (hipEvent_t e : d->ev)
        if (e) (void)hipEventDestroy(e);
    delete d;
}

extern "C" DcsStatus dcs_device_path_create(DcsCtx *ctx, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames, DcsDevicePath **out)
{
    if (ctx == nullptr || streams == nullptr || nStreams == 0 || out == nullptr)
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;
    DcsDevicePath *d = new (std::nothrow) DcsDevicePath;
    if (d == nullptr)
        return DCS_ERR_NO_MEMORY;
    d->ctx = ctx; d->nStreams = nStreams; d->extraFrames = extraFrames;
    // the streams end to end, each on a 4-byte boundary (what pipelineUpload lays out per list)
    std::vector<DcsStreamLoc> locs(nStreams);
    std::vector<uint64_t> firstRecord(nStreams);
    size_t blobLen = 0;
    uint64_t totalRec = 0;
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        const DcsStreamRef &sr = streams[k];
        if (sr.data == nullptr || sr.len < 3 || sr.os < DCS_OS93A || sr.os > DCS_OS95)
        {
            delete d;
            return DCS_ERR_INVALID_ARG;
        }
        const uint32_t nFrames = (static_cast<uint32_t>(sr.data[0]) << 8) | sr.data[1];
        if (nFrames == 0)
        {
            delete d;
            return DCS_ERR_BAD_STREAM;
        }
        const size_t most = 2 + 16 + (static_cast<size_t>(nFrames) * DCS_MAX_FRAME_BITS + 7) / 8 + 8;
        const size_t len = sr.len < most ? sr.len : most;
        blobLen = (blobLen + 3) & ~size_t(3);
        locs[k].off = blobLen; locs[k].len = static_cast<uint32_t>(len); locs[k].os = sr.os; locs[k].firstRecord = totalRec;
        firstRecord[k] = totalRec;
        blobLen += len;
        totalRec += nFrames;
    }
    d->totalRec = totalRec;
    d->blobLen = blobLen;
    d->blobCap = ((blobLen + 3) & ~size_t(3)) + 64;
    std::vector<DcsPlanStream> table;
    uint64_t nJobs = 0, payload = 0;
    bool all94 = true, has93a = false;
    DcsStatus st = planTableFor(streams, nStreams, extraFrames, locs.data(), firstRecord.data(), table, d->firstJob, &nJobs, &payload, &all94, &has93a);
    if (st == DCS_OK && (nJobs > 0xFFFFFFFFull || totalRec > 0xFFFFFFFFull))
        st = DCS_ERR_CAPACITY;
    if (st != DCS_OK)
    {
        delete d;
        return st;
    }
    d->nJobs = nJobs;
    std::vector<uint8_t> blob(d->blobCap, 0);
    for (uint32_t k = 0 ; k < nStreams ; ++k)
        memcpy(blob.data() + locs[k].off, streams[k].data, locs[k].len);
    st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        d->recBytes = sizeof(DcsFrameIndex) * (totalRec ? totalRec : 1);
        d->infoBytes = sizeof(DcsStreamInfo) * nStreams;
        d->locBytes = sizeof(DcsStreamLoc) * nStreams;
        HIPCHK(ctx, cacheAlloc(ctx, false, &d->dBlob, d->blobCap));
        HIPCHK(ctx, cacheAlloc(ctx, false, &d->dRec, d->recBytes));
        HIPCHK(ctx, cacheAlloc(ctx, false, &d->dInfo, d->infoBytes));
        HIPCHK(ctx, cacheAlloc(ctx, false, &d->dLocs, d->locBytes));
        HIPCHK(ctx, hipMemcpyAsync(d->dBlob, blob.data(), d->blobCap, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d->dLocs, locs.data(), d->locBytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d->dRec, 0, d->recBytes, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (hipEvent_t &e : d->ev)
            HIPCHK(ctx, hipEventCreate(&e));
        // the first pass: index walk, then the batch (which queues planner and packer behind it) and its decode launch
        HIPCHK(ctx, launchIndexWave(ctx->stream, reinterpret_cast<uintptr_t>(d->dBlob), static_cast<const DcsStreamLoc *>(d->dLocs), nStreams, ctx->dTables,
                                    static_cast<DcsFrameIndex *>(d->dRec), static_cast<DcsStreamInfo *>(d->dInfo), nullptr));
        return DCS_OK;
    }();
    if (st == DCS_OK)
        st = createBatchPlannedOnDevice(ctx, table.data(), nStreams, extraFrames, static_cast<uint32_t>(nJobs), static_cast<uint32_t>(totalRec), all94, has93a,
                                        payload, static_cast<const DcsFrameIndex *>(d->dRec), static_cast<const DcsStreamInfo *>(d->dInfo),
                                        static_cast<const uint8_t *>(d->dBlob), d->blobLen, ctx->stream, &d->batch);
    if (st == DCS_OK) st = dcs_batch_run(d->batch, nullptr);
    if (st == DCS_OK) st = batchQueuePlanFlag(d->batch);
    if (st == DCS_OK) st = dcs_batch_sync(d->batch);
    if (st == DCS_OK && batchPlanFlag(d->batch) != 0)
    {
        setError(ctx, "dcs_device_path_create: the device planner cannot serve this list (DCS_PLAN_* flags " + std::to_string(batchPlanFlag(d->batch)) + ")");
        st = DCS_ERR_BAD_STREAM;
    }
    if (st != DCS_OK)
    {
        dcs_device_path_destroy(d);
        return st;
    }
    *out = d;
    return DCS_OK;
}

// one pass: index walk, planner, packer, decode, on the context's stream; e = five events around the four kernels (or null)
static DcsStatus devicePathPass(DcsDevicePath *d, hipEvent_t *e)
{
    DcsCtx *ctx = d->ctx;
    DcsBatch *b = d->batch;
    if (e) HIPCHK(ctx, hipEventRecord(e[0], ctx->stream));
    HIPCHK(ctx, launchIndexWave(ctx->stream, reinterpret_cast<uintptr_t>(d->dBlob), static_cast<const DcsStreamLoc *>(d->dLocs), d->nStreams, ctx->dTables,
                                static_cast<DcsFrameIndex *>(d->dRec), static_cast<DcsStreamInfo *>(d->dInfo), nullptr));
    if (e) HIPCHK(ctx, hipEventRecord(e[1], ctx->stream));
    DcsStatus st = queuePlanAndPack(b, d->nStreams, d->extraFrames, static_cast<const DcsFrameIndex *>(d->dRec), static_cast<const DcsStreamInfo *>(d->dInfo),
                                    static_cast<const uint8_t *>(d->dBlob), d->blobLen, e ? e[2] : nullptr);
    if (st != DCS_OK)
        return st;
    if (e) HIPCHK(ctx, hipEventRecord(e[3], ctx->stream));
    st = launchOnce(b, ctx->stream);
    if (st != DCS_OK)
        return st;
    if (e) HIPCHK(ctx, hipEventRecord(e[4], ctx->stream));
    return DCS_OK;
}

extern "C" DcsStatus dcs_device_path_run(DcsDevicePath *d, int iters, DcsDevicePathTimes *t)
{
    if (d == nullptr || iters < 1 || t == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = d->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(t, 0, sizeof(*t));
    // `iters` passes back to back: the time of a pass
    HIPCHK(ctx, hipEventRecord(d->ev[0], ctx->stream));
    for (int i = 0 ; i < iters ; ++i)
    {
        const DcsStatus st = devicePathPass(d, nullptr);
        if (st != DCS_OK)
            return st;
    }
    HIPCHK(ctx, hipEventRecord(d->ev[5], ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(d->ev[5]));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, d->ev[0], d->ev[5]));
    t->passMs = ms / static_cast<float>(iters);
    // a few passes with events around every kernel
    const int n = iters < 5 ? iters : 5;
    for (int i = 0 ; i < n ; ++i)
    {
        DcsStatus st = devicePathPass(d, d->ev);
        if (st != DCS_OK)
            return st;
        HIPCHK(ctx, hipEventSynchronize(d->ev[4]));
        float k[4] = { 0 };
        for (int j = 0 ; j < 4 ; ++j)
            HIPCHK(ctx, hipEventElapsedTime(&k[j], d->ev[j], d->ev[j + 1]));
        t->indexMs += k[0] / n; t->planMs += k[1] / n; t->packMs += k[2] / n; t->decodeMs += k[3] / n;
    }
    {
        const DcsStatus st = markLaunched(d->batch, ctx->stream);
        if (st != DCS_OK)
            return st;
    }
    DcsStatus st = batchQueuePlanFlag(d->batch);
    if (st == DCS_OK) st = dcs_batch_sync(d->batch);
    if (st != DCS_OK)
        return st;
    t->planFlags = batchPlanFlag(d->batch);
    t->nStreams = d->nStreams;
    t->nFrames = static_cast<uint32_t>(d->nJobs);
    t->framesPerWave = static_cast<uint32_t>(d->batch->fpw);
    t->algorithmicBytes = d->batch->algoBytes;
    return DCS_OK;
}

// `iters` passes back to back and a wait for the last: nothing is timed here (a caller with several paths in flight, each driven
// by a thread of its own, takes the wall clock around all of them)
extern "C" DcsStatus dcs_device_path_run_many(DcsDevicePath *d, int iters)
{
    if (d == nullptr || iters < 1)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = d->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int i = 0 ; i < iters ; ++i)
    {
        const DcsStatus st = devicePathPass(d, nullptr);
        if (st != DCS_OK)
            return st;
    }
    const DcsStatus st = markLaunched(d->batch, ctx->stream);
    return st != DCS_OK ? st : dcs_batch_sync(d->batch);
}

extern "C" DcsStatus dcs_device_path_download(DcsDevicePath *d, int16_t *pcmOut, uint32_t *errOut, uint32_t *frameOffsets)
{
    if (d == nullptr)
        return DCS_ERR_INVALID_ARG;
    if (frameOffsets != nullptr)
        for (uint32_t k = 0 ; k <= d->nStreams ; ++k)
            frameOffsets[k] = d->firstJob[k];
    return dcs_batch_download(d->batch, pcmOut, errOut, nullptr);
}

// dcs_runtime.hip -- context, resident batches and kernel launches of libdcs_hip.so (C ABI of
// include/dcs_hip.h).  The product path has no CPU decode: without a usable gfx950 device
// dcs_ctx_create fails and nothing below can run.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <new>
#include "dcs_common.h"
#include "dcs_kernels.hip.h"

struct DcsCtx
{
    int device = 0;
    hipStream_t stream = nullptr;
    DcsDevTables *dTables = nullptr;
    int fpwOverride = 0;
    int numCUs = 256;
    std::string lastError;
};

struct DcsBatch
{
    DcsCtx *ctx = nullptr;
    uint32_t nJobs = 0, nSrcs = 0, nTailsIn = 0;
    size_t blobLen = 0;
    int fpw = 0;
    uint32_t nChunks = 0;
    uint64_t algoBytes = 0;
    // device buffers
    uint8_t *dBlob = nullptr;
    DcsSrcDesc *dSrcs = nullptr;
    DcsSlot *dSlots = nullptr;
    int16_t *dTailsIn = nullptr;
    int16_t *dPcm = nullptr;
    uint32_t *dErr = nullptr;
    int16_t *dTailsOut = nullptr;
    unsigned long long *dDebug = nullptr;   // DCS_STAMPS builds only
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static std::string g_createError;

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[256];                                                                      \
            snprintf(buf_, sizeof(buf_), "%s failed: %s", #call, hipGetErrorString(e_));         \
            (ctx)->lastError = buf_;                                                             \
            return DCS_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

extern "C" uint32_t dcs_abi_version(void) { return DCS_ABI_VERSION; }

extern "C" int dcs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" const char *dcs_last_error(const DcsCtx *ctx)
{
    return ctx ? ctx->lastError.c_str() : g_createError.c_str();
}

extern "C" DcsStatus dcs_ctx_create(int deviceId, DcsCtx **out)
{
    if (out == nullptr)
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
    {
        g_createError = "no HIP device available (this library has no CPU fallback)";
        return DCS_ERR_NO_DEVICE;
    }
    if (deviceId < 0 || deviceId >= n)
    {
        g_createError = "device id out of range";
        return DCS_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, deviceId) != hipSuccess)
    {
        g_createError = "hipGetDeviceProperties failed";
        return DCS_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        g_createError = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return DCS_ERR_NO_DEVICE;
    }

    DcsCtx *ctx = new (std::nothrow) DcsCtx;
    if (ctx == nullptr)
        return DCS_ERR_NO_MEMORY;
    ctx->device = deviceId;
    ctx->numCUs = prop.multiProcessorCount;
    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(deviceId));
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
        g_createError = ctx->lastError;
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
    if (ctx->dTables) (void)hipFree(ctx->dTables);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" DcsStatus dcs_ctx_set_frames_per_wave(DcsCtx *ctx, int fpw)
{
    if (ctx == nullptr || !(fpw == 0 || fpw == 4 || fpw == 8 || fpw == 16))
        return DCS_ERR_INVALID_ARG;
    ctx->fpwOverride = fpw;
    return DCS_OK;
}

// frames per wavefront.  Four lanes unpack one frame, so 16 frames fill the 64 lanes; small batches
// use 8 or 4 frames per wavefront: more wavefronts, and a shorter serial path in each.
static int chooseFpw(const DcsCtx *ctx, uint32_t nJobs)
{
    if (ctx->fpwOverride != 0)
        return ctx->fpwOverride;
    const uint64_t simds = static_cast<uint64_t>(ctx->numCUs) * 4;
    return nJobs >= simds * 16 ? 16 : nJobs >= simds * 6 ? 8 : 4;
}

extern "C" void dcs_batch_destroy(DcsBatch *b)
{
    if (b == nullptr)
        return;
    (void)hipSetDevice(b->ctx->device);
    void *ptrs[] = { b->dBlob, b->dSrcs, b->dSlots, b->dTailsIn, b->dPcm, b->dErr, b->dTailsOut, b->dDebug };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    delete b;
}

extern "C" DcsStatus dcs_batch_create(DcsCtx *ctx,
                                      const uint8_t *blob, size_t blobLen,
                                      const DcsSrcDesc *srcs, uint32_t nSrcs,
                                      const DcsFrameJob *jobs, uint32_t nJobs,
                                      const int16_t *tailsIn, uint32_t nTailsIn,
                                      DcsBatch **out)
{
    if (ctx == nullptr || out == nullptr || jobs == nullptr || nJobs == 0 || (nSrcs != 0 && (srcs == nullptr || blob == nullptr)))
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;

    // validate the description on the host: the kernel trusts indices and formats
    uint64_t payloadBits = 0;
    for (uint32_t j = 0 ; j < nJobs ; ++j)
    {
        const DcsFrameJob &jb = jobs[j];
        if (jb.nSrc > DCS_MAX_CHANNELS || jb.volShift > 8 || jb.xform > DCS_XFORM_94
            || (jb.nSrc != 0 && (jb.firstSrc >= nSrcs || jb.firstSrc + jb.nSrc > nSrcs)))
        {
            ctx->lastError = "job " + std::to_string(j) + ": bad source range / volShift / xform";
            return DCS_ERR_INVALID_ARG;
        }
        if (jb.prev != DCS_PREV_NONE)
        {
            const bool ext = (jb.prev & DCS_PREV_EXT) != 0;
            if (ext ? ((jb.prev & 0x7FFFFFFFu) >= nTailsIn || tailsIn == nullptr) : (jb.prev >= nJobs || jb.prev == j))
            {
                ctx->lastError = "job " + std::to_string(j) + ": bad overlap predecessor";
                return DCS_ERR_INVALID_ARG;
            }
        }
    }
    for (uint32_t s = 0 ; s < nSrcs ; ++s)
    {
        const DcsSrcDesc &sd = srcs[s];
        if (sd.format > DCS_FMT_94_T1_S3 || (sd.hdrLen != 16 && sd.hdrLen != 1) || sd.streamOff + 2 + sd.hdrLen > blobLen)
        {
            ctx->lastError = "source " + std::to_string(s) + ": bad format / header length / stream offset";
            return DCS_ERR_INVALID_ARG;
        }
        payloadBits += sd.idx.nBits;
    }

    DcsBatch *b = new (std::nothrow) DcsBatch;
    if (b == nullptr)
        return DCS_ERR_NO_MEMORY;
    b->ctx = ctx;
    b->nJobs = nJobs; b->nSrcs = nSrcs; b->nTailsIn = nTailsIn; b->blobLen = blobLen;
    b->fpw = chooseFpw(ctx, nJobs);

    std::vector<DcsSlot> slots;
    b->nChunks = dcsPlanChunks(jobs, nJobs, srcs, b->fpw, slots);

    // algorithmic bytes: compressed payload + descriptors read (one DcsSrcDesc per source, one
    // 16-byte job record per frame -- the device reads it in its DcsSlot form), PCM written
    b->algoBytes = (payloadBits + 7) / 8 + static_cast<uint64_t>(nSrcs) * sizeof(DcsSrcDesc)
                 + static_cast<uint64_t>(nJobs) * sizeof(DcsFrameJob) + static_cast<uint64_t>(nJobs) * DCS_FRAME_SAMPLES * 2;

    DcsStatus st = [&]() -> DcsStatus {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        const size_t blobAlloc = ((blobLen + 3) & ~size_t(3)) + 64;         // zero tail: the bit reader prefetches past the end
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dBlob), blobAlloc));
        HIPCHK(ctx, hipMemsetAsync(b->dBlob, 0, blobAlloc, ctx->stream));
        if (blobLen)
            HIPCHK(ctx, hipMemcpyAsync(b->dBlob, blob, blobLen, hipMemcpyHostToDevice, ctx->stream));
        if (nSrcs)
        {
            HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dSrcs), sizeof(DcsSrcDesc) * nSrcs));
            HIPCHK(ctx, hipMemcpyAsync(b->dSrcs, srcs, sizeof(DcsSrcDesc) * nSrcs, hipMemcpyHostToDevice, ctx->stream));
        }
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dSlots), sizeof(DcsSlot) * slots.size()));
        HIPCHK(ctx, hipMemcpyAsync(b->dSlots, slots.data(), sizeof(DcsSlot) * slots.size(), hipMemcpyHostToDevice, ctx->stream));
        if (nTailsIn)
        {
            HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dTailsIn), sizeof(int16_t) * 16 * nTailsIn));
            HIPCHK(ctx, hipMemcpyAsync(b->dTailsIn, tailsIn, sizeof(int16_t) * 16 * nTailsIn, hipMemcpyHostToDevice, ctx->stream));
        }
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dPcm), sizeof(int16_t) * DCS_FRAME_SAMPLES * nJobs));
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dErr), sizeof(uint32_t) * nJobs));
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dTailsOut), sizeof(int16_t) * 16 * nJobs));
        HIPCHK(ctx, hipMemsetAsync(b->dErr, 0, sizeof(uint32_t) * nJobs, ctx->stream));
#ifdef DCS_STAMPS
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&b->dDebug), sizeof(unsigned long long) * 8 * (b->nChunks + 4)));
        HIPCHK(ctx, hipMemsetAsync(b->dDebug, 0, sizeof(unsigned long long) * 8 * (b->nChunks + 4), ctx->stream));
#endif
        HIPCHK(ctx, hipEventCreate(&b->ev0));
        HIPCHK(ctx, hipEventCreate(&b->ev1));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return DCS_OK;
    }();
    if (st != DCS_OK)
    {
        dcs_batch_destroy(b);
        return st;
    }
    *out = b;
    return DCS_OK;
}

template <int FPW>
static hipError_t launch(const DcsKernelArgs &args, hipStream_t stream)
{
    // DCS_EXTRA_LDS (bytes): experiment knob, lowers occupancy by requesting unused LDS
    static const int extraLds = getenv("DCS_EXTRA_LDS") ? atoi(getenv("DCS_EXTRA_LDS")) : 0;
    const uint32_t blocks = (args.nChunks + dcsk::kWavesPerBlock - 1) / dcsk::kWavesPerBlock;
    dcsk::dcsDecodeKernel<FPW><<<dim3(blocks), dim3(64 * dcsk::kWavesPerBlock), dcsk::ldsBytes(FPW) + extraLds, stream>>>(args);
    return hipGetLastError();
}

extern "C" DcsStatus dcs_batch_run(DcsBatch *b, void *hipStream)
{
    if (b == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    hipStream_t stream = hipStream ? static_cast<hipStream_t>(hipStream) : ctx->stream;
    DcsKernelArgs args;
    args.blob = b->dBlob;
    args.blobLen = b->blobLen;
    args.srcs = b->dSrcs;
    args.slots = b->dSlots;
    args.nChunks = b->nChunks;
    args.nJobs = b->nJobs;
    args.pcm = b->dPcm;
    args.err = b->dErr;
    args.tailsIn = b->dTailsIn;
    args.tailsOut = b->dTailsOut;
    args.tables = ctx->dTables;
    args.debug = b->dDebug;
    hipError_t e;
    e = (b->fpw == 16) ? launch<16>(args, stream) : (b->fpw == 8) ? launch<8>(args, stream) : launch<4>(args, stream);
    if (e != hipSuccess)
    {
        ctx->lastError = std::string("kernel launch failed: ") + hipGetErrorString(e);
        return DCS_ERR_HIP;
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_batch_time(DcsBatch *b, void *hipStream, int iters, float *avgMs)
{
    if (b == nullptr || iters < 1 || avgMs == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    hipStream_t stream = hipStream ? static_cast<hipStream_t>(hipStream) : ctx->stream;
    HIPCHK(ctx, hipEventRecord(b->ev0, stream));
    for (int i = 0 ; i < iters ; ++i)
    {
        DcsStatus st = dcs_batch_run(b, hipStream);
        if (st != DCS_OK)
            return st;
    }
    HIPCHK(ctx, hipEventRecord(b->ev1, stream));
    HIPCHK(ctx, hipEventSynchronize(b->ev1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *avgMs = ms / static_cast<float>(iters);
    return DCS_OK;
}

extern "C" DcsStatus dcs_batch_sync(DcsBatch *b)
{
    if (b == nullptr)
        return DCS_ERR_INVALID_ARG;
    HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
    return DCS_OK;
}

extern "C" DcsStatus dcs_batch_download(DcsBatch *b, int16_t *pcmOut, uint32_t *errOut, int16_t *tailsOut)
{
    if (b == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsCtx *ctx = b->ctx;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (pcmOut)
        HIPCHK(ctx, hipMemcpy(pcmOut, b->dPcm, sizeof(int16_t) * DCS_FRAME_SAMPLES * b->nJobs, hipMemcpyDeviceToHost));
    if (errOut)
        HIPCHK(ctx, hipMemcpy(errOut, b->dErr, sizeof(uint32_t) * b->nJobs, hipMemcpyDeviceToHost));
    if (tailsOut)
        HIPCHK(ctx, hipMemcpy(tailsOut, b->dTailsOut, sizeof(int16_t) * 16 * b->nJobs, hipMemcpyDeviceToHost));
    return DCS_OK;
}

extern "C" void *dcs_batch_device_pcm(DcsBatch *b) { return b ? b->dPcm : nullptr; }

#ifdef DCS_STAMPS
// diagnostic builds only: copy the per-chunk phase stamps (8 x uint64 per chunk) to the host
extern "C" int dcs_debug_stamps(DcsBatch *b, unsigned long long *out, uint32_t capChunks)
{
    if (b == nullptr || b->dDebug == nullptr) return -1;
    (void)hipStreamSynchronize(b->ctx->stream);
    const uint32_t n = b->nChunks < capChunks ? b->nChunks : capChunks;
    (void)hipMemcpy(out, b->dDebug, sizeof(unsigned long long) * 8 * n, hipMemcpyDeviceToHost);
    return static_cast<int>(n);
}
#endif
extern "C" uint64_t dcs_batch_algorithmic_bytes(const DcsBatch *b) { return b ? b->algoBytes : 0; }
extern "C" uint32_t dcs_batch_num_jobs(const DcsBatch *b) { return b ? b->nJobs : 0; }

extern "C" DcsStatus dcs_decode_batch(DcsCtx *ctx,
                                      const uint8_t *blob, size_t blobLen,
                                      const DcsSrcDesc *srcs, uint32_t nSrcs,
                                      const DcsFrameJob *jobs, uint32_t nJobs,
                                      const int16_t *tailsIn, uint32_t nTailsIn,
                                      int16_t *pcmOut, uint32_t *errOut, int16_t *tailsOut)
{
    DcsBatch *b = nullptr;
    DcsStatus st = dcs_batch_create(ctx, blob, blobLen, srcs, nSrcs, jobs, nJobs, tailsIn, nTailsIn, &b);
    if (st != DCS_OK)
        return st;
    st = dcs_batch_run(b, nullptr);
    if (st == DCS_OK)
        st = dcs_batch_download(b, pcmOut, errOut, tailsOut);
    dcs_batch_destroy(b);
    return st;
}

// dcs_node.hip.h -- several GPUs of one node behind one object: N persistent contexts, one dcs_pipeline each, lists dealt
// to the least-loaded device, results in submission order; threads and pinned buffers placed on each GPU's NUMA node.
// Included at the end of dcs_runtime.hip, behind dcs_pipeline.hip.h.
//
// The reference decodes its batch job (DCSExplorer.cpp:1628-1907) on one thread.  Streams are the independent units of
// the path, so several devices need nothing from each other: no collective, no peer copies.  What a node-level
// object adds over N separate pipelines is what the host side of N GPUs needs: contexts and pipelines that live across
// calls (dcs_decode_streams_sharded created and destroyed a context per device per call), load balance by frames in
// flight, one submission order, and host threads / pinned memory next to the GPU they feed.
#pragma once
#include <sched.h>
#include <map>
#include <memory>

// NUMA node of a HIP device (from its PCI address, /sys/bus/pci/devices/<addr>/numa_node), or -1
extern "C" int dcs_device_numa_node(int deviceId)
{
    char bus[64] = { 0 };
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || deviceId < 0 || deviceId >= count
        || hipDeviceGetPCIBusId(bus, sizeof(bus), deviceId) != hipSuccess)
    {
        (void)hipGetLastError();        // (a failed query must not be what the next kernel launch's hipGetLastError reports)
        return -1;
    }
    for (char *c = bus ; *c ; ++c)
        *c = static_cast<char>(tolower(*c));
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE *f = fopen(path.c_str(), "r");
    if (f == nullptr)
        return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1)
        node = -1;
    fclose(f);
    return node;
}

namespace {

// the CPUs of a NUMA node this process may use (its current affinity mask cut with the node's cpulist); empty = unknown
static bool numaCpus(int node, cpu_set_t *out)
{
    CPU_ZERO(out);
    if (node < 0)
        return false;
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (f == nullptr)
        return false;
    char buf[4096] = { 0 };
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
        return false;
    int any = 0;
    for (char *p = buf ; *p ; )
    {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p)
            break;
        long b = a;
        p = end;
        if (*p == '-')
        {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a ; c <= b && c < CPU_SETSIZE ; ++c)
            if (CPU_ISSET(static_cast<int>(c), &allowed))
            {
                CPU_SET(static_cast<int>(c), out);
                ++any;
            }
        while (*p == ',' || *p == '\n' || *p == ' ')
            ++p;
    }
    return any != 0;
}

// runs `fn` with the calling thread bound to the CPUs of `node` (threads it starts inherit the mask, and memory it pins
// or first touches is taken from that node under the default local policy); restores the mask afterwards
template <class F>
static auto onNumaNode(int node, F fn) -> decltype(fn())
{
    cpu_set_t old, want;
    const bool have = sched_getaffinity(0, sizeof(old), &old) == 0 && numaCpus(node, &want);
    if (have)
        (void)sched_setaffinity(0, sizeof(want), &want);
    auto r = fn();
    if (have)
        (void)sched_setaffinity(0, sizeof(old), &old);
    return r;
}

}   // namespace

struct DcsNode
{
    struct Dev
    {
        int id = 0, numa = -1;
        DcsCtx *ctx = nullptr;
        DcsPipeline *pipe = nullptr;            // created with the first list (dcs_node_submit)
        uint64_t framesInFlight = 0;
        int listsInFlight = 0;
        uint64_t listsDone = 0;
    };
    std::vector<Dev> devs;
    int depth = 0;
    uint32_t flags = 0;
    std::mutex m;
    std::condition_variable room;
    struct Pending { uint32_t dev; uint64_t frames; };
    std::deque<Pending> order;                  // submitted, not yet collected
    std::string lastError;
    std::mutex submitMutex;                     // dcs_node_submit: pipeline creation and the hand-over to a pipeline, one at a time
    std::mutex callMutex;                       // dcs_decode_streams_sharded: one call at a time per node
};

extern "C" void dcs_node_destroy(DcsNode *n)
{
    if (n == nullptr)
        return;
    for (DcsNode::Dev &d : n->devs)
    {
        if (d.pipe) dcs_pipeline_destroy(d.pipe);
        if (d.ctx) dcs_ctx_destroy(d.ctx);
    }
    delete n;
}

extern "C" DcsStatus dcs_node_create(const int *deviceIds, uint32_t nDevices, int depth, uint32_t flags, DcsNode **out)
{
    if (deviceIds == nullptr || nDevices == 0 || nDevices > 64 || out == nullptr || depth < 1 || depth > 64
        || (flags & ~(DCS_PIPE_INDEX_ON_DEVICE | DCS_PIPE_PACK_ON_DEVICE | DCS_PIPE_PLAN_ON_DEVICE)) != 0)
        return DCS_ERR_INVALID_ARG;
    *out = nullptr;
    DcsNode *n = new (std::nothrow) DcsNode;
    if (n == nullptr)
        return DCS_ERR_NO_MEMORY;
    n->depth = depth;
    n->flags = flags;
    n->devs.resize(nDevices);
    for (uint32_t d = 0 ; d < nDevices ; ++d)
    {
        DcsNode::Dev &dev = n->devs[d];
        dev.id = deviceIds[d];
        dev.numa = dcs_device_numa_node(dev.id);
        // (the context's stream, tables and -- later -- its pipelines' threads and pinned buffers come from this thread)
        const DcsStatus st = onNumaNode(dev.numa, [&] { return dcs_ctx_create(dev.id, &dev.ctx); });
        if (st != DCS_OK)
        {
            dcs_node_destroy(n);
            return st;
        }
    }
    *out = n;
    return DCS_OK;
}

extern "C" uint32_t dcs_node_num_devices(const DcsNode *n) { return n ? static_cast<uint32_t>(n->devs.size()) : 0u; }

extern "C" DcsStatus dcs_node_device_info(const DcsNode *n, uint32_t index, int *deviceId, int *numaNode, uint64_t *listsDone)
{
    if (n == nullptr || index >= n->devs.size())
        return DCS_ERR_INVALID_ARG;
    if (deviceId) *deviceId = n->devs[index].id;
    if (numaNode) *numaNode = n->devs[index].numa;
    if (listsDone) *listsDone = n->devs[index].listsDone;
    return DCS_OK;
}

extern "C" const char *dcs_node_last_error(const DcsNode *n)
{
    // a copy owned by the calling thread: submit and collect (two threads) both write the string
    thread_local std::string copy;
    if (n == nullptr)
        return "";
    std::lock_guard<std::mutex> lk(const_cast<DcsNode *>(n)->m);
    copy = n->lastError;
    return copy.c_str();
}

extern "C" DcsStatus dcs_node_submit(DcsNode *n, const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames)
{
    if (n == nullptr || streams == nullptr || nStreams == 0)
        return DCS_ERR_INVALID_ARG;
    uint64_t frames = 0;
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        if (streams[k].data == nullptr || streams[k].len < 3)
            return DCS_ERR_INVALID_ARG;
        frames += ((static_cast<uint64_t>(streams[k].data[0]) << 8) | streams[k].data[1]) + extraFrames;
    }
    uint32_t pick = 0;
    {
        // the device with the fewest frames in flight among those that have room for a list; wait while none has
        std::unique_lock<std::mutex> lk(n->m);
        for (;;)
        {
            bool found = false;
            for (uint32_t d = 0 ; d < n->devs.size() ; ++d)
                if (n->devs[d].listsInFlight < n->depth && (!found || n->devs[d].framesInFlight < n->devs[pick].framesInFlight))
                {
                    pick = d;
                    found = true;
                }
            if (found)
                break;
            n->room.wait(lk);
        }
        n->devs[pick].listsInFlight += 1;           // (the place is taken; the list joins `order` once its pipeline has it)
        n->devs[pick].framesInFlight += frames;
    }
    DcsNode::Dev &dev = n->devs[pick];
    DcsStatus st = DCS_OK;
    {
        // one submission at a time from here on: a device's pipeline is made once, and `order` is the order in which the
        // pipelines received their lists (dcs_node_collect may run on another thread and must find every list it is
        // told about already in its pipeline)
        std::lock_guard<std::mutex> sub(n->submitMutex);
        if (dev.pipe == nullptr)
            st = onNumaNode(dev.numa, [&] { return dcs_pipeline_create(dev.ctx, n->depth, n->flags, &dev.pipe); });
        if (st == DCS_OK)
            st = dcs_pipeline_submit(dev.pipe, streams, nStreams, extraFrames);
        std::lock_guard<std::mutex> lk(n->m);
        if (st == DCS_OK)
            n->order.push_back(DcsNode::Pending{ pick, frames });
        else
        {
            n->lastError = std::string("device ") + std::to_string(dev.id) + ": " + dcs_last_error(dev.ctx);
            dev.listsInFlight -= 1;
            dev.framesInFlight -= frames;
        }
    }
    if (st != DCS_OK)
        n->room.notify_all();
    return st;
}

extern "C" DcsStatus dcs_node_collect(DcsNode *n, DcsPipelineResult *out, int *deviceIndexOut)
{
    if (n == nullptr || out == nullptr)
        return DCS_ERR_INVALID_ARG;
    DcsNode::Pending p;
    {
        std::lock_guard<std::mutex> lk(n->m);
        if (n->order.empty())
            return DCS_ERR_INVALID_ARG;
        p = n->order.front();
    }
    DcsNode::Dev &dev = n->devs[p.dev];
    const DcsStatus st = dcs_pipeline_collect(dev.pipe, out);        // (a device's lists come back in its own submission order)
    {
        std::lock_guard<std::mutex> lk(n->m);
        n->order.pop_front();
        dev.listsInFlight -= 1;
        dev.framesInFlight -= p.frames;
        dev.listsDone += 1;
        if (st != DCS_OK)
            n->lastError = std::string("device ") + std::to_string(dev.id) + ": " + dcs_last_error(dev.ctx);
    }
    n->room.notify_all();
    if (deviceIndexOut != nullptr)
        *deviceIndexOut = static_cast<int>(p.dev);
    return st;
}

// ---------------------------------------------------------------------------------------------------------
// dcs_decode_streams_sharded on persistent contexts: one node per device list, kept for the life of the process (or until
// dcs_node_cache_release): ONE list cut into contiguous ranges balanced by frames, range d decoded by dcs_decode_streams on
// device d's context -- which takes a large range through the context's own pipeline in parts -- from a thread bound to
// that device's NUMA node.  No context, stream, table upload or pipeline is created per call any more.
// ---------------------------------------------------------------------------------------------------------
namespace {
static std::mutex g_nodeCacheMutex;
// key: the device ids SORTED ([0, 1] and [1, 0] are one node; a device named twice is two contexts).  shared_ptr: a call holds
// its node while it runs, so dcs_node_cache_release() from another thread only drops the cache's reference and the contexts
// go when the last call that uses them returns (the raw pointer was used after the cache's mutex was dropped).
static std::map<std::vector<int>, std::shared_ptr<DcsNode>> g_nodeCache;
// What a cached node's contexts may keep between calls: buffers of a couple of parts in flight, not the eighth of every card
// an occasional caller would otherwise lose for the life of the process (a context's default: min(32 GB, free / 8)).
static const uint64_t kShardedDevCache = uint64_t(2) << 30, kShardedPinCache = uint64_t(1) << 30;
}

extern "C" void dcs_node_cache_release(void)
{
    std::map<std::vector<int>, std::shared_ptr<DcsNode>> drop;
    {
        std::lock_guard<std::mutex> lk(g_nodeCacheMutex);
        drop.swap(g_nodeCache);
    }
    // (outside the mutex: a node whose last reference this was destroys its contexts here; one that a running call still holds
    // is destroyed by that call's thread when it returns)
}

extern "C" DcsStatus dcs_decode_streams_sharded(const int *deviceIds, uint32_t nDevices,
                                                const DcsStreamRef *streams, uint32_t nStreams, uint32_t extraFrames,
                                                int16_t *pcmOut, size_t pcmCapFrames, uint32_t *frameOffsets,
                                                uint32_t *errOut, uint32_t *firstStreamOfDevice)
{
    if (deviceIds == nullptr || nDevices == 0 || nDevices > 64 || streams == nullptr || nStreams == 0 || pcmOut == nullptr)
        return DCS_ERR_INVALID_ARG;
    // output frames per stream, from the U16 prefix (what dcs_decode_streams will produce)
    std::vector<uint32_t> frames(nStreams);
    std::vector<uint64_t> firstFrame(static_cast<size_t>(nStreams) + 1, 0);
    for (uint32_t k = 0 ; k < nStreams ; ++k)
    {
        if (streams[k].data == nullptr || streams[k].len < 3)
            return DCS_ERR_INVALID_ARG;
        frames[k] = ((static_cast<uint32_t>(streams[k].data[0]) << 8) | streams[k].data[1]) + extraFrames;
        firstFrame[k + 1] = firstFrame[k] + frames[k];
    }
    if (firstFrame[nStreams] > pcmCapFrames || firstFrame[nStreams] > 0xFFFFFFFFull)
        return DCS_ERR_CAPACITY;
    std::vector<uint32_t> cut(static_cast<size_t>(nDevices) + 1);
    DcsStatus st = dcs_partition_streams(frames.data(), nStreams, nDevices, cut.data());
    if (st != DCS_OK)
        return st;
    if (firstStreamOfDevice != nullptr)
        memcpy(firstStreamOfDevice, cut.data(), sizeof(uint32_t) * cut.size());

    // the caller's d-th device is context slot[d] of the node of the sorted list
    std::vector<uint32_t> byId(nDevices);
    for (uint32_t d = 0 ; d < nDevices ; ++d)
        byId[d] = d;
    std::stable_sort(byId.begin(), byId.end(), [&](uint32_t a, uint32_t b) { return deviceIds[a] < deviceIds[b]; });
    std::vector<int> key(nDevices);
    std::vector<uint32_t> slot(nDevices);
    for (uint32_t i = 0 ; i < nDevices ; ++i)
    {
        key[i] = deviceIds[byId[i]];
        slot[byId[i]] = i;
    }
    std::shared_ptr<DcsNode> node;
    {
        std::lock_guard<std::mutex> lk(g_nodeCacheMutex);
        auto it = g_nodeCache.find(key);
        if (it == g_nodeCache.end())
        {
            DcsNode *made = nullptr;
            st = dcs_node_create(key.data(), nDevices, 8, DCS_PIPE_ALL_ON_DEVICE, &made);
            if (st != DCS_OK)
                return st;
            node.reset(made, [](DcsNode *n) { dcs_node_destroy(n); });
            for (DcsNode::Dev &dev : node->devs)
                (void)dcs_ctx_set_cache_limits(dev.ctx, kShardedDevCache, kShardedPinCache);
            g_nodeCache[key] = node;
        }
        else
            node = it->second;
    }
    std::lock_guard<std::mutex> call(node->callMutex);
    // one host thread per device, every range writes its own part of the outputs
    std::vector<DcsStatus> status(nDevices, DCS_OK);
    std::vector<std::thread> workers;
    for (uint32_t d = 0 ; d < nDevices ; ++d)
    {
        const uint32_t lo = cut[d], hi = cut[d + 1];
        if (lo == hi)
            continue;
        workers.emplace_back([&, d, lo, hi]() {
            DcsNode::Dev &dev = node->devs[slot[d]];
            const uint64_t f0 = firstFrame[lo];
            status[d] = onNumaNode(dev.numa, [&] {
                return dcs_decode_streams(dev.ctx, streams + lo, hi - lo, extraFrames, pcmOut + f0 * DCS_FRAME_SAMPLES,
                                          static_cast<size_t>(firstFrame[hi] - f0), nullptr, errOut ? errOut + f0 : nullptr);
            });
        });
    }
    for (std::thread &w : workers)
        w.join();
    if (frameOffsets != nullptr)
        for (uint32_t k = 0 ; k <= nStreams ; ++k)
            frameOffsets[k] = static_cast<uint32_t>(firstFrame[k]);
    for (uint32_t d = 0 ; d < nDevices ; ++d)
        if (status[d] != DCS_OK)
        {
            std::lock_guard<std::mutex> lk(node->m);
            node->lastError = std::string("device ") + std::to_string(node->devs[slot[d]].id) + ": " + dcs_last_error(node->devs[slot[d]].ctx);
            return status[d];
        }
    return DCS_OK;
}

#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of dcs_decode_batch: host buffers in, host PCM out, per call; and the
cost of the index pass in front of it: one host thread, all host threads (dcs_index_streams), and the GPU
index kernel (dcs_index_streams_gpu: call = upload + kernel + download; kernel = HIP events)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads


def best(fn, n=3):
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return min(t)


ctx = D.Context(0)
for wl in ("dcs93_4096", "dcs94_65536", "mixed_16384"):
    b = workloads.build(wl)
    streams = b["streams"]
    ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    dt = (time.perf_counter() - t0) / n
    frames = b["jobs"].size
    t1 = best(lambda: D.index_streams(streams, threads=1))
    tN = best(lambda: D.index_streams(streams, threads=0))
    tG = best(lambda: ctx.index_streams_gpu(streams))
    kG = ctx.index_gpu_time(5)
    print("%s: %d streams, %d frames: decode host->host %.3f ms = %.3e samples/s ; index pass: 1 host thread %.2f ms, "
          "all host threads (%d) %.2f ms, GPU call %.2f ms (kernel %.3f ms)"
          % (wl, len(streams), frames, dt * 1e3, frames * 240 / dt, t1 * 1e3, os.cpu_count(), tN * 1e3, tG * 1e3, kG))

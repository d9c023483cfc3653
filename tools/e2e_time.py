#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of dcs_decode_batch: host buffers in, host PCM out, per call."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
ctx = D.Context(0)
for wl in ("dcs93_4096", "dcs94_65536"):
    b = workloads.build(wl)
    t0 = time.perf_counter(); b2 = D.build_stream_batch(b["streams"]); t_index = time.perf_counter() - t0
    ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    dt = (time.perf_counter() - t0) / n
    frames = b["jobs"].size
    print("%s: host->host %.3f ms per batch = %.3e samples/s ; host index pass + descriptors (python+C, 1 thread) %.1f ms"
          % (wl, dt * 1e3, frames * 240 / dt, t_index * 1e3))

#!/usr/bin/env python3
"""The device index pass per unpack layout: 256 streams x 256 frames of one layout (a lone wavefront per SIMD: the walk's latency)
and 8 192 (eight per SIMD: its throughput).  Shows what a frame of each layout costs the walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads

NAMES = {D.FMT_93_T0: "93-T0", D.FMT_93B_T1: "93b-T1", D.FMT_93A_T1: "93a-T1", D.FMT_94_T0: "94-T0", D.FMT_94_T1_S0: "94-T1s0", D.FMT_94_T1_S3: "94-T1s3"}
ctx = D.Context(0)
for fmt in sorted(NAMES):
    base = workloads.streams_one_layout(fmt, 256, 256)
    nbytes = sum(len(s[1]) if isinstance(s, tuple) else len(s) for s in base) / (256.0 * 256)
    line = "%-8s %5.0f B/frame:" % (NAMES[fmt], nbytes)
    for mult in (1, 32):
        streams = base * mult
        ctx.index_streams_gpu(streams)
        ms = ctx.index_gpu_time(3)
        line += "  %5d streams %6.2f ms (%6.2f us/frame/stream, %5.2f ns/frame)" % (len(streams), ms, ms * 1e3 / 256, ms * 1e6 / (len(streams) * 256))
    print(line)

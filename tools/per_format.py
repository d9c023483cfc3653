#!/usr/bin/env python3
"""Kernel time per frame of each unpack layout on its own: 65 536 frames (512 seeded streams of 128 frames) per layout,
at every frames-per-wavefront setting.  Shows which layout a mixed list pays most for."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads

NAMES = {D.FMT_93_T0: "93-T0", D.FMT_93B_T1: "93b-T1", D.FMT_93A_T1: "93a-T1", D.FMT_94_T0: "94-T0", D.FMT_94_T1_S0: "94-T1s0", D.FMT_94_T1_S3: "94-T1s3"}
ctx = D.Context(0)
n_streams, n_frames = 512, 128
for fmt in sorted(NAMES):
    streams = workloads.streams_one_layout(fmt, n_streams, n_frames)
    b = D.build_stream_batch(streams)
    line = "%-8s %6d frames, %5.0f bits/frame:" % (NAMES[fmt], b["jobs"].size, float(b["srcs"]["idx"]["nBits"].mean()))
    for fpw in (4, 8, 16):
        ctx.set_frames_per_wave(fpw)
        bt = ctx.batch(b["blob"], b["srcs"], b["jobs"])
        bt.time(10)
        ms = min(bt.time(50) for _ in range(3))
        line += "  fpw%-2d %7.2f us (%.3f ns/frame)" % (fpw, ms * 1e3, ms * 1e6 / b["jobs"].size)
        bt.close()
    print(line)

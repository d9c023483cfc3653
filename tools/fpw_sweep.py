#!/usr/bin/env python3
"""Kernel time (HIP events) of each workload at every frames-per-wavefront setting, and of sub-sampled
batch sizes, to tune chooseFpw in dcs_runtime.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads

ctx = D.Context(0)
for wl in ("dcs93_4096", "mixed_16384", "dcs94_65536"):
    full = workloads.build(wl)
    streams = full["streams"]
    for frac in (1, 2, 4, 8, 16):
        sub = streams[:max(1, len(streams) // frac)]
        b = D.build_stream_batch(sub)
        line = "%-12s %6d frames:" % (wl, b["jobs"].size)
        for fpw in (4, 8, 16):
            ctx.set_frames_per_wave(fpw)
            bt = ctx.batch(b["blob"], b["srcs"], b["jobs"])
            bt.time(10)
            ms = min(bt.time(50) for _ in range(3))
            line += "  fpw%-2d %7.2f us" % (fpw, ms * 1e3)
            bt.close()
        print(line)

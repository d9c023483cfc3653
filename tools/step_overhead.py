#!/usr/bin/env python3
"""What the wall clock around K back-to-back launches adds to K kernel durations (bench.py's timed region against its
event-timed kernel average) in a FRESH process whose GPU has been idle, by how long the clock-probe kernel is run first
(argv[1], milliseconds): the shader clock needs tens of milliseconds of load to settle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
settle_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0
k, w = 20, 5
ctx = D.Context(0)
b = workloads.build("dcs94_65536")
bt = ctx.batch(b["blob"], b["srcs"], b["jobs"])
stream = torch.cuda.current_stream().cuda_stream
time.sleep(1.0)
t0 = time.perf_counter(); n = 0; mhz = []
while (time.perf_counter() - t0) * 1e3 < settle_ms:
    mhz.append(ctx.clock_mhz()); n += 1
for _ in range(w):
    bt.run(stream)
torch.cuda.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
bt.run_many(k, stream)
torch.cuda.synchronize(); torch.cuda.synchronize()
us = (time.perf_counter() - t0) / k * 1e6
kern = sorted(bt.time(k, stream) for _ in range(3))[1] * 1e3
print("settle %5.0f ms (%3d probes, clock %s): wall per step %.2f us, event per step afterwards %.2f us" %
      (settle_ms, n, " ".join("%.0f" % m for m in (mhz[:2] + mhz[-2:])), us, kern))

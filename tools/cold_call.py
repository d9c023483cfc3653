#!/usr/bin/env python3
"""One synchronous dcs_decode_streams call per list (65 536 frames, pageable memory in and out): time per call.
DCS_PIPE_TRACE=1 prints the stages of every part."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = W.streams_dcs94_65536()
ctx = D.Context(0)
refs, keep = D.make_refs(streams)
n = 65536
pcm = np.zeros((n, 240), dtype=np.int16); first = np.zeros(257, dtype=np.uint32)
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    st = ctx.L.dcs_decode_streams(ctx.h, refs, 256, 0, pcm.ctypes.data_as(ctypes.c_void_p), n, first.ctypes.data_as(ctypes.c_void_p), None)
    ts.append((time.perf_counter() - t0) * 1e3)
    if calls <= 8:
        print("call %d: %.2f ms" % (i, ts[-1]), st)
ts = np.array(ts[2:])
print("%d calls: median %.2f ms, mean %.2f, min %.2f, 90%% %.2f" % (len(ts), np.median(ts), ts.mean(), ts.min(), np.percentile(ts, 90)))

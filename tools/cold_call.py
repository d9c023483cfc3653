import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
streams = W.streams_dcs94_65536()
ctx = D.Context(0)
refs, keep = D.make_refs(streams)
n = 65536
pcm = np.zeros((n, 240), dtype=np.int16); first = np.zeros(257, dtype=np.uint32)
for i in range(8):
    t0 = time.perf_counter()
    st = ctx.L.dcs_decode_streams(ctx.h, refs, 256, 0, pcm.ctypes.data_as(ctypes.c_void_p), n, first.ctypes.data_as(ctypes.c_void_p), None)
    print("call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), st)

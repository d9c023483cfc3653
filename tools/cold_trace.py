"""one synchronous dcs_decode_streams call on the 256 x 256 list [argv[1]: device | host]: the call's time and
(DCS_PIPE_TRACE=1) how long the index pass and the parts behind it took"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
streams = W.streams_dcs94_65536()
ctx = D.Context(0)
if len(sys.argv) > 1:
    ctx.set_large_list_path(sys.argv[1] == "device")      # (default: device)
refs, keep = D.make_refs(streams)
n = sum(((s[1][0] << 8) | s[1][1]) for s in streams)
pcm = np.zeros((n, 240), dtype=np.int16); first = np.zeros(len(streams) + 1, dtype=np.uint32)
def call():
    st = ctx.L.dcs_decode_streams(ctx.h, refs, len(streams), 0, pcm.ctypes.data_as(ctypes.c_void_p), n, first.ctypes.data_as(ctypes.c_void_p), None)
    assert st == 0
for _ in range(4): call()
ts = []
for _ in range(15):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
print("calls: median %.2f ms, min %.2f, max %.2f" % (sorted(ts)[len(ts) // 2], min(ts), max(ts)))

"""one synchronous dcs_decode_streams call on the default list (256 streams x 256 frames), traced: DCS_PIPE_TRACE=1 python tools/one_list_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, ctypes
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
streams = workloads.WORKLOADS["survey3_65536"]()
refs, keep = D.make_refs(streams)
ctx = D.Context(0)
if os.environ.get("ONE_LIST_MODE"):
    ctx.set_large_list_path(int(os.environ["ONE_LIST_MODE"]))
n_frames = sum((s[1][0] << 8) | s[1][1] for s in streams)
pcm = np.zeros((n_frames, 240), dtype=np.int16)
first = np.zeros(len(streams) + 1, dtype=np.uint32)
ts = []
for i in range(12):
    if i == 11:
        sys.stderr.write("==== last call\n")
    t0 = time.perf_counter()
    st = ctx.L.dcs_decode_streams(ctx.h, refs, len(streams), 0, pcm.ctypes.data_as(ctypes.c_void_p), n_frames, first.ctypes.data_as(ctypes.c_void_p), None)
    ts.append((time.perf_counter() - t0) * 1e3)
    assert st == 0
print("calls ms:", " ".join("%.2f" % t for t in ts))

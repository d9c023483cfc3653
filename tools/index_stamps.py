"""where the device index pass spends its cycles (diagnostic library: make -C dcsexplorer_amd/csrc variant NAME=idxstamps
DEFS=-DDCS_IDX_STAMPS; run with DCS_HIP_LIB=dcsexplorer_amd/libdcs_hip_idxstamps.so).  argv[1]: workload (dcs94, survey3, dcs93, mixed)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
ctx = D.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "dcs94"
streams = {"dcs94": W.streams_dcs94_65536, "survey3": W.streams_survey3_65536, "dcs93": W.streams_dcs93_4096, "mixed": W.streams_mixed_16384}[which]()
ctx.index_streams_gpu(streams)
out = (ctypes.c_ulonglong * 12)()
ctx.L.dcs_debug_index_stamps(out)
ctx.index_streams_gpu(streams)
ctx.L.dcs_debug_index_stamps(out)
names = ["walk", "header deltas", "huffRun", "  its chains", "record out", "slides", "runs", "symbols", "frames", "94: set-up", "94: band loop", "94: tail"]
fr = max(out[8], 1)
print("%s: %d streams, %d frames; kernel %.2f ms" % (which, len(streams), out[8], ctx.index_gpu_time(3)))
for k in range(12):
    print("  %-14s %12d  %9.1f per frame" % (names[k], out[k], out[k] / fr))
if out[6]:
    print("  per run: %.0f cycles, %.1f symbols; per symbol of chain: %.1f cycles" % (out[2] / out[6], out[7] / out[6], out[3] / max(out[7], 1)))

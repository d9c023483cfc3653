"""what tools/prof_index.sh runs under rocprofv3: the device index pass over <lists> x the workload's list of 256 streams x 256
frames (argv[1] lists, argv[2] workload; list r is the share of rank r, so the streams differ from list to list)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import sharding, workloads as W
W.register_recordings(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "encoder_golden.npz")))
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = sys.argv[2] if len(sys.argv) > 2 else "survey3_65536"
ctx = D.Context(0)
streams = [s for r in range(mult) for s in sharding.rank_streams(wl, r)]
ctx.index_streams_gpu(streams)
frames = sum((s[1][0] << 8) | s[1][1] for s in streams)
print("%s: %d streams, %d frames: index kernel %.3f ms" % (wl, len(streams), frames, ctx.index_gpu_time(10)))

"""what tools/prof_index.sh runs under rocprofv3: the device index pass over a list of 256 streams x 256 frames (x argv[1] lists)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = D.Context(0)
streams = W.streams_dcs94_65536() * mult
ctx.index_streams_gpu(streams)
print("%d streams x 256 frames: index kernel %.3f ms" % (len(streams), ctx.index_gpu_time(10)))

"""the device index pass alone: kernel time by number of streams (256 frames each)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
ctx = D.Context(0)
base = W.streams_dcs94_65536()
import sys
for mult in ((1,) if len(sys.argv) > 1 else (1, 4, 8, 16, 24, 32)):
    streams = base * mult
    ctx.index_streams_gpu(streams[:len(streams)])
    print("%5d streams x 256 frames: index kernel %.2f ms" % (len(streams), ctx.index_gpu_time(10 if len(sys.argv) > 1 else 3)))

"""the device index pass of one 256 x 256 list by shader clock: straight after an idle period, and after the clock probe has
held every SIMD busy for a while (bench.py settles the clock the same way before its timed steps)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
ctx = D.Context(0)
streams = W.streams_dcs94_65536()
ctx.index_streams_gpu(streams)
time.sleep(0.5)
print("idle first: index kernel %.3f ms (1 launch), then %.3f ms (avg of 3)" % (ctx.index_gpu_time(1), ctx.index_gpu_time(3)))
for settle in (20, 50, 100):
    time.sleep(0.5)
    t0 = time.perf_counter(); mhz = 0.0
    while (time.perf_counter() - t0) * 1e3 < settle:
        mhz = ctx.clock_mhz()
    a = ctx.index_gpu_time(1); b = ctx.index_gpu_time(10); mhz2 = ctx.clock_mhz()
    print("settled %3d ms (probe %.0f MHz): %.3f ms (1 launch), %.3f ms (avg of 10); probe after %.0f MHz" % (settle, mhz, a, b, mhz2))

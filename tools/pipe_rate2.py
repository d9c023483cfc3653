"""two pipelines one after the other on one context (is the slow start of a pipeline the context's buffer cache, or the new streams?)
and the rate per block of 50 lists; argv[1] depth, argv[2] lists"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
depth = int(sys.argv[1]); n = int(sys.argv[2])
streams = W.WORKLOADS["survey3_65536"]()
ctx = D.Context(0)
refs, keep = D.make_refs(streams)
for rnd in range(3):
    pipe = ctx.pipeline(depth, index_on_device=True, pack_on_device=True, plan_on_device=True)
    t0 = time.perf_counter(); done = 0; marks = [t0]
    for k in range(n):
        pipe.submit_refs(refs, len(streams))
        if k >= depth - 1:
            pipe.collect(); done += 1
            if done % 50 == 0: marks.append(time.perf_counter())
    while done < n:
        pipe.collect(); done += 1
        if done % 50 == 0: marks.append(time.perf_counter())
    print("pipeline %d (no warm-up): per 50 lists: %s" % (rnd, " ".join("%.3f" % ((b - a) * 20) for a, b in zip(marks, marks[1:]))), "cache", [x >> 20 for x in ctx.cache_bytes()])
    pipe.close()
ctx.close()

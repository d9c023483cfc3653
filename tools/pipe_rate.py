"""sustained rate of dcs_pipeline (index pass, planner and packer on the device) over a long run, per block of 100 lists:
argv[1] depth, argv[2] lists, [argv[3] workload]; environment: DCS_PIPE_ROUND_STREAMS, DCS_PIPE_WORKERS"""
import sys, time, os, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
depth = int(sys.argv[1]); n = int(sys.argv[2]); wl = sys.argv[3] if len(sys.argv) > 3 else "survey3_65536"
streams = W.corpus_streams(W.corpus_manifest(29, 20, 2000, 5)) if wl == "corpus" else W.WORKLOADS[wl]()
if os.environ.get("STREAMS"):                       # e.g. STREAMS=0:150: a slice of the workload's streams as the list
    a, b = os.environ["STREAMS"].split(":"); streams = streams[int(a):int(b)]
if os.environ.get("BIND"):
    print("bound to NUMA node", D.bind_process_to_device_numa(0))
ctx = D.Context(0)
print("link %.1f GB/s" % ctx.link_rate())
refs, keep = D.make_refs(streams)
pipe = ctx.pipeline(depth, index_on_device=True, pack_on_device=True, plan_on_device=True)
for _ in range(int(os.environ.get("WARM_ROUNDS", "1"))):
    for _ in range(depth): pipe.submit_refs(refs, len(streams))
    for _ in range(depth): pipe.collect()
r0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.perf_counter(); done = 0; marks = [t0]; other = 0
for k in range(n):
    pipe.submit_refs(refs, len(streams))
    if k >= depth - 1:
        pipe.collect(); done += 1; other += pipe.last_path != 7
        if done % 100 == 0: marks.append(time.perf_counter())
while done < n:
    pipe.collect(); done += 1; other += pipe.last_path != 7
    if done % 100 == 0: marks.append(time.perf_counter())
dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF)
print("depth %d lists %d (%d not wholly on the device) round_streams %s workers %s: %.3f ms/list overall, cpu %.2f ms/list; per 100 lists: %s" % (
    depth, n, other, os.environ.get("DCS_PIPE_ROUND_STREAMS", "-"), os.environ.get("DCS_PIPE_WORKERS", "-"), dt / n * 1e3,
    ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) / n * 1e3, " ".join("%.3f" % ((b - a) * 10) for a, b in zip(marks, marks[1:]))))
pipe.close(); ctx.close()

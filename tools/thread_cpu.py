"""CPU time per thread (by name) of this process over a timed section: where the host CPU-milliseconds per list go.
usage on the GPU box:  python tools/thread_cpu.py [node2|pipe] [lists]      (node2: dcs_node over [0, 0]; pipe: one dcs_pipeline)"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads

TICK = os.sysconf("SC_CLK_TCK")

def snapshot():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
        except OSError:
            continue
        name = st[st.index("(") + 1:st.rindex(")")]
        f = st[st.rindex(")") + 2:].split()
        out[int(tid)] = (name, (int(f[11]) + int(f[12])) / TICK)      # utime + stime, seconds
    return out

mode = sys.argv[1] if len(sys.argv) > 1 else "node2"
n_lists = int(sys.argv[2]) if len(sys.argv) > 2 else 400
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 24
streams = workloads.WORKLOADS["survey3_65536"]()
refs, keep = D.make_refs(streams)
class Two:
    """two contexts with a pipeline each, lists dealt alternately (what dcs_node does, without dcs_node)"""
    def __init__(self, depth):
        self.ctxs = [D.Context(0), D.Context(0)]
        self.pipes = [c.pipeline(depth, index_on_device=True, pack_on_device=True, plan_on_device=True) for c in self.ctxs]
        self.i = self.o = 0
    def submit(self):
        self.pipes[self.i & 1].submit_refs(refs, len(streams)); self.i += 1
    def collect(self):
        r = self.pipes[self.o & 1].collect(); self.o += 1; return r
    def close(self):
        for p in self.pipes: p.close()
        for c in self.ctxs: c.close()
if mode == "node2busy":
    # a context whose own pipeline exists (one synchronous call on a large list made it: 14 more streams, idle from then on),
    # as in a bench.py process that has run its other sections before the node's
    import ctypes
    other = D.Context(0)
    n_frames = sum((s[1][0] << 8) | s[1][1] for s in streams)
    pcm = np.zeros((n_frames, 240), dtype=np.int16); first = np.zeros(len(streams) + 1, dtype=np.uint32)
    for _ in range(3):
        other.L.dcs_decode_streams(other.h, refs, len(streams), 0, pcm.ctypes.data_as(ctypes.c_void_p), n_frames, first.ctypes.data_as(ctypes.c_void_p), None)
    mode = "node2"
if mode == "node2":
    obj = D.Node([0, 0], depth=depth); inflight = 2 * depth
    submit, collect = (lambda: obj.submit_refs(refs, len(streams))), obj.collect
elif mode == "node1":
    obj = D.Node([0], depth=2 * depth); inflight = 2 * depth
    submit, collect = (lambda: obj.submit_refs(refs, len(streams))), obj.collect
elif mode == "pipe2":
    obj = Two(depth); inflight = 2 * depth
    submit, collect = obj.submit, obj.collect
else:
    ctx = D.Context(0)
    obj = ctx.pipeline(2 * depth, index_on_device=True, pack_on_device=True, plan_on_device=True); inflight = 2 * depth
    submit, collect = (lambda: obj.submit_refs(refs, len(streams))), obj.collect
for _ in range(2):
    for _ in range(inflight): submit()
    for _ in range(inflight): collect()
s0 = snapshot(); t0 = time.perf_counter()
done = 0
for k in range(n_lists):
    submit()
    if k >= inflight - 1:
        collect(); done += 1
while done < n_lists:
    collect(); done += 1
dt = time.perf_counter() - t0; s1 = snapshot()
by = collections.defaultdict(lambda: [0, 0.0])
for tid, (name, cpu) in s1.items():
    d = cpu - s0.get(tid, (name, 0.0))[1]
    by[name][0] += 1; by[name][1] += d
total = sum(v[1] for v in by.values())
print("%s: %d lists, %.3f ms per list, %.2f CPU-ms per list in all threads" % (mode, n_lists, dt / n_lists * 1e3, total / n_lists * 1e3))
for name, (n, cpu) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-18s x%-3d %.3f CPU-ms per list" % (name, n, cpu / n_lists * 1e3))
main = os.getpid()
top = sorted(((cpu - s0.get(tid, (name, 0.0))[1], tid, name) for tid, (name, cpu) in s1.items()), reverse=True)[:8]
print("  busiest threads: " + ", ".join("%s%s %.3f" % (name, " (main)" if tid == main else "", d / n_lists * 1e3) for d, tid, name in top))
obj.close()

import sys, time, os, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
streams = W.streams_dcs94_65536()
ctx = D.Context(0)
refs, keep = D.make_refs(streams)
depth = int(sys.argv[1]); dev = sys.argv[2] in ("dev", "devpack", "devplan")
pipe = ctx.pipeline(depth, index_on_device=dev, pack_on_device=sys.argv[2] in ("devpack", "devplan"), plan_on_device=sys.argv[2] == "devplan")
# warm-up: the context's buffer cache takes about three rounds of `depth` lists until nothing is allocated any more
for _ in range(3):
    for _ in range(depth): pipe.submit_refs(refs, len(streams))
    for _ in range(depth): pipe.collect()
n = 3 * depth
def cpustat():
    try:
        return dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat').read().splitlines())
    except Exception:
        return {}
def threadstat():
    import glob, collections
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    tck = os.sysconf("SC_CLK_TCK")
    for t in glob.glob("/proc/self/task/*"):
        try:
            comm = open(t + "/comm").read().strip()
            f = open(t + "/stat").read().rsplit(")", 1)[1].split()
            agg[comm][0] += 1; agg[comm][1] += int(f[11]) / tck; agg[comm][2] += int(f[12]) / tck
            tid = "tid " + t.rsplit("/", 1)[1] + " (" + comm + ")"
            agg[tid][0] += 1; agg[tid][1] += int(f[11]) / tck; agg[tid][2] += int(f[12]) / tck
        except Exception:
            pass
    return agg
th0 = threadstat()
r0 = resource.getrusage(resource.RUSAGE_SELF); c0 = cpustat()
t0 = time.perf_counter(); done = 0
for k in range(n):
    pipe.submit_refs(refs, len(streams))
    if k >= depth - 1: pipe.collect(); done += 1
while done < n: pipe.collect(); done += 1
dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF); c1 = cpustat()
print("ms/list", dt / n * 1e3, "cpu user ms/list", (r1.ru_utime - r0.ru_utime) / n * 1e3, "sys ms/list", (r1.ru_stime - r0.ru_stime) / n * 1e3,
      "minflt/list", (r1.ru_minflt - r0.ru_minflt) / n, "nvcsw/list", (r1.ru_nvcsw - r0.ru_nvcsw) / n,
      "throttled_usec/list", (int(c1.get("throttled_usec", 0)) - int(c0.get("throttled_usec", 0))) / n, "nr_throttled", int(c1.get("nr_throttled", 0)) - int(c0.get("nr_throttled", 0)))


# CPU time by thread name over the measured lists only
th1 = threadstat()
for k in sorted(th1, key=lambda k: -(th1[k][1] - th0.get(k, [0, 0, 0])[1])):
    u = th1[k][1] - th0.get(k, [0, 0.0, 0.0])[1]; sy = th1[k][2] - th0.get(k, [0, 0.0, 0.0])[2]
    if (u + sy) / n * 1e3 >= 0.05:
        sys.stderr.write("  threads %-18s n=%3d user %.2f ms/list sys %.2f ms/list\n" % (k, th1[k][0], u / n * 1e3, sy / n * 1e3))

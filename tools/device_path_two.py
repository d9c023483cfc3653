"""the whole device path (dcs_device_path: index walk, planner, packer, decode; nothing over PCIe) of TWO resident lists at once, each
on a context (HIP stream) of its own and driven by a thread of its own: does one list's scalar-bound index walk overlap the other's
VALU-bound decode?  argv[1]: lists of 256 streams per path (default 16)"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import dcsexplorer_amd as D
from dcsexplorer_amd import sharding
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 16
npaths = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctxs = [D.Context(0) for _ in range(npaths)]
paths = []
for i, c in enumerate(ctxs):
    streams = [s for r in range(mult) for s in sharding.rank_streams("survey3_65536", (i * mult + r) % 32)]
    paths.append(c.device_path(streams))
frames = paths[0].n_frames
t = paths[0].run(6)
print("one path alone: %d frames, %.3f ms per pass (index %.3f, plan %.3f, pack %.3f, decode %.3f) -> %.3g samples/s" % (
    frames, t["passMs"], t["indexMs"], t["planMs"], t["packMs"], t["decodeMs"], frames * 240 / (t["passMs"] * 1e-3)))
res = [None] * npaths
def run(i):
    res[i] = paths[i].run(12)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i,)) for i in range(npaths)]
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t0
worst = max(r["passMs"] for r in res)
print("%d paths at once: %s ms per pass each; wall %.1f ms -> about %.3g samples/s together (%.2f ns per frame)" % (
    npaths, " / ".join("%.3f" % r["passMs"] for r in res), dt * 1e3, npaths * frames * 240 / (worst * 1e-3), worst * 1e6 / (npaths * frames)))

#!/usr/bin/env python3
"""Timeline of one launch: when every wavefront started and ended on the clock all compute units share (100 MHz), from
the diagnostic library built with `make stamps DEFS=-DDCS_STAMPS_REALTIME`.  Prints how many wavefronts were resident
over time, which shows the ramp at the start of the launch, the turn-over between workgroups and the tail."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcsexplorer_amd.api as api
api.lib_path = lambda: os.environ.get("DCS_STAMPS_LIB", os.path.join(ROOT, "dcsexplorer_amd", "libdcs_hip_stamps.so"))
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "dcs94_65536"
fpw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = D.Context(0)
if fpw: ctx.set_frames_per_wave(fpw)
b = workloads.build(wl)
bt = ctx.batch(b["blob"], b["srcs"], b["jobs"])
for _ in range(3): bt.run()
bt.sync()
ms = bt.time(20)
L = D.load_library()
cap = 1 << 17
out = np.zeros((cap, 16), dtype=np.uint64)
L.dcs_debug_stamps.restype = ctypes.c_int
n = L.dcs_debug_stamps(bt.h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cap))
st = out[:n].astype(np.int64)
t0 = st[:, 0].min()
start, pkg, unp, end = (st[:, 0] - t0) * 0.01, (st[:, 1] - t0) * 0.01, (st[:, 4] - t0) * 0.01, (st[:, 6] - t0) * 0.01      # microseconds
print(wl, "chunks", n, "kernel (events, this build) %.1f us; first start -> last end %.1f us" % (ms * 1e3, end.max()))
print("wavefront life: median %.1f us, 10%% %.1f, 90%% %.1f, max %.1f" % (np.median(end - start), *np.percentile(end - start, [10, 90]), (end - start).max()))
print("  of which until the package is there: median %.1f us (first 4096: %.1f, rest: %.1f)" %
      (np.median(pkg - start), np.median((pkg - start)[np.argsort(start)[:4096]]), np.median((pkg - start)[np.argsort(start)[4096:]]) if n > 4096 else 0))
edges = np.arange(0, end.max() + 1, 1.0)
print("  t(us)  resident  started  ended")
for a in edges:
    res = int(((start <= a) & (end > a)).sum())
    print("  %5.0f  %8d  %7d  %5d" % (a, res, int(((start >= a) & (start < a + 1)).sum()), int(((end >= a) & (end < a + 1)).sum())))
# wavefront life by stream (the seeded workloads write their streams one after the other, n_frames each)
try:
    plan = D.plan_chunks(b["jobs"], bt.frames_per_wave, b["srcs"], handoff=True)
    if plan.shape[0] == n:
        nf = b["jobs"].size // len(b["streams"])
        stream = (plan[:, 0]["job"].astype(np.int64) // nf)
        life = end - start
        per = np.array([life[stream == k].mean() for k in range(len(b["streams"]))])
        order = np.argsort(per)
        print("per-stream mean wavefront life: min %.1f us (stream %d), median %.1f, max %.1f (stream %d)" %
              (per.min(), order[0], np.median(per), per.max(), order[-1]))
        print("  ten slowest streams:", [(int(k), round(float(per[k]), 1)) for k in order[-10:]])
        print("  ten fastest streams:", [(int(k), round(float(per[k]), 1)) for k in order[:10]])
        bits = np.array([b["srcs"]["idx"]["nBits"][stream_k * nf:(stream_k + 1) * nf].mean() for stream_k in range(len(b["streams"]))])
        print("  correlation of a stream's mean life with its mean frame length in bits: %.2f" % np.corrcoef(per, bits)[0, 1])
        lvl = np.array([life[plan[:, 0]["job"].astype(np.int64) % nf // bt.frames_per_wave == d].mean() for d in range(nf // bt.frames_per_wave)])
        print("  mean life by position in the stream (chunk 0, 1, ...):", np.round(lvl, 1).tolist())
except Exception as e:
    print("(no per-stream breakdown: %s)" % e)

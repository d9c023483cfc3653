#!/usr/bin/env python3
"""Which chunks are the slow ones: percentiles of the per-chunk cycle counts and the phase breakdown of the slowest
5 % next to the median chunk (diagnostic library, see tools/stamps.py)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcsexplorer_amd.api as api
api.lib_path = lambda: os.path.join(ROOT, "dcsexplorer_amd", "libdcs_hip_stamps.so")
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed_16384"
ctx = D.Context(0)
b = workloads.build(wl)
bt = ctx.batch(b["blob"], b["srcs"], b["jobs"])
for _ in range(3): bt.run()
bt.sync()
L = D.load_library()
cap = 1 << 16
out = np.zeros((cap, 16), dtype=np.uint64)
L.dcs_debug_stamps.restype = ctypes.c_int
n = L.dcs_debug_stamps(bt.h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(cap))
st = out[:n].astype(np.int64)
tot = st[:, 6] - st[:, 0]
t0 = st[:, 0].min()
print(wl, "chunks", n, "percentiles of chunk cycles 5/50/95/100:", [int(np.percentile(tot, p)) for p in (5, 50, 95, 100)])
print("  start skew (first stamp - earliest): median %d max %d ; end of last chunk %d" % (np.median(st[:, 0] - t0), (st[:, 0] - t0).max(), (st[:, 6] - t0).max()))
slow = tot >= np.percentile(tot, 95)
names = ["tables+barrier", "staging", "hdr", "unpack", "err/dc", "(sync)", "transform+emit"]
for k in range(6):
    d = st[:, k + 1] - st[:, k]
    print("  %-16s median %7d   slowest-5%% median %7d" % (names[k] if k < 5 else names[6], np.median(d), np.median(d[slow])))
for a_, b_, name in ((5, 13, "transform passes"), (13, 6, "imported tails"), (0, 1, "start -> after barrier")):
    ok = (st[:, a_] != 0) & (st[:, b_] != 0)
    d = st[:, b_] - st[:, a_]
    print("  %-24s median %7d   slowest-5%% median %7d" % (name, np.median(d[ok]), np.median(d[ok & slow])))
order = np.argsort(st[:, 0])
print("  chunk start times (cycles after the first): p50 %d p95 %d max %d ; end times: p50 %d p95 %d max %d" % (
    np.percentile(st[:, 0] - st[:, 0].min(), 50), np.percentile(st[:, 0] - st[:, 0].min(), 95), (st[:, 0] - st[:, 0].min()).max(),
    np.percentile(st[:, 6] - st[:, 0].min(), 50), np.percentile(st[:, 6] - st[:, 0].min(), 95), (st[:, 6] - st[:, 0].min()).max()))
# formats of the chunks (first slot's first source), slow vs all
try:
    fpw = 4 if b["jobs"].size <= 1024 * 16 else 8 if b["jobs"].size <= 1024 * 192 else 16
    plan = D.plan_chunks(b["jobs"], fpw, b["srcs"])
    fmt = b["srcs"]["format"][b["jobs"]["firstSrc"][plan[:n, 0]["job"]]]
    d = st[:, 4] - st[:, 3]
    for f in np.unique(fmt):
        m = fmt == f
        print("  format %d: %5d chunks, unpack median %6d, total median %6d, share of slowest 5%%: %.2f" % (f, m.sum(), np.median(d[m]), np.median(tot[m]), (m & slow).sum() / max(1, slow.sum())))
except Exception as e:
    print("format breakdown failed:", e)

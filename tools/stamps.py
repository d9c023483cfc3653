#!/usr/bin/env python3
"""Where a wavefront spends its cycles: runs a workload on the DIAGNOSTIC library (make stamps) and prints
the median cycle count of each phase.  Shares only, not absolute times (the stamps serialise)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcsexplorer_amd.api as api
api.lib_path = lambda: os.path.join(ROOT, "dcsexplorer_amd", "libdcs_hip_stamps.so")
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "dcs93_4096"
fpw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = D.Context(0)
if fpw: ctx.set_frames_per_wave(fpw)
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
names = ["tables+zero+barrier", "lane consts + slot/desc + prefix + stage loads", "header loads", "unpack", "err/dc", "(sync)", "transform+emit"]
tot = st[:, 6] - st[:, 0]
print(wl, "fpw", fpw or "auto", "chunks", n, "median total cycles", int(np.median(tot)), "min", int(tot.min()), "max", int(tot.max()))
for k in range(6):
    d = st[:, k + 1] - st[:, k]
    print("  %-48s median %7d  (%.1f%%)" % (names[k] if k < 5 else names[6], int(np.median(d)), 100 * np.median(d) / np.median(tot)))
# finer stamps (7 is unused): 12 bit readers ready, 8 unpacker entered, 9 first band set up, 10 its symbol loop done,
# 11 unpacker's band loop done, 13 transform passes done (then: frames that import their tail)
fine = [(3, 12, "Q set-up + bit reader init"), (12, 8, "-> unpacker entry"), (8, 9, "first band set-up"),
        (9, 10, "first band symbol loop"), (10, 11, "rest of the bands"), (11, 4, "DC fix-up + sync"),
        (5, 7, "lane constants arrive"), (7, 14, "first pass: transform"), (14, 15, "first pass: tails, export, sync"),
        (5, 13, "transform passes"), (13, 6, "imported tails")]
for a_, b_, name in fine:
    ok = (st[:, a_] != 0) & (st[:, b_] != 0)
    if ok.any():
        d = st[ok, b_] - st[ok, a_]
        print("    %-44s median %7d" % (name, int(np.median(d))))

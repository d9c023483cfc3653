#!/usr/bin/env python3
"""Cycle stamps INSIDE the 1993 transform of a chunk's first pass (library built with
make variant NAME=xstamps DEFS="-DDCS_STAMPS -DDCS_STAMPS_XFORM"); shares only."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcsexplorer_amd.api as api
api.lib_path = lambda: os.path.join(ROOT, "dcsexplorer_amd", "libdcs_hip_xstamps.so")
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "dcs93_4096"
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
rows = [(7, 8, "pass set-up + expand"), (8, 9, "stages A (3)"), (9, 10, "transpose"), (10, 11, "stages B (4)"), (11, 14, "volume shift"),
        (14, 15, "tails, export, sync"), (15, 13, "overlap + PCM stores"), (13, 6, "imported tails"), (0, 6, "whole chunk")]
for a_, b_, name in rows:
    d = st[:, b_] - st[:, a_]
    print("  %-28s median %7d  min %7d  max %7d" % (name, int(np.median(d)), int(d.min()), int(d.max())))

#!/usr/bin/env python3
"""Where the host->host time of dcs_decode_batch goes: batch_create (validate + plan + malloc + H2D), run, download (D2H)."""
import sys, time, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
ctx = D.Context(0)
for wl in ("dcs93_4096", "dcs94_65536"):
    b = workloads.build(wl)
    for rep in range(3):
        t0 = time.perf_counter()
        bt = ctx.batch(b["blob"], b["srcs"], b["jobs"]); bt.sync()
        t1 = time.perf_counter()
        bt.run(); bt.sync()
        t2 = time.perf_counter()
        pcm, err = bt.download()
        t3 = time.perf_counter()
        v0 = time.perf_counter(); pv, ev = bt.download_view(); tv = time.perf_counter() - v0
        assert np.array_equal(pv, pcm)
        bt.close()
        t4 = time.perf_counter()
    plan_t0 = time.perf_counter(); D.plan_chunks(b["jobs"], 16, b["srcs"]); plan_t = time.perf_counter() - plan_t0
    print("%s: create %.3f ms (of which chunk plan ~%.3f ms) | run %.3f | download %.3f (pinned view %.3f) | destroy %.3f" %
          (wl, (t1 - t0) * 1e3, plan_t * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, tv * 1e3, (t4 - t3 - tv) * 1e3))

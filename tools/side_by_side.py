"""three contexts on one GPU, each relaunching a resident 65 536-frame batch from a thread of its own, with and without
dcs_ctx_set_concurrent_batches (chain order + XCD ranges): wall time, and frames flagged DCS_FRAME_TAIL_LOST"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
streams = workloads.streams_survey3_65536()
b = D.build_stream_batch(streams, indexer=D.index_streams)
for ranges in (False, True):
    ctxs = [D.Context(0) for _ in range(3)]
    batches = []
    for c in ctxs:
        c.set_concurrent_batches(ranges)
        batches.append(c.batch(b["blob"], b["srcs"], b["jobs"]))
    lost = [0] * len(batches)
    def drive(i):
        for _ in range(10):
            batches[i].run_many(100); batches[i].sync()
            pcm, err = batches[i].download()
            lost[i] += int(((err & D.FRAME_TAIL_LOST) != 0).sum())
    t0 = time.perf_counter()
    th = [threading.Thread(target=drive, args=(i,)) for i in range(len(batches))]
    for t in th: t.start()
    for t in th: t.join()
    print("xcd ranges %s: three contexts x 1000 launches side by side in %.2f s, frames flagged TAIL_LOST per context: %s" % (ranges, time.perf_counter() - t0, lost))
    for bt in batches: bt.close()
    for c in ctxs: c.close()

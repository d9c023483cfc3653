"""what slows the decode kernel when a pipeline is busy: its time alone, next to copies down (dcs_ctx_link_rate on another context),
next to the index kernel (another context), next to both"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
a, b, c = D.Context(0), D.Context(0), D.Context(0)
streams = W.streams_survey3_65536()
bb = D.build_stream_batch(streams, indexer=D.index_streams)
batch = a.batch(bb["blob"], bb["srcs"], bb["jobs"])
c.index_streams_gpu(streams * 8)
print("decode alone: %.1f us" % (batch.time(200) * 1e3))
def load(copy, walk):
    stop = [False]
    def copies():
        while not stop[0]: b.link_rate()
    def walks():
        while not stop[0]: c.index_gpu_time(3)
    th = ([threading.Thread(target=copies)] if copy else []) + ([threading.Thread(target=walks)] if walk else [])
    for t in th: t.start()
    time.sleep(0.1)
    r = [batch.time(300) * 1e3 for _ in range(3)]
    stop[0] = True
    for t in th: t.join()
    return r
print("next to copies down:        %s us" % " ".join("%.1f" % x for x in load(True, False)))
print("next to the index kernel:   %s us" % " ".join("%.1f" % x for x in load(False, True)))
print("next to both:               %s us" % " ".join("%.1f" % x for x in load(True, True)))

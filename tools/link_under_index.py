"""does the device index pass running next to it slow the PCM's way down?  dcs_ctx_link_rate (device -> pinned host, 64 MB copies)
alone, and while another context's index kernel walks argv[1] x 256 streams in a loop"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a, b = D.Context(0), D.Context(0)
streams = W.streams_survey3_65536() * mult
a.index_streams_gpu(streams)
print("alone: %.1f GB/s  %.1f GB/s" % (b.link_rate(), b.link_rate()))
stop = False
def walk():
    while not stop:
        a.index_gpu_time(5)
t = threading.Thread(target=walk); t.start()
time.sleep(0.05)
rates = [b.link_rate() for _ in range(6)]
stop = True; t.join()
print("next to the index kernel over %d streams: %s GB/s" % (len(streams), " ".join("%.1f" % r for r in rates)))
print("index kernel alone: %.2f ms" % a.index_gpu_time(5))

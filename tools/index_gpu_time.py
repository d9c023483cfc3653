"""the device index pass alone: kernel time by number of streams (256 frames each).  argv: [one] [survey3|dcs94|realistic]
(one: only the 256-stream list, ten launches; default workload: dcs94, the streams all of profiles/NOTES.md's figures are for)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W
args = sys.argv[1:]
one = "one" in args
which = [a for a in args if a != "one"]
which = which[0] if which else "dcs94"
if which == "realistic":
    W.register_recordings(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "encoder_golden.npz")))
if which == "corpus":
    # ragged lengths (20..2 000 frames, six layouts): 8 192 streams of the Config-5 stand-in in one launch
    m = W.corpus_manifest(titles=29, streams_per_title=283, max_frames=2000, seed=0x0005)
    streams = W.corpus_streams(m, 0, 8192)
    ctx = D.Context(0)
    ctx.index_streams_gpu(streams)
    print("corpus: %5d streams, %d frames: index kernel %.2f ms" % (len(streams), int(W.corpus_frames(m)[:8192].sum()), ctx.index_gpu_time(3)))
    sys.exit(0)
base = {"dcs94": W.streams_dcs94_65536, "survey3": W.streams_survey3_65536, "realistic": W.streams_realistic_65536}[which]()
ctx = D.Context(0)
for mult in ((1,) if one else (1, 4, 8, 16, 24, 32)):
    streams = base * mult
    ctx.index_streams_gpu(streams)
    print("%s: %5d streams x 256 frames: index kernel %.2f ms" % (which, len(streams), ctx.index_gpu_time(10 if one else 3)))

"""soak test of dcs_pipeline on a GPU box: many lists through every mode, PCM of every list hashed and compared, memory watched"""
import sys, os, time, resource, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dcsexplorer_amd as D
from dcsexplorer_amd import workloads as W

n_lists = int(sys.argv[1]) if len(sys.argv) > 1 else 600
base = W.streams_mixed_16384(n_streams=48, n_frames=96)           # all six layouts, 4 608 frames per list
variants = [base[i:] + base[:i] for i in (0, 7, 19)]              # three different lists
ctx = D.Context(0)
want = [zlib.crc32(ctx.decode_streams(v)[0].tobytes()) for v in variants]
refs = [D.make_refs(v) for v in variants]
for mode, name in ((0, "host index"), (1, "device index"), (2, "device index + pack"), (3, "device index + plan + pack")):
    pipe = ctx.pipeline(12, index_on_device=mode >= 1, pack_on_device=mode >= 2, plan_on_device=mode == 3)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.time(); bad = 0; done = 0
    for k in range(n_lists):
        r, keep = refs[k % 3]
        pipe.submit_refs(r, len(variants[k % 3]))
        if k >= 11:
            pcm, err, first, _, _ = pipe.collect()
            bad += zlib.crc32(pcm.tobytes()) != want[done % 3] or bool(err.any()); done += 1
    while done < n_lists:
        pcm, err, first, _, _ = pipe.collect()
        bad += zlib.crc32(pcm.tobytes()) != want[done % 3] or bool(err.any()); done += 1
    pipe.close()
    print("%-20s %d lists in %.1f s, %d wrong, max RSS grew %d MB" % (name, n_lists, time.time() - t0, bad,
          (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0) // 1024))
ctx.close()

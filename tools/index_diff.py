"""debug aid: device index pass against the host walk on the fuzz tool's lists; prints the first record field that differs.
argv[1]: number of seeds (default 200), argv[2]: first seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dcsexplorer_amd as D
from util import ALL_FORMATS, FORMAT_NAMES, make_stream, os_for, corrupt, splitmix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = D.Context(0)
bad = 0
for seed in range(seed0, seed0 + n):
    g = splitmix(0xF022 + seed)
    streams, fmts = [], []
    for k in range(24):
        fmt = ALL_FORMATS[next(g) % 6]
        nfr = 3 + next(g) % 70
        nb_max = 18 if fmt == D.FMT_93A_T1 else 16
        nbands = nb_max if next(g) % 3 else 1 + next(g) % nb_max
        stride_from = 16 if next(g) % 2 else next(g) % 16
        if fmt == D.FMT_93_T0 and stride_from < 16:
            nbands = min(nbands, 12)
        s = make_stream(fmt, nfr, seed=(seed << 8) + k, profile=next(g) % 4, stride_from=stride_from, nbands=nbands)
        r = next(g) % 8
        if r == 0 and len(s) > 24:
            s = corrupt(s, next(g) & 0xFFFF, nflips=1 + next(g) % 4) + bytes(256)
        elif r == 1 and len(s) > 40:
            s = s[:18 + (len(s) - 18) * (1 + next(g) % 3) // 4]
        os_ = os_for(fmt, next(g) & 1)
        next(g); next(g)
        streams.append((os_, s, 255, 0x64)); fmts.append(fmt)
    host = D.index_streams(streams)
    dev = ctx.index_streams_gpu(streams)
    for k, ((hr, hi), (dr, di)) in enumerate(zip(host, dev)):
        if hr.tobytes() == dr.tobytes() and bytes(hi) == bytes(di):
            continue
        bad += 1
        print("seed %d stream %d (%s): " % (seed, k, FORMAT_NAMES[fmts[k]]), end="")
        if bytes(hi) != bytes(di):
            for name, _ in hi._fields_:
                a, b = getattr(hi, name), getattr(di, name)
                if (bytes(a) if hasattr(a, "__len__") else a) != (bytes(b) if hasattr(b, "__len__") else b):
                    print("info.%s host %s device %s; " % (name, list(a) if hasattr(a, "__len__") else a, list(b) if hasattr(b, "__len__") else b), end="")
        if len(hr) != len(dr):
            print("records %d vs %d" % (len(hr), len(dr)))
        else:
            for f in range(len(hr)):
                if hr[f].tobytes() != dr[f].tobytes():
                    for name in hr.dtype.names:
                        if hr[f][name].tobytes() != dr[f][name].tobytes():
                            print("frame %d field %s host %s device %s" % (f, name, hr[f][name].tolist(), dr[f][name].tolist()))
                    break
            else:
                print()
        if bad >= 8:
            sys.exit(1)
print("index_diff: %d seeds, %d streams differ" % (n, bad))
sys.exit(1 if bad else 0)

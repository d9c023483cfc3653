#!/usr/bin/env python3
"""Generates tests/golden/rank_golden_hashes.json: per-stream FNV-1a-64 hashes of the PCM the UNMODIFIED reference
decoder (oracle/_ref/libdcsref.so) produces for the share of every rank 0..7 of the weak-scaling bench workloads
(rank r decodes streams [r*n, (r+1)*n) of the seeded corpus: dcsexplorer_amd/sharding.py rank_streams), so that
`bench.py --gpus N` can hold EVERY rank's PCM against the reference, not rank 0's only.  Each stream is played alone
from a fresh decoder, as --extract-streams plays them (DCSExplorer.cpp:1900-1907).  Build container only; the output is
data and travels to the GPU box."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dcsexplorer_amd import sharding, workloads              # noqa: E402
from oracle.dcs_oracle import Oracle, Reference              # noqa: E402

RANKS = 8
WORKLOADS = ("survey3_65536", "dcs94_65536", "dcs93_4096", "mixed_16384", "realistic_65536")


def main():
    workloads.register_recordings(np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz")))
    ref, orc = Reference(), Oracle()

    def one(s):
        os_, data, vol, lvl = s
        nf = (data[0] << 8) | data[1]
        return orc.fnv1a64(ref.decode(os_, vol, [data], [lvl], nf))

    out = {}
    with ThreadPoolExecutor(max_workers=8) as ex:
        for wl in WORKLOADS:
            per_rank = []
            for r in range(RANKS):
                streams = sharding.rank_streams(wl, r)
                per_rank.append(["%016x" % h for h in ex.map(one, streams)])
            out[wl] = dict(streams_per_rank=len(per_rank[0]), rank_stream_hashes=per_rank)
            print(wl, "done")
    with open(os.path.join(ROOT, "tests", "golden", "rank_golden_hashes.json"), "w") as f:
        json.dump(dict(ranks=RANKS, rule="rank r decodes streams [r*n, (r+1)*n) of the workload's seeded corpus", workloads=out), f)


if __name__ == "__main__":
    main()

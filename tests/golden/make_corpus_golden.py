#!/usr/bin/env python3
"""Generates tests/golden/corpus_golden.json: per-stream FNV-1a-64 hashes of the PCM the UNMODIFIED reference
decoder (oracle/_ref/libdcsref.so, built by oracle/Makefile from /root/reference) produces for the reduced
config-5 corpus (BASELINE.json configs[4] stand-in: 29 synthetic titles x 20 streams of U[20, 2000] frames, all six
unpack layouts; dcsexplorer_amd/workloads.py corpus_manifest).  Each stream is played alone from a fresh decoder
(LoadAudioStream(0, ptr, level) + nFrames x 240 GetNextSample), the way --extract-streams plays them
(DCSExplorer.cpp:1900-1907).  Runs only in the build container; the output is data and travels to the GPU box."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dcsexplorer_amd import workloads                        # noqa: E402
from oracle.dcs_oracle import Oracle, Reference              # noqa: E402

CORPUS = dict(titles=29, streams_per_title=20, max_frames=2000, seed=0x0005)
# SURVEY 8(d) Config 5 at its stated size: 29 titles x 600 streams (17.6 M frames); `--full` writes corpus_golden_full.json
CORPUS_FULL = dict(titles=29, streams_per_title=600, max_frames=2000, seed=0x0005)


def main():
    full = "--full" in sys.argv[1:]
    corpus = CORPUS_FULL if full else CORPUS
    ref, orc = Reference(), Oracle()
    manifest = workloads.corpus_manifest(**corpus)

    def one(m):
        # (a stream is written, played and dropped inside the worker: the full corpus is 2 GB of streams and 8 GB of PCM)
        os_, data, vol, lvl = workloads.corpus_streams([m])[0]
        nf = (data[0] << 8) | data[1]
        return orc.fnv1a64(ref.decode(os_, vol, [data], [lvl], nf))

    with ThreadPoolExecutor(max_workers=8) as ex:
        hashes = list(ex.map(one, manifest, chunksize=16))
    fmts = sorted(set(m["format"] for m in manifest))
    out = dict(corpus=corpus, streams=len(manifest), frames=int(sum(m["frames"] for m in manifest)), formats=fmts,
               fnv1a64_of_stream_hashes="%016x" % orc.fnv1a64(np.array(hashes, dtype=np.uint64)),
               stream_hashes=["%016x" % h for h in hashes])
    with open(os.path.join(ROOT, "tests", "golden", "corpus_golden_full.json" if full else "corpus_golden.json"), "w") as f:
        json.dump(out, f, indent=None if full else 1)
    print("corpus golden: %d streams, %d frames, layouts %s" % (out["streams"], out["frames"], fmts))


if __name__ == "__main__":
    main()

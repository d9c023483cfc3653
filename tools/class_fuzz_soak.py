#!/usr/bin/env python3
"""tests/cpp/dcs_class_fuzz.cpp for many seeds: DCSDecoderHIP behind the reference's real base class in lock step with the reference's
DCSDecoderNative, a seeded random caller (loads on any channel, ClearTracks, volume, IsStreamPlaying, pulls of 0..2 500 frames), every
sample compared.  argv: seeds per OS version [40], operations per seed [400].  Seeds whose calls kill the reference ALONE (it has
undefined behaviour on some damaged streams) run on the undamaged streams."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcsexplorer_amd as D
from util import make_stream, corrupt, splitmix

FUZZ = os.path.join(ROOT, "oracle", "_ref", "dcs_class_fuzz")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 400
t0 = time.time()
total_frames, runs, damaged, bad = 0, 0, 0, 0
with tempfile.TemporaryDirectory() as tmp:
    for os_ in (D.OS93A, D.OS93B, D.OS94, D.OS95):
        fmts = [f for f in range(6) if os_ in (D.format_os(f), D.format_os(f, prefer_95=True), D.format_os(f, prefer_93a=True))]
        g = splitmix(0x50AC + os_)
        paths, clean = [], []
        for k, n in enumerate([2, 9, 33, 64, 65, 130, 384, 385, 900, 2200, 60, 300, 800]):
            data = make_stream(fmts[next(g) % len(fmts)], n, seed=0x50AC0 + 32 * os_ + k, profile=(0, 1, 2, 3, 5)[next(g) % 5])
            p = os.path.join(tmp, "c%d_%d.bin" % (os_, k)); open(p, "wb").write(data); clean.append(p)
            if k >= 10:
                data = corrupt(data, seed=700 + k + 16 * os_)
            p = os.path.join(tmp, "f%d_%d.bin" % (os_, k)); open(p, "wb").write(data); paths.append(p)
        for seed in range(n_seeds):
            la = (-1, -1, -1, 0, 1, 5, 64, 333)[seed % 8]
            ops = n_ops if la != 1 else max(40, n_ops // 6)
            args = [str(os_), str(77000 + 1000 * os_ + seed), str(ops), str(la)]
            alone = subprocess.run([FUZZ] + args + paths, capture_output=True, text=True, env=dict(os.environ, DCS_FUZZ_REF_ONLY="1"))
            use = paths if alone.returncode == 0 else clean
            damaged += use is paths
            r = subprocess.run([FUZZ] + args + use, capture_output=True, text=True)
            runs += 1
            if r.returncode != 0 or not r.stdout.startswith("ok:"):
                bad += 1
                print("OS %d seed %d look-ahead %d: rc %d %s %s" % (os_, seed, la, r.returncode, r.stdout.strip()[-300:], r.stderr.strip()[-300:]))
            else:
                total_frames += int(r.stdout.split()[1])
print("%d runs (%d with damaged streams), %d frames compared sample by sample with the reference, %d different; %.0f s"
      % (runs, damaged, total_frames, bad, time.time() - t0))
print("all equal" if bad == 0 else "DIFFERENCES")
sys.exit(0 if bad == 0 else 1)

#!/usr/bin/env python3
"""dcs_decode_batch_live: microseconds per one-shot call over batch sizes, with the packages read / the PCM written over the link
(zero copy) or copied, by threshold (DCS_LIVE_ZC_UP_KB, DCS_LIVE_ZC_DOWN_FRAMES).  How the defaults were chosen."""
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pump_bench as P

sizes = [1, 8, 64, 256, 512, 1024, 2000]
with tempfile.TemporaryDirectory() as tmp:
    inp = P.make_inputs(tmp)
    name, os_, path = inp["recipe"][-1]
    print("%-34s" % "up KB / down frames" + "".join("%9d" % n for n in sizes))
    for up in (0, 64, 256, 1024, 1 << 20):
        for down in (0, 64, 256, 1024, 1 << 20):
            env = dict(os.environ, DCS_LIVE_ZC_UP_KB=str(up), DCS_LIVE_ZC_DOWN_FRAMES=str(down))
            p = subprocess.run([P.BUILDS["hip-mirror"], "oneshot", str(os_), "255", "100", "200", path] + [str(n) for n in sizes],
                               capture_output=True, text=True, env=env)
            if p.returncode != 0:
                print(up, down, "failed", p.stderr[-200:]); continue
            calls = json.loads(p.stdout.strip().splitlines()[-1])["calls"]
            print("%-34s" % ("zc up <= %d KB, down <= %d frames" % (up, down)) + "".join("%9.1f" % c["us_per_call"] for c in calls))

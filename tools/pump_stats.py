#!/usr/bin/env python3
"""where the class's time goes: the recipe / extract / script runs of tools/pump_bench.py on the stand-alone HIP build with
DCS_CLASS_STATS=1 DCS_LIVE_STATS=1 (lines on stderr when the decoder objects and their context go)"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pump_bench as P

env = dict(os.environ, DCS_CLASS_STATS="1", DCS_LIVE_STATS="1")
la = sys.argv[1] if len(sys.argv) > 1 else "-1"
with tempfile.TemporaryDirectory() as tmp:
    inp = P.make_inputs(tmp)
    exe = P.BUILDS["hip-mirror"]
    for name, os_, path in inp["recipe"]:
        p = subprocess.run([exe, "recipe", str(os_), "255", "100", la, "3", "-", path], capture_output=True, text=True, env=env)
        print("== recipe", name, p.stdout.strip()[-160:]); print(p.stderr)
    p = subprocess.run([exe, "extract", str(inp["extract"]["os"]), "255", "100", la, "2", "-"] + inp["extract"]["paths"], capture_output=True, text=True, env=env)
    print("== extract", p.stdout.strip()[-160:]); print(p.stderr)
    s = inp["script"]
    p = subprocess.run([exe, "script", str(s["volume"]), la, "2", "-", str(s["ticks"]), s["events"]] + s["roms"], capture_output=True, text=True, env=env)
    print("== script", p.stdout.strip()[-160:]); print(p.stderr)

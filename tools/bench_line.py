#!/usr/bin/env python3
"""print a compact summary of bench.py JSON lines read from stdin"""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(d["config"]["workload"], "fpw", d["config"]["frames_per_wave"], "%.3e samples/s" % d["value"],
          "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_avg_ms"],
          "GB/s %.1f" % d["roofline"]["achieved"], "bit_exact", d.get("bit_exact"), *sys.argv[1:])

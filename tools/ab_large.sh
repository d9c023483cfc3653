#!/bin/bash
# tools/ab_large.sh libA.so libB.so ... -- the large-batch rates (1 048 576 1994+ frames, 262 144 1993 frames) of several builds
for i in 1 2 3; do for lib in "$@"; do DCS_HIP_LIB=$PWD/$lib python bench.py --workload dcs94_65536 --scale 16 --no-cpu-baseline --no-end-to-end --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib x16 %.2f us %s' % (d['roofline']['kernel_avg_ms']*1e3, d['bit_exact']))"; DCS_HIP_LIB=$PWD/$lib python bench.py --workload dcs93_4096 --scale 64 --no-cpu-baseline --no-end-to-end --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib 93x64 %.2f us %s' % (d['roofline']['kernel_avg_ms']*1e3, d['bit_exact']))"; done; done | sort

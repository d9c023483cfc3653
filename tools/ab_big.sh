#!/bin/bash
# tools/ab_big.sh rounds libA.so libB.so ... -- kernel time on the workloads that run 8 and 16 frames per wavefront
N=$1; shift
for i in $(seq $N); do for lib in "$@"; do for spec in dcs94_65536:1 dcs94_65536:16 dcs93_4096:64; do
  wl=${spec%%:*}; sc=${spec##*:}
  DCS_HIP_LIB=$PWD/$lib python bench.py --workload $wl --scale $sc --no-cpu-baseline --no-end-to-end --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib ${wl}x$sc %.2f' % (d['roofline']['kernel_avg_ms']*1e3))"
done; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k])m[k]=$3} END{for(k in s) printf "%s mean %.2f min %.2f us\n", k, s[k]/n[k], m[k]}' | sort -k2,2 -k1,1

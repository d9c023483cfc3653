#!/bin/bash
# tools/ab.sh libA.so libB.so [rounds] -- kernel time of two builds of the library, interleaved on the same GPU box
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for lib in $A $B; do for wl in dcs93_4096 dcs94_65536 mixed_16384; do
  DCS_HIP_LIB=$PWD/$lib python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib %-12s %.2f us' % ('$wl', d['roofline']['kernel_avg_ms']*1e3))"
done; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k])m[k]=$3} END{for(k in s) printf "%s mean %.2f min %.2f us\n", k, s[k]/n[k], m[k]}' | sort

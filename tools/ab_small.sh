#!/bin/bash
# tools/ab_small.sh rounds libA.so libB.so ... -- like ab.sh, only the two workloads that run 4 frames per wavefront
N=$1; shift
for i in $(seq $N); do for lib in "$@"; do for wl in dcs93_4096 mixed_16384; do
  DCS_HIP_LIB=$PWD/$lib python bench.py --workload $wl --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib %-12s %.2f us %s' % ('$wl', d['roofline']['kernel_avg_ms']*1e3, d['bit_exact']))"
done; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k])m[k]=$3; if($5!="True")bad[k]=1} END{for(k in s) printf "%s mean %.2f min %.2f us%s\n", k, s[k]/n[k], m[k], (k in bad)?"  NOT BIT-EXACT":""}' | sort -k2,2 -k1,1

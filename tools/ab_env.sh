#!/bin/bash
# tools/ab_env.sh rounds VAR -- the three workloads (+ the two large ones) with and without an environment switch set
N=$1; V=$2
for i in $(seq $N); do for mode in off on; do for spec in dcs93_4096:1 dcs94_65536:1 mixed_16384:1 dcs94_65536:16 dcs93_4096:64 mixed_16384:8; do
  wl=${spec%%:*}; sc=${spec##*:}
  if [ $mode = on ]; then export $V=1; else unset $V; fi
  python bench.py --workload $wl --scale $sc --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V=$mode ${wl}x$sc %.2f %s' % (d['roofline']['kernel_avg_ms']*1e3, d['bit_exact']))"
done; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if($4!="True" && $4!="None")bad[k]=1} END{for(k in s) printf "%s mean %.2f us%s\n", k, s[k]/n[k], (k in bad)?"  NOT BIT-EXACT":""}' | sort -k2,2 -k1,1

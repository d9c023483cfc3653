#!/bin/bash
# tools/ab.sh [-w "workload[:scale] ..."] [-s steps] rounds libA.so libB.so ...
#   kernel time of several builds of the library (make -C dcsexplorer_amd/csrc variant NAME=.. DEFS=..), interleaved on the
#   same GPU box; -w picks the workloads (default "dcs93_4096 dcs94_65536 mixed_16384"; presets: small = the two that run
#   4 frames per wavefront, big = the ones that run 8 and 16, large = the 1 048 576-frame batch), workload:K = --scale K
WLS="dcs93_4096 dcs94_65536 mixed_16384"; STEPS=20
while getopts "w:s:" o; do case $o in w) WLS=$OPTARG;; s) STEPS=$OPTARG;; esac; done; shift $((OPTIND - 1))
case "$WLS" in
  small) WLS="dcs93_4096 mixed_16384";;
  big)   WLS="dcs94_65536 dcs94_65536:16 dcs93_4096:64";;
  large) WLS="dcs94_65536:16";;
esac
N=$1; shift
for i in $(seq $N); do for lib in "$@"; do for spec in $WLS; do
  wl=${spec%%:*}; sc=1; [ "$spec" != "$wl" ] && sc=${spec##*:}
  DCS_HIP_LIB=$PWD/$lib python bench.py --workload $wl --scale $sc --steps $STEPS --no-cpu-baseline --no-end-to-end --no-device-path --no-second-workload --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib %-18s %.2f us %s' % ('$spec', d['roofline']['kernel_avg_ms']*1e3, d['bit_exact']))"
done; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k])m[k]=$3; if($5!="True")bad[k]=1} END{for(k in s) printf "%s mean %.2f min %.2f us%s\n", k, s[k]/n[k], m[k], (k in bad)?"  NOT BIT-EXACT":""}' | sort -k2,2 -k1,1

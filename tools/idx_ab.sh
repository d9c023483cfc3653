#!/bin/bash
# tools/idx_ab.sh rounds libA.so libB.so ... -- the device index pass of one 256 x 256 list with several builds of the library, interleaved
N=$1; shift
for i in $(seq $N); do for lib in "$@"; do
  DCS_HIP_LIB=$PWD/$lib python tools/index_gpu_time.py one 2>/dev/null | awk -v l=$lib '/index kernel/ {print l, $(NF-1)}'
done; done | sort | awk '{s[$1]+=$2; n[$1]++; if(!($1 in m)||$2<m[$1])m[$1]=$2} END{for(k in s) printf "%s mean %.3f min %.3f ms\n", k, s[k]/n[k], m[k]}' | sort

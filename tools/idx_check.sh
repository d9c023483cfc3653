#!/bin/bash
# tools/idx_check.sh libA.so libB.so ...  -- on the GPU box: the index tests (records == host walk on every layout), then the device index
# pass of the default workload with each build: one list (256 streams), 8 and 32 lists in one launch; three interleaved rounds
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index" 2>&1 | tail -2
for i in 1 2 3; do for lib in "$@"; do
  DCS_HIP_LIB=$PWD/dcsexplorer_amd/$lib python tools/index_gpu_time.py survey3 2>/dev/null | awk -v l=$lib '/index kernel/ {printf "%s %s %s\n", l, $2, $(NF-1)}'
done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k])m[k]=$3} END{for(k in s) printf "%-32s streams mean %.3f min %.3f ms\n", k, s[k]/n[k], m[k]}' | sort -k1,1 -k2,2n

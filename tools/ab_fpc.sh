#!/bin/bash
# tools/ab_fpc.sh rounds -- one wavefront per frame (and per two frames) against four frames per wavefront, same kernel,
# on the 4 096- and 16 384-frame 1993 batches: occupancy 4 / 2 / 1 wavefronts per SIMD at 4 096 frames
N=${1:-3}
for i in $(seq $N); do for sc in 1 4; do for fpc in 1 2 4; do
  python bench.py --workload dcs93_4096 --scale $sc --fpw 4 --frames-per-chunk $fpc --no-cpu-baseline --no-end-to-end --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frames %6d fpc %d %.2f us waves %d' % (d['config']['frames_rank0_per_step'], $fpc, d['roofline']['kernel_avg_ms']*1e3, d['config']['wavefronts_per_launch']))"
done; done; done | sort | awk '{k=$1" "$2" "$3" "$4" waves "$8; s[k]+=$5; n[k]++} END{for(k in s) printf "%s mean %.2f us\n", k, s[k]/n[k]}' | sort -k2,2n -k4,4n

#!/bin/bash
# tools/sweep_fpw.sh -- kernel time per frames-per-wavefront setting over batch sizes (run on the GPU box)
run() { python bench.py --workload $1 --scale $2 --no-cpu-baseline --no-end-to-end --fpw $3 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 x$2 (%d frames) fpw $3: kernel %.2f us  %.3e samples/s' % (d['config']['frames_per_gpu_per_step'], d['roofline']['kernel_avg_ms']*1e3, d['value']))"; }
for sc in 2 4 16; do for f in 8 16; do run dcs94_65536 $sc $f; done; done
for sc in 2 4 8 16 64; do for f in 4 8 16; do run dcs93_4096 $sc $f; done; done
for sc in 2 4; do for f in 4 8 16; do run mixed_16384 $sc $f; done; done

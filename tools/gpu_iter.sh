#!/bin/bash
# tools/gpu_iter.sh <tag> [notest]  -- one kernel iteration on the GPU box (via gpurun): parity tests, the three
# bench workloads (kernel time only), phase stamps when the diagnostic library was built.
TAG=${1:-it}; OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$2" != "notest" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
  tail -2 $OUT/pytest.log
fi
for wl in dcs93_4096 dcs94_65536 mixed_16384; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-end-to-end > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 1; }
  python - $OUT/bench_$wl.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-12s kernel %.2f us  value %.3e  bit_exact %s" % (d["config"]["workload"], d["roofline"]["kernel_avg_ms"]*1e3, d["value"], d.get("bit_exact")))
PY
done
if [ -f dcsexplorer_amd/libdcs_hip_stamps.so ]; then
  for wl in dcs93_4096 dcs94_65536; do timeout -k 10 120 python tools/stamps.py $wl 2>/dev/null | tee $OUT/stamps_$wl.txt; done
fi

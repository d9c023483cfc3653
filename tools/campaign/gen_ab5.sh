python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpus.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do for v in "$@"; do for spec in survey3_65536 dcs94_65536 realistic_65536 mixed_16384 dcs93_4096:4 survey3_65536:16; do
  wl=${spec%%:*}; sc=1; [ "$spec" != "$wl" ] && sc=${spec##*:}
  lib=$PWD/dcsexplorer_amd/libdcs_hip_$v.so; [ $v = ship ] && lib=$PWD/dcsexplorer_amd/libdcs_hip.so
  DCS_HIP_LIB=$lib python bench.py --workload $wl --scale $sc --steps 100 --no-cpu-baseline --no-end-to-end --no-device-path --no-second-workload --no-class-surface --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v %-18s %.2f us %s' % ('$spec', d['roofline']['kernel_avg_ms']*1e3, d['bit_exact']))"
done; done; done

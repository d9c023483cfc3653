# tools/ab_cold.sh libA.so libB.so ... : warm and cold kernel time of the default workload per build, three interleaved rounds
for i in 1 2 3; do for lib in "$@"; do
  DCS_HIP_LIB=$PWD/dcsexplorer_amd/$lib python bench.py --steps 200 --no-cpu-baseline --no-end-to-end --no-device-path --no-second-workload 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib warm %.2f us cold %.2f us bit_exact %s cold_exact %s' % (d['roofline']['kernel_avg_ms']*1e3, d['roofline_cold']['kernel_avg_ms']*1e3, d['bit_exact'], d['roofline_cold']['bit_exact']))"
done; done

for i in 1 2 3; do for lib in dcsexplorer_amd/libdcs_hip_base.so dcsexplorer_amd/libdcs_hip.so; do DCS_HIP_LIB=$PWD/$lib python bench.py --workload dcs94_65536 --scale 16 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib x16 %.2f us' % (d['roofline']['kernel_avg_ms']*1e3))"; DCS_HIP_LIB=$PWD/$lib python bench.py --workload dcs93_4096 --scale 64 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib 93x64 %.2f us' % (d['roofline']['kernel_avg_ms']*1e3))"; done; done | sort

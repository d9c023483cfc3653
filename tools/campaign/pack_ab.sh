# the packer on the device, variants against the shipped one: device_full_path's per-kernel times
DCS_HIP_LIB=$PWD/dcsexplorer_amd/libdcs_hip_$1.so python -m pytest tests/test_gpu_corpus.py tests/test_gpu_parity.py -m gpu -x -q -k "pack or device or pipeline" 2>&1 | tail -1
for i in 1 2 3; do for v in ship "$@"; do
  lib=$PWD/dcsexplorer_amd/libdcs_hip_$v.so; [ $v = ship ] && lib=$PWD/dcsexplorer_amd/libdcs_hip.so
  DCS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-end-to-end --no-second-workload --no-class-surface --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d['device_full_path']; k=fp['saturated']['kernel_ms']; k1=fp['one_list']['kernel_ms']
print('$v sat pack %.4f plan %.4f total %.3f   one-list pack %.4f' % (k['dcsPackKernel'], [v for n,v in k.items() if 'Plan' in n][0], fp['saturated']['ms_per_pass'], k1['dcsPackKernel']))"
done; done

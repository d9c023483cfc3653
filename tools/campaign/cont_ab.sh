# the index kernel with looks that serve several bands (libdcs_hip_cont.so) against the shipped one
DCS_HIP_LIB=$PWD/dcsexplorer_amd/libdcs_hip_cont.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpus.py -m gpu -x -q -k "index or device or pipeline" 2>&1 | tail -2
for i in 1 2 3; do for v in ship cont; do
  lib=$PWD/dcsexplorer_amd/libdcs_hip_$v.so; [ $v = ship ] && lib=$PWD/dcsexplorer_amd/libdcs_hip.so
  DCS_HIP_LIB=$lib python tools/index_gpu_time.py survey3 2>/dev/null | grep -v "1024 streams\|2048 streams\|6144 streams" | sed "s/^/$v /"
done; done
for v in ship cont; do
  lib=$PWD/dcsexplorer_amd/libdcs_hip_$v.so; [ $v = ship ] && lib=$PWD/dcsexplorer_amd/libdcs_hip.so
  DCS_HIP_LIB=$lib python tools/index_gpu_time.py realistic 2>/dev/null | grep -v "1024 streams\|2048 streams\|6144 streams" | sed "s/^/$v /"
  DCS_HIP_LIB=$lib python tools/index_gpu_time.py dcs94 2>/dev/null | grep -v "1024 streams\|2048 streams\|6144 streams" | sed "s/^/$v /"
done

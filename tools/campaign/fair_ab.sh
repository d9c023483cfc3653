# A/B of the index kernel's fair shares (libdcs_hip_base.so: without, libdcs_hip_fair.so: with)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index" 2>&1 | tail -2
for i in 1 2 3; do for lib in libdcs_hip_base.so libdcs_hip_fair.so; do
  DCS_HIP_LIB=$PWD/dcsexplorer_amd/$lib python tools/index_gpu_time.py survey3 2>/dev/null | sed "s/^/$lib /"
done; done
for lib in libdcs_hip_base.so libdcs_hip_fair.so; do
  DCS_HIP_LIB=$PWD/dcsexplorer_amd/$lib python tools/index_gpu_time.py corpus 2>/dev/null | sed "s/^/$lib /"
  DCS_HIP_LIB=$PWD/dcsexplorer_amd/$lib python tools/index_gpu_time.py realistic 2>/dev/null | sed "s/^/$lib /"
done

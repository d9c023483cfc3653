# sweep of the index kernel's pacing parameters: libdcs_hip_p<name>.so built by `make variant` (see tools/campaign/README)
for i in 1 2; do for v in off c h i j k; do
  lib=libdcs_hip_p$v.so
  for wl in survey3 realistic corpus; do
    DCS_HIP_LIB=$PWD/dcsexplorer_amd/$lib python tools/index_gpu_time.py $wl 2>/dev/null | grep -v " 256 streams\| 1024 streams\| 2048 streams" | sed "s/^/$v /"
  done
done; done

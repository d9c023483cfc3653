#!/bin/bash
# tools/asan_cpu.sh -- build the library with AddressSanitizer + UBSan on the HOST side (device code untouched;
# GPU sanitizers are not available on this pool) and run the CPU test files that exercise the host code
# (index pass, planner, ROM ingestion, sequencer, file formats) under it.  The reference build in oracle/_ref
# pairs new[] with delete in its own destructors, hence alloc_dealloc_mismatch=0.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/dcs_asan; mkdir -p $OUT; rm -f $OUT/asanlog* $OUT/*.txt
CL=/opt/rocm/lib/llvm/bin/clang++
for f in dcs_tables dcs_index dcs_params dcs_synth dcs_plan dcs_streams dcs_files dcs_rom dcs_sequencer dcs_decoder_hip; do
  $CL -x c++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -c $ROOT/dcsexplorer_amd/csrc/$f.cpp -o $OUT/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -include $ROOT/dcsexplorer_amd/csrc/build/dcs_build_id.h -c $ROOT/dcsexplorer_amd/csrc/dcs_runtime.hip -o $OUT/dcs_runtime.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $OUT/libdcs_hip.so $OUT/*.o -lz -Wl,-rpath,/opt/rocm/lib
cp $ROOT/dcsexplorer_amd/libdcs_hip.so $OUT/libdcs_hip_orig.so
trap 'cp $OUT/libdcs_hip_orig.so $ROOT/dcsexplorer_amd/libdcs_hip.so' EXIT
cp $OUT/libdcs_hip.so $ROOT/dcsexplorer_amd/libdcs_hip.so
ASAN=$($CL -print-file-name=libclang_rt.asan-x86_64.so)
cd $ROOT
rc=0
for t in tests/test_host.py tests/test_files.py tests/test_rom.py tests/test_sequencer.py tests/test_abi.py tests/test_multirank_gloo.py; do
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0:log_path=$OUT/asanlog \
    python -m pytest $t -x -q -m "not gpu" -p no:cacheprovider > $OUT/$(basename $t).txt 2>&1 || rc=1
  echo "$t: $(tail -1 $OUT/$(basename $t).txt)"
done
if ls $OUT/asanlog* >/dev/null 2>&1 || grep -l "runtime error" $OUT/*.txt >/dev/null 2>&1; then echo "sanitizer reports in $OUT"; rc=1; fi
exit $rc

#!/bin/bash
# tools/tsan_walker.sh -- the host side of the sequencer (index walk, background walker, snapshots and replays) under ThreadSanitizer:
# tests/cpp/dcs_walker_tsan.cpp plans long streams of every layout while they are being walked on the walker's thread, goes back
# inside the plan, loads a second stream (which waits for the first walk), clears the tracks and destroys sequencers with a walk in
# progress.  No GPU (the decode entry points are stubs).  Prints ThreadSanitizer's reports, "no data race reported" if there are none.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/dcs_tsan; mkdir -p $OUT
CL=/opt/rocm/lib/llvm/bin/clang++
for f in dcs_tables dcs_index dcs_params dcs_synth dcs_plan dcs_rom dcs_sequencer dcs_files; do
  $CL -x c++ -O1 -g -std=c++17 -fPIC -fsanitize=thread -c $ROOT/dcsexplorer_amd/csrc/$f.cpp -o $OUT/$f.o
done
$CL -O1 -g -std=c++17 -fsanitize=thread $ROOT/tests/cpp/dcs_walker_tsan.cpp $OUT/*.o -lz -lpthread -o $OUT/dcs_walker_tsan
TSAN_OPTIONS="halt_on_error=0" $OUT/dcs_walker_tsan > $OUT/run.txt 2>&1 || true
tail -3 $OUT/run.txt
if grep -q "WARNING: ThreadSanitizer" $OUT/run.txt; then grep -A12 "WARNING: ThreadSanitizer" $OUT/run.txt | head -60; exit 1; fi
grep -q "^done" $OUT/run.txt && echo "no data race reported"

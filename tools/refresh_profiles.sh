#!/bin/bash
# tools/refresh_profiles.sh <round-tag>  -- on the GPU box (via gpurun): the rocprofv3 kernel-trace summary and the PMC
# passes of every bench workload (tools/prof.sh), copied to profiles/<tag>_<workload>_rocprofv3_summary.txt and
# profiles/traffic_<workload>.json under gpurun_out/profiles_new/, then the bench line of each workload with those
# counters in place.  Copy gpurun_out/profiles_new/* into profiles/ afterwards.
TAG=${1:-r02}
NEW=$PWD/gpurun_out/profiles_new; mkdir -p $NEW
for wl in survey3_65536 dcs94_65536 dcs93_4096 mixed_16384 realistic_65536 corpus; do
  # (the status is prof.sh's, not grep's: a prof.sh that dies after its first pass must not go unnoticed -- ADVICE r4)
  bash tools/prof.sh ${TAG}_$wl $wl > gpurun_out/prof_${TAG}_$wl.log 2>&1; rc=$?
  grep '^prof ' gpurun_out/prof_${TAG}_$wl.log
  if [ $rc -ne 0 ] || [ ! -s gpurun_out/prof_${TAG}_$wl/traffic.json ] || [ ! -s gpurun_out/prof_${TAG}_$wl/summary.txt ]; then
    echo "prof $wl failed (rc $rc)"; tail -5 gpurun_out/prof_${TAG}_$wl.log; exit 1
  fi
  cp gpurun_out/prof_${TAG}_$wl/summary.txt $NEW/${TAG}_${wl}_rocprofv3_summary.txt
  cp gpurun_out/prof_${TAG}_$wl/traffic.json $NEW/traffic_$wl.json
  cp gpurun_out/prof_${TAG}_$wl/traffic.json profiles/traffic_$wl.json
  echo "profiled $wl"
done
for wl in survey3_65536 dcs94_65536 dcs93_4096 mixed_16384 realistic_65536 corpus; do
  # (the class surface is timed once, with the default workload's line)
  python bench.py --workload $wl $([ $wl = survey3_65536 ] || echo --no-class-surface) > $NEW/${TAG}_bench_$wl.json 2> gpurun_out/bench_${TAG}_$wl.err || { echo "bench $wl failed"; tail -5 gpurun_out/bench_${TAG}_$wl.err; exit 1; }
  echo "benched $wl"
done

#!/bin/bash
# tools/prof_pipeline.sh <tag> [depth] [devpack|devplan]  -- on the GPU box: rocprofv3 kernel trace of the pipeline with index pass, (planner) and packer on
# the device (tools/pipe_trace.py <depth> devpack): which kernels the GPU spends its time in end to end
TAG=${1:-r03}; DEPTH=${2:-48}
OUT=$PWD/gpurun_out/prof_pipeline_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 $PWD/tools/pipe_trace.py $DEPTH ${3:-devplan}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/run.txt 2> $OUT/trace.log
cd $OUT
( cut -c1-160 run.txt; for f in $(find trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done ) > summary.txt
cat summary.txt

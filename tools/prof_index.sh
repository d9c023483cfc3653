#!/bin/bash
# tools/prof_index.sh <tag> [lists]  -- on the GPU box (via gpurun): rocprofv3 kernel trace and PMC passes of the device index
# pass (dcsIndexWaveKernel) over <lists> x 256 streams x 256 frames; summary under gpurun_out/prof_index_<tag>/summary.txt
set -u
TAG=${1:-r03}; MULT=${2:-1}
OUT=$PWD/gpurun_out/prof_index_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 $PWD/tools/index_prof_run.py $MULT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/run_trace.txt 2> $OUT/trace.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- $RUN > /dev/null 2> $OUT/pmc_$name.log
done
cd $OUT
python3 - <<'PY'
import csv, glob, collections
out = open("summary.txt", "w")
out.write(open("run_trace.txt").read())
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    out.write("== %s\n" % f); out.write(open(f).read())
for f in sorted(glob.glob("pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", "")[:60], row.get("Counter_Name", ""))
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    out.write("== %s\n" % f)
    for (kn, cn), (n, v) in sorted(agg.items()):
        if "dcsIndex" in kn:
            out.write("%-60s %-28s dispatches=%d avg=%.1f\n" % (kn, cn, n, v / n))
out.close()
print(open("summary.txt").read())
PY

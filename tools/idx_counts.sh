#!/bin/bash
# tools/idx_counts.sh libA.so libB.so ...  -- on the GPU box: scalar / vector / LDS instructions per frame of dcsIndexWaveKernel for each
# build (one rocprofv3 --pmc pass over one list of survey3_65536; the counts per frame do not depend on the number of lists)
export TMPDIR=/tmp
REPO=$PWD
for lib in "$@"; do
  OUT=$REPO/gpurun_out/idx_counts_${lib%.so}
  rm -rf $OUT; mkdir -p $OUT
  export DCS_HIP_LIB=$REPO/dcsexplorer_amd/$lib
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -- python3 $REPO/tools/index_prof_run.py 1 survey3_65536 > $OUT/run.txt 2> $OUT/log.txt)
  python3 - $OUT $lib <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "dcsIndex" in row.get("Kernel_Name", ""):
            k = row["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"] or 0)
frames = 65536.0
print("%-28s" % sys.argv[2], "  ".join("%s/frame %.1f" % (k.replace("SQ_INSTS_", ""), v / n / frames) for k, (n, v) in sorted(agg.items()) if k != "SQ_WAVES"))
PY
done

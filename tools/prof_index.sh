#!/bin/bash
# tools/prof_index.sh <tag> [lists] [workload]  -- on the GPU box (via gpurun): rocprofv3 kernel trace and PMC passes of the device
# index pass (dcsIndexWaveKernel) over <lists> x 256 streams x 256 frames; summary under gpurun_out/prof_index_<tag>/summary.txt and
# the per-launch counters with the library's build id in traffic_index.json (-> profiles/traffic_index_<workload>.json)
set -u
TAG=${1:-r04}; MULT=${2:-1}; WL=${3:-survey3_65536}
REPO=$PWD
OUT=$PWD/gpurun_out/prof_index_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 $PWD/tools/index_prof_run.py $MULT $WL"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/run_trace.txt 2> $OUT/trace.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- $RUN > /dev/null 2> $OUT/pmc_$name.log
  echo "prof_index $TAG: pass $name done"
done
cd $OUT
DCS_REPO=$REPO DCS_WL=$WL DCS_MULT=$MULT python3 - <<'PY'
import csv, glob, collections, os, sys, re, json
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
# per-launch counters of the index kernel, tied to the library build (bench.py device_full_path reads them)
vals = {}
for line in open("summary.txt"):
    m = re.search(r"dcsIndexWaveKernel.*\s([A-Z_]+)\s+dispatches=(\d+) avg=([0-9.]+)", line)
    if m:
        vals[m.group(1)] = float(m.group(3))
    m = re.search(r": (\d+) streams, (\d+) frames: index kernel ([0-9.]+) ms", line)
    if m:
        vals["streams"] = int(m.group(1)); vals["frames"] = int(m.group(2)); vals["index_ms_events"] = float(m.group(3))
    m = re.match(r'"dcsidx::dcsIndexWaveKernel\(.*\)",(\d+),(\d+),([0-9.]+)', line)
    if m:
        vals["trace_calls"] = int(m.group(1)); vals["trace_avg_ns"] = float(m.group(3))
if "SQ_INSTS_SALU" in vals and "frames" in vals:
    sys.path.insert(0, os.environ["DCS_REPO"])
    import dcsexplorer_amd as D
    d = dict(vals)
    d["workload"] = os.environ["DCS_WL"]; d["lists"] = int(os.environ["DCS_MULT"])
    # FETCH_SIZE / WRITE_SIZE are kilobytes; the walk's reads are 4-byte-per-lane loads (no wide-read correction)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        d["hbm_bytes_per_launch"] = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        d["hbm_bytes_per_frame"] = d["hbm_bytes_per_launch"] / vals["frames"]
    d["lib_build_id"] = D.build_id(); d["lib_sha256"] = D.lib_sha256()
    json.dump(d, open("traffic_index.json", "w"))
PY

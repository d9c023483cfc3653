#!/bin/bash
# tools/prof.sh <tag> <workload> [extra bench args]  -- run on the GPU box (via gpurun).
# Writes rocprofv3 kernel-trace stats and PMC counter passes under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}; WL=${2:-survey3_65536}; shift 2 || true
REPO=$PWD
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-class-surface --no-end-to-end --no-second-workload --no-device-path --rotate 0 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
echo "prof $TAG: kernel trace done"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- $BENCH > /dev/null 2> $OUT/pmc_$name.log
  echo "prof $TAG: pass $name done"
done
cd $OUT
# compact summaries
DCS_REPO=$REPO DCS_WL=$WL DCS_EXTRA="$*" python3 - <<'PY'
import csv, glob, collections, os, sys
out = open("summary.txt", "w")
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    out.write("== %s\n" % f); out.write(open(f).read())
for f in sorted(glob.glob("pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", "")[:60], row.get("Counter_Name", ""))
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    out.write("== %s\n" % f)
    for (kn, cn), (n, v) in sorted(agg.items()):
        if "dcsDecode" in kn:
            out.write("%-60s %-28s dispatches=%d avg=%.1f\n" % (kn, cn, n, v / n))
out.close()
print(open("summary.txt").read())
# HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes (kilobytes; MI355X_MICROARCH.md: on gfx950
# FETCH_SIZE tallies 128-byte requests at 64 bytes for wide streaming reads -- reported raw AND doubled)
import json, re
vals = {}
for line in open("summary.txt"):
    m = re.search(r"dcsDecodeKernel<(\d+)>.*\s([A-Z_]+)\s+dispatches=(\d+) avg=([0-9.]+)", line)
    if m:
        vals[m.group(2)] = float(m.group(4)); vals["fpw"] = int(m.group(1))
    m = re.match(r'"void dcsk::dcsDecodeKernel<\d+>\(.*\)",(\d+),(\d+),([0-9.]+)', line)
    if m:
        vals["trace_calls"] = int(m.group(1)); vals["trace_avg_ns"] = float(m.group(3))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    d = {"fetch_kb_raw": vals["FETCH_SIZE"], "write_kb": vals["WRITE_SIZE"], "fpw": vals["fpw"],
         "traffic_bytes_fetch_raw": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
         "traffic_bytes_fetch_x2": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024}
    # issue-side counters of the same kernel (per launch), so that the HBM fraction can be read next to what
    # actually bounds the kernel: VALU instruction issue (4 cycles per wave64 instruction per SIMD)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
              "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"):
        if k in vals:
            d[k] = vals[k]
    if "trace_avg_ns" in vals:
        d["trace_avg_ns"] = vals["trace_avg_ns"]; d["trace_calls"] = vals["trace_calls"]
    # which binary and which workload these counters belong to: bench.py reports them only for a library with this id
    sys.path.insert(0, os.environ["DCS_REPO"])
    import dcsexplorer_amd as D
    d["workload"] = os.environ["DCS_WL"]; d["bench_args"] = os.environ.get("DCS_EXTRA", "")
    d["lib_build_id"] = D.build_id(); d["lib_sha256"] = D.lib_sha256()
    json.dump(d, open("traffic.json", "w"))
PY

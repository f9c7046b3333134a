#!/bin/bash
# rocprofv3 passes for the headline kernel: kernel-trace stats, then PMC passes (each on its own).
# usage: tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r2}"
OUT="$R/gpurun_out/prof_$TAG"
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (1) the DEFAULT bench command under --kernel-trace --stats; tools/trace_phases.py splits it by workload
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 $R/bench.py > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace_default.log"
python3 $R/tools/trace_phases.py "$OUT/trace_default" > "$OUT/kernel_stats_by_workload.txt"
# (2) PMC passes on the headline alone (--only-headline: one kernel template = one workload), few launches each
BENCH="python3 $R/bench.py --no-cpu --only-headline --steps 5 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
i=0
for PMC in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" "$R" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
sys.path.insert(0, sys.argv[2])
from fss_amd import _build
res = collections.OrderedDict()
res["_captured_for"] = {"headline_sources_sha256": _build.headline_digest(), "sources": list(_build.HEADLINE_SOURCES)}
for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if any(t in row["Kernel_Name"] for t in ("eval_points_kernel", "expand_kernel", "frontier_kernel")):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in agg.items():
        res.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Name"].split("(")[0]
        if k in res:
            res[k]["AverageNs"] = float(row["AverageNs"]); res[k]["Calls"] = int(row["Calls"])
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

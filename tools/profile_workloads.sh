#!/bin/bash
# rocprofv3 evidence for every workload of bench.py's default line (round 3): 
#   (1) the DEFAULT command under --kernel-trace --stats (kernel_stats.csv; tools/trace_phases.py splits it by workload),
#   (2) per workload, `bench.py --workload <w> --no-cpu` under --kernel-trace --stats and under three SEPARATE --pmc
#       passes (LDS / cycles; FETCH_SIZE; WRITE_SIZE -- MI355X_MICROARCH.md: separate passes, FETCH_SIZE x2 on gfx950).
# Output: gpurun_out/prof_<tag>/ with pmc_workloads.json (what bench.py replays as `roofline.traffic` of the
# sub-records, each entry stamped with the digest of the sources its kernel is compiled from), pmc_summary.json (the
# headline's, the format of rounds 1-2) and kernel_stats_by_workload.txt.  Copy the summaries to profiles/r<N>/.
# usage: tools/profile_workloads.sh <tag>
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r3}"
OUT="$R/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 $R/bench.py --steps 20 --warmup 5 > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace_default.log"
python3 $R/tools/trace_phases.py "$OUT/trace_default" > "$OUT/kernel_stats_by_workload.txt"
find "$OUT/trace_default" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for W in dpf_points dpf_2e20 dcf_2e20 evalall_c4 ht_c5 grotto_c5; do
  EXTRA="--steps 10 --warmup 5"
  [ "$W" = dpf_points ] && EXTRA="--only-headline --steps 5 --warmup 2"
  BENCH="python3 $R/bench.py --workload $W --no-cpu $EXTRA"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/trace" -- $BENCH > "$OUT/$W/line.json" 2> "$OUT/$W/trace.log"
  i=0
  for PMC in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/$W/pmc$i" -- $BENCH > "$OUT/$W/pmc$i.log" 2>&1
  done
done
python3 - "$OUT" "$R" <<'PY'
import collections, csv, glob, json, sys
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from fss_amd import _build
KERNEL = {"dpf_points": "eval_points_kernel", "dpf_2e20": "eval_points_kernel", "dcf_2e20": "eval_points_kernel",
          "evalall_c4": "evalall_fused_kernel", "ht_c5": "eval_points_kernel", "grotto_c5": "grotto_walk"}
res = collections.OrderedDict()
for w, ksub in KERNEL.items():
    name = "headline" if w == "dpf_points" else w
    try:
        line = json.loads([l for l in open(f"{out}/{w}/line.json") if l.startswith("{")][-1])
    except Exception as e:
        print("no bench line for", w, e); continue
    agg, kname = collections.defaultdict(list), None
    for f in sorted(glob.glob(f"{out}/{w}/pmc*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if ksub in row["Kernel_Name"]:
                kname = row["Kernel_Name"].split("(")[0]
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    ent = {"kernel": kname, "units_per_launch": line["config"].get("units_per_launch", 1 << 24),
           "_captured_for": _build.workload_digest(name), "sources": list(_build.WORKLOAD_SOURCES[name]),
           "bench_line_of_the_trace_run": {"kernel_ms": line["roofline"].get("kernel_ms"), "frac": line["roofline"]["frac"]}}
    for c, v in agg.items():
        v = v[len(v) // 2:]  # the later half: the timed launches at steady clocks
        ent[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    for f in glob.glob(f"{out}/{w}/trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if ksub in row["Name"] and (kname is None or row["Name"].split("(")[0] == kname):
                ent["AverageNs"], ent["MinNs"], ent["Calls"] = float(row["AverageNs"]), float(row["MinNs"]), int(row["Calls"])
    res[name] = ent
json.dump(res, open(out + "/pmc_workloads.json", "w"), indent=1)
# the headline in the rounds 1-2 format (bench.py: _headline_pmc)
h = res.get("headline", {})
old = collections.OrderedDict()
old["_captured_for"] = {"headline_sources_sha256": _build.headline_digest(), "sources": list(_build.HEADLINE_SOURCES)}
if h.get("kernel"):
    old[h["kernel"]] = {k: v for k, v in h.items() if isinstance(v, dict) and "mean_per_launch" in v}
    old[h["kernel"]].update({"AverageNs": h.get("AverageNs"), "Calls": h.get("Calls")})
json.dump(old, open(out + "/pmc_summary.json", "w"), indent=1)
for k, v in res.items():
    cyc = v.get("GRBM_GUI_ACTIVE", {}).get("mean_per_launch", 0) / 8
    line = f"{k:12s} {str(v['kernel'])[-70:]:70s} avg_us {v.get('AverageNs', 0) / 1e3:9.2f}"
    if cyc:
        line += f"  LDS busy {v['SQ_LDS_IDX_ACTIVE']['mean_per_launch'] / 256 / cyc:.4f}  conflicts {v.get('SQ_LDS_BANK_CONFLICT', {}).get('mean_per_launch', -1):.0f}"
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        line += f"  HBM fetch x2 {2 * v['FETCH_SIZE']['mean_per_launch'] / 1024:.1f} MiB + write {v['WRITE_SIZE']['mean_per_launch'] / 1024:.1f} MiB"
    print(line)
PY

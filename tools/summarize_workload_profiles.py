#!/usr/bin/env python3
"""Summaries of tools/profile_workloads.sh's rocprofv3 output: <dir>/pmc_workloads.json (per workload: the dominant
kernel's counters per launch, kernel-trace average, the digest of the sources the kernel is compiled from -- what
bench.py replays as `roofline.traffic`), <dir>/pmc_summary.json (the headline in the format of rounds 1-2) and a table.
usage: tools/summarize_workload_profiles.py <dir>"""
import collections, csv, glob, json, sys
import os
out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    # the workload's dominant kernel: the one of the matching names with the most device time in the trace pass (the
    # headline run also launches the shared-prefix walk, the same template with another argument)
    kname, best = None, -1.0
    for f in glob.glob(f"{out}/{w}/trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if ksub in row["Name"] and float(row["TotalDurationNs"]) > best:
                kname, best = row["Name"].split("(")[0], float(row["TotalDurationNs"])
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{out}/{w}/pmc*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].split("(")[0] == kname:
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
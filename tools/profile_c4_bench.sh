#!/bin/bash
# The configs[3] workload exactly as bench.py runs it on one GPU (`python3 bench.py --workload evalall_c4 --no-cpu`: the whole
# 2^28 domain as one block) under rocprofv3: kernel stats, then one PMC pass -> gpurun_out/c4bench/
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/c4bench"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/bench.py --workload evalall_c4 --no-cpu > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace.log"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc1" -- python3 $R/bench.py --workload evalall_c4 --no-cpu --steps 10 --warmup 5 > "$OUT/pmc1.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc2" -- python3 $R/bench.py --workload evalall_c4 --no-cpu --steps 10 --warmup 5 > "$OUT/pmc2.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections, json
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fssamd" in r["Name"]:
            print("%-74s calls %4s avg_us %9.2f min_us %9.2f" % (r["Name"].split("(")[0][-74:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
try:
    d = json.loads(open(out + "/bench_line_under_rocprof.json").read().strip().splitlines()[-1])
    print("bench.py's own line of that run: kernel_ms %.3f, %.2f G leaves/s, frac %.4f" % (d["roofline"]["kernel_ms"], d["value"] / 1e9, d["roofline"]["frac"]))
except Exception as e:
    print("no bench line:", e)
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "evalall_fused_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
c = {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in agg.items()}
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("PMC: LDS busy %.4f of %.0f cycles per launch, bank conflicts %.0f" % (c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, cyc, c.get("SQ_LDS_BANK_CONFLICT", -1)))
if "WRITE_SIZE" in c:
    print("PMC: HBM writes %.1f MiB per launch (algorithmic: 4096 MiB of leaves + 278 MiB of nodes)" % (c["WRITE_SIZE"] / 1024))
PY

#!/bin/bash
# A bench.py workload exactly as the driver would run it on one GPU, under rocprofv3: kernel stats, then PMC passes.
# usage: tools/profile_workload.sh <ht_c5|grotto_c5|evalall_c4> <kernel name substring>  -> gpurun_out/wl_<workload>/summary.txt
R="${GRAFT_REPO_ROOT:-/root/repo}"
W="$1"; KSUB="$2"
OUT="$R/gpurun_out/wl_$W"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/bench.py --workload $W --no-cpu > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace.log"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc1" -- python3 $R/bench.py --workload $W --no-cpu --steps 10 --warmup 5 > "$OUT/pmc1.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc2" -- python3 $R/bench.py --workload $W --no-cpu --steps 10 --warmup 5 > "$OUT/pmc2.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc3" -- python3 $R/bench.py --workload $W --no-cpu --steps 10 --warmup 5 > "$OUT/pmc3.log" 2>&1
python3 - "$OUT" "$KSUB" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections, json
out, ksub = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fssamd" in r["Name"] and float(r["TotalDurationNs"]) > 1e6:
            print("%-78s calls %4s avg_us %9.2f min_us %9.2f" % (r["Name"].split("(")[0][-78:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
try:
    d = json.loads(open(out + "/bench_line_under_rocprof.json").read().strip().splitlines()[-1])
    print("bench.py's own line of that run: kernel_ms %.3f, value %.4g %s, frac %.4f" % (d["roofline"]["kernel_ms"], d["value"], d["unit"], d["roofline"]["frac"]))
    alg = d.get("roofline_hbm", {}).get("algorithmic_bytes_per_launch")
except Exception as e:
    print("no bench line:", e); alg = None
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
c = {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in agg.items()}
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("PMC: LDS busy %.4f of %.0f cycles per launch, bank conflicts %.0f" % (c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, cyc, c.get("SQ_LDS_BANK_CONFLICT", -1)))
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    print("PMC: HBM traffic per launch: fetch x2 %.1f MiB + write %.1f MiB%s" % (2 * c["FETCH_SIZE"] / 1024, c["WRITE_SIZE"] / 1024,
          (" (algorithmic %.1f MiB)" % (alg / 2**20)) if alg else ""))
PY

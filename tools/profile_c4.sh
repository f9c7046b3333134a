#!/bin/bash
# configs[3] per-GPU share (EvalAll n = 28, the 2^25-leaf block): per-kernel durations at steady clocks
# (300 back-to-back evaluations under rocprofv3 --kernel-trace --stats) next to the end-to-end time per call.
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/c4"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in dpf ht; do
  python3 $R/tools/bench_evalall.py --scheme $s --n 28 --log2-leaves 25 --steady > "$OUT/$s.unprofiled.txt" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$s" -- python3 $R/tools/bench_evalall.py --scheme $s --n 28 --log2-leaves 25 --steps 300 > "$OUT/$s.log" 2>&1
  tail -1 "$OUT/$s.unprofiled.txt"; tail -1 "$OUT/$s.log"
  python3 - "$OUT/$s" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "fssamd" in r["Name"]: print("%-78s calls %4s avg_us %8.2f total_us %10.1f"%(r["Name"].split("(")[0][-78:],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e3))
PY
done 2>&1 | tee "$OUT/summary.txt"

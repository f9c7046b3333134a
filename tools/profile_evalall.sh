#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for n in 24 26; do
rm -rf $R/gpurun_out/ea$n
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ea$n -- python3 $R/tools/bench_evalall.py --scheme dpf --n $n --steps 10 > $R/gpurun_out/ea$n.log 2>&1
tail -1 $R/gpurun_out/ea$n.log
python3 - $R/gpurun_out/ea$n <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "fssamd" in r["Name"]: print("%-70s calls %4s avg_us %8.1f total_us %9.1f"%(r["Name"].split("(")[0][-70:],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e3))
PY
done

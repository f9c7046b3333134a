#!/bin/bash
# usage: tools/profile_cmd.sh <tag> <kernel-substring> -- <python args...>   (rocprofv3 kernel stats + PMC passes)
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; KSUB="$2"; shift 3
OUT="$R/gpurun_out/prof_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$@" > "$OUT/trace.log" 2>&1
i=0
for PMC in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 "$@" > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"], row["Grid_Size"])].append(float(row["Counter_Value"]))
for k in sorted(agg): print("%-60s grid %-9s %-22s mean %.5g (n=%d)" % (k[0][-60:], k[2], k[1], sum(agg[k])/len(agg[k]), len(agg[k])))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Name"]: print(row["Name"].split("(")[0], "calls", row["Calls"], "avg_us", float(row["AverageNs"])/1e3)
PY

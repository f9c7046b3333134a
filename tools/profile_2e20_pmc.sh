#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# BASELINE configs[1] / [2] (one key, 2^20 points): LDS-busy fraction of the launch under rotating (STATIC_PRIO=1) and
# progress-ranked (default) wave priorities.  One PMC pass each, 400 launches -> gpurun_out/points_2e20_pmc.txt
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/p20"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in dpf dcf; do
for m in 1 2; do
  FSS_AMD_STATIC_PRIO=$m timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$OUT/$s$m" -- python3 $R/tools/bench_points.py --scheme $s --log2-points 20 --steps 400 > "$OUT/$s$m.log" 2>&1
done
done
python3 - "$OUT" <<'PY' | tee "$R/gpurun_out/points_2e20_pmc.txt"
import csv, glob, sys, collections
out = sys.argv[1]
for s in ("dpf", "dcf"):
    for m in ("1", "2"):
        agg = collections.defaultdict(list); dur = []
        for f in glob.glob(f"{out}/{s}{m}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "eval_points_kernel" in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for f in glob.glob(f"{out}/{s}{m}/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "eval_points_kernel" in row["Kernel_Name"]:
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        if not agg: print(s, m, "no data"); continue
        half = lambda v: v[len(v) // 2:]
        c = {k: sum(half(v)) / len(half(v)) for k, v in agg.items()}
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        d = sorted(dur)[len(dur) // 2]
        print(f"{s} n=32 2^20 points, STATIC_PRIO={m} ({'turns' if m == '1' else 'ranked'}): LDS busy {c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.4f} of {cyc:.0f} cycles, median {d:.1f} us under the pass, clock {cyc / d / 1e3:.3f} GHz")
PY

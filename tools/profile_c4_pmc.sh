#!/bin/bash
# configs[3] per-GPU share (EvalAll n = 28, 2^25-leaf block) as ONE launch (evalall_fused_kernel): PMC passes, each on
# its own, 60 evaluations per pass -> gpurun_out/c4pmc/summary.txt (LDS-busy and VALU-busy fractions, clock, HBM bytes).
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/c4pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in dpf ht; do
  i=0
  for PMC in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/$s$i" -- python3 $R/tools/bench_evalall.py --scheme $s --n 28 --log2-leaves 25 --steps 60 > "$OUT/$s$i.log" 2>&1
  done
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
for s in ("dpf", "ht"):
    agg = collections.defaultdict(list)
    dur = []
    for f in sorted(glob.glob(f"{out}/{s}[0-9]/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "evalall_fused_kernel" in row["Kernel_Name"] or "expand_kernel" in row["Kernel_Name"]:
                agg[(row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for f in sorted(glob.glob(f"{out}/{s}1/**/*kernel_trace.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "evalall_fused_kernel" in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    byk = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        v = v[len(v) // 2:]  # the second half of the run: clocks settled
        byk[k][c] = sum(v) / len(v)
    for k, c in byk.items():
        print(s, k)
        for name, val in sorted(c.items()):
            print(f"    {name:24s} {val:16.1f}")
        if "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            print(f"    cycles per launch (GUI_ACTIVE / 8 XCDs)      {cyc:12.0f}")
            if dur:
                d = sorted(dur)[len(dur) // 2]
                print(f"    median duration under the PMC pass           {d:12.2f} us -> clock {cyc / d / 1e3:.3f} GHz")
            if "SQ_LDS_IDX_ACTIVE" in c:
                print(f"    LDS busy (IDX_ACTIVE / 256 CUs / cycles)     {c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:12.4f}")
    for k, c in byk.items():
        if "SQ_ACTIVE_INST_VALU" in c:
            for k2, c2 in byk.items():
                if k2 == k and "GRBM_GUI_ACTIVE" in c2:
                    pass
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            print(f"    HBM: fetch x2 {2 * c.get('FETCH_SIZE', 0) / 1024:.1f} MiB, write {c.get('WRITE_SIZE', 0) / 1024:.1f} MiB per launch (algorithmic: 512 MiB of leaves)")
PY

#!/usr/bin/env python3
"""Summary of tools/profile_workloads.sh's shared-prefix passes: per kernel of the plan (node-table launch, walk) the
kernel-trace durations, and for the walk the PMC counters per launch (L2 hit rate and fabric reads of the node gathers,
LDS busy).  usage: summarize_shared_prefix.py <dir>"""
import collections, csv, glob, sys
d = sys.argv[1]
print(open(d + "/unprofiled.txt").read())
rows = collections.defaultdict(list)
for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void fssamd::", "").replace("fssamd::", "")
        if "eval_points_kernel" in n or "evalall_fused" in n or "expand_kernel" in n or "frontier" in n:
            rows[(n, int(r.get("Grid_Size_X", r.get("Grid_Size", 0))))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel trace of the same command (all dispatches of the run, by kernel and grid size):")
for (n, g), v in sorted(rows.items()):
    print(f"  {n[:72]:72s} grid {g:8d}  calls {len(v):5d}  avg_us {sum(v) / len(v):9.2f}  min_us {min(v):9.2f}")
agg = collections.defaultdict(list)
for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "eval_points_kernel" in k and "1024, true>" in k and int(r["Grid_Size"]) == 256 * 1024:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in agg.items()}
print("PMC, the walk kernel (eval_points_kernel<..., PREFIX = true>, DPF 2^24 points), mean per launch over the later half of the launches:")
for c, v in sorted(m.items()):
    print(f"  {c:28s} {v:16.1f}")
if "TCC_HIT_sum" in m:
    print(f"  L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f};  fabric read requests x 64 B = {m.get('TCC_EA0_RDREQ_sum', 0) * 64 / 2**20:.1f} MiB per launch")
if "GRBM_GUI_ACTIVE" in m and "SQ_LDS_IDX_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"  LDS busy {m['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.4f} of the kernel's {cyc:.0f} cycles;  waves parked on s_waitcnt / barriers: "
          f"{m.get('SQ_WAIT_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f} of wave cycles;  bank-conflict cycles {m.get('SQ_LDS_BANK_CONFLICT', -1):.0f}")

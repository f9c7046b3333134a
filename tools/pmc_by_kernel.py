#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters out of a run's *_counter_collection.csv files (and kernel durations from
*_kernel_trace.csv when the run had --kernel-trace): `python tools/pmc_by_kernel.py <rocprof dir> [name filter] [grid size]`.
Sums a counter over its dimension instances per dispatch, then averages over the dispatches of (kernel, grid)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, grid, dispatch) -> counter -> value
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        g = int(r["Grid_Size"])
        if grid and g != grid:
            continue
        per[(k, g, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, g, _), cs in per.items():
    for c, v in cs.items():
        agg[(k, g)][c].append(v)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
        if grid and g != grid:
            continue
        dur[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(set(agg) | set(dur)):
    k, g = key
    print(f"{k[:150]}  grid {g}")
    if dur[key]:
        v = sorted(dur[key])
        print(f"   launches {len(v)}  us median {v[len(v) // 2]:.1f}  min {v[0]:.1f}")
    for c, vals in sorted(agg[key].items()):
        print(f"   {c:28s} {sum(vals) / len(vals):16.1f}   (n={len(vals)})")

#!/usr/bin/env python3
"""Per-workload kernel statistics from a rocprofv3 --kernel-trace of `python3 bench.py` (default command).

The default run times several workloads one after the other and three of them (DPF kInBits 32 / 16 / 64) launch the
SAME kernel template, so rocprofv3's own per-name --stats average mixes them.  This splits the dispatches, in time
order, into phases -- a phase starts at every key-generation launch (gen_kernel / ht_gen_kernel: bench.py makes a key
before each workload) -- and prints calls / average / min / max per kernel name and phase.
The last column is the average over the LAST `timed` calls of the phase (default 20 = the driver's --steps): the
launches bench.py's HIP events bracket, without the warm-up ones before them (the first launch after an idle GPU runs
at a low clock and pulls the all-calls average up).
usage: tools/trace_phases.py <dir with *_kernel_trace.csv> [timed]"""
import csv
import glob
import sys

TIMED = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r.get("Kernel_Id", 0) or 0)))
rows.sort()
# Round 5: after a record's timed launches bench.py runs ~50 ms of the same kernel through the DIAGNOSTIC build of the
# library (in-kernel clock, `clk`): same kernel name, another code object -- a higher Kernel_Id than the product
# library's, which is loaded first.  Those dispatches are listed apart ("[diag]"), never mixed into the timed ones.
first_id = {}
for _, _, name, kid in rows:
    first_id[name] = min(first_id.get(name, kid), kid)
phase, stats, order = 0, {}, []
for st, en, name, kid in rows:
    if "fssamd" not in name:
        continue
    diag = kid != first_id[name]
    if "gen_kernel" in name and "prg_gen" not in name and not diag:
        phase += 1
    key = (phase, ("[diag] " if diag else "") + name.replace("void fssamd::", "").replace("fssamd::", ""))
    if key not in stats:
        stats[key] = []
        order.append(key)
    stats[key].append((en - st) / 1e3)
for key in order:
    v = stats[key]
    t = v[-TIMED:]
    print("phase %2d  %-66s calls %4d  avg_us %9.2f  min_us %9.2f  max_us %9.2f  last_%d_avg_us %9.2f" % (
        key[0], key[1][:66], len(v), sum(v) / len(v), min(v), max(v), len(t), sum(t) / len(t)))

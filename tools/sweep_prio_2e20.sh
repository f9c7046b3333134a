#!/bin/bash
# 2^20 points, fixed shares: the progress-ranking parameters again, on the two-then-one kernel (round 3).
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r3/prio_2e20.txt"; mkdir -p "$(dirname "$O")"; : > "$O"
for rep in 1 2; do
for sch in dpf dcf; do
  for v in "X=0" "FSS_AMD_PRIO_EVERY=0" "FSS_AMD_PRIO_EVERY=1" "FSS_AMD_PRIO_EVERY=7" "FSS_AMD_PRIO_SLACK=1" "FSS_AMD_PRIO_SLACK=2" "FSS_AMD_PRIO_EVERY=1 FSS_AMD_PRIO_SLACK=1" "FSS_AMD_STATIC_PRIO=1" "FSS_AMD_STATIC_PRIO=0"; do
    echo -n "[$sch $v] " >> "$O"
    env $v python3 tools/bench_points.py --scheme $sch --n 32 --log2-points 20 --steady 2>/dev/null | tail -1 >> "$O"
  done
done
done
cat "$O"

#!/bin/bash
# fixed shares with ranked priorities against the dynamic queue at large batches, on the two-then-one kernel (round 3)
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r3/dyn_big.txt"; mkdir -p "$(dirname "$O")"; : > "$O"
for rep in 1 2; do
for lg in 22 23 24; do
  for v in "X=0" "FSS_AMD_DYN=0" "FSS_AMD_DYN=1" "FSS_AMD_DYN=0 FSS_AMD_MIX_TAIL_UNITS=4" "FSS_AMD_DYN=1 FSS_AMD_ROTPRIO=2"; do
    echo -n "[2^$lg $v] " >> "$O"
    env $v python3 tools/bench_points.py --scheme dpf --n 32 --log2-points $lg --steady 2>/dev/null | tail -1 >> "$O"
  done
done
done
cat "$O"

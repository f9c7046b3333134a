#!/bin/bash
# 2^20 points: the per-workgroup dynamic queue (no launch-wide tail) with the two-then-one split, against fixed shares
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r3/dyn_2e20.txt"; mkdir -p "$(dirname "$O")"; : > "$O"
for rep in 1 2; do
  for v in "X=0" "FSS_AMD_DYN=1 FSS_AMD_TAIL_PCT=0 FSS_AMD_MIX_MID_PCT=50" "FSS_AMD_DYN=1 FSS_AMD_TAIL_PCT=0 FSS_AMD_MIX_MID_PCT=50 FSS_AMD_ROTPRIO=2" \
           "FSS_AMD_DYN=1 FSS_AMD_TAIL_PCT=0 FSS_AMD_MIX_MID_PCT=25 FSS_AMD_ROTPRIO=2" "FSS_AMD_DYN=1 FSS_AMD_TAIL_PCT=6 FSS_AMD_MIX_MID_PCT=44 FSS_AMD_ROTPRIO=2"; do
    echo -n "[$v] " >> "$O"
    env $v python3 tools/bench_points.py --scheme dpf --n 32 --log2-points 20 --steady 2>/dev/null | tail -1 >> "$O"
  done
done
cat "$O"
